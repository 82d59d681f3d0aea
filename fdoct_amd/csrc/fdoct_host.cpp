// fdoct_host.cpp -- one-time tables of the reconstruction path, in double
// precision on the host, exactly in the reference's order of operations.
#include "fdoct_host.h"

#include <cmath>

namespace fdoct {

// BscanFFT.cpp:609
static const double kPi = 3.141592653589793;

void build_resample_table(int W, int M, int N, double lambdamin, double lambdamax, std::vector<int32_t>& idx,
                          std::vector<double>& frac) {
  const int MW = M * W;
  std::vector<double> k(MW), diffk(MW), klinear(N);
  const double deltalambda = (lambdamax - lambdamin) / W;      // :615
  for (int i = 0; i < MW; i++) {
    const double lambda = lambdamin + i * deltalambda / M;     // :641
    k[i] = 2 * kPi / lambda;                                   // :644
  }
  const double kmin = 2 * kPi / (lambdamax - deltalambda);     // :645
  const double kmax = 2 * kPi / lambdamin;                     // :646
  const double deltak = (kmax - kmin) / N;                     // :647
  for (int f = 0; f < N; f++) klinear[f] = kmin + (f + 1) * deltak;  // :652
  for (int i = 1; i < MW; i++) diffk[i] = k[i - 1] - k[i];     // :667
  if (MW > 1) diffk[0] = diffk[1];                             // :671
  idx.assign(N, 0);
  frac.assign(N, 0.0);
  // first sample whose k is below the linear k (:673-690).  k is strictly
  // decreasing, so a moving cursor finds the same index as the reference's
  // restart-from-zero scan.
  for (int f = 0; f < N; f++) {
    int found = 0;
    for (int i = 0; i < MW; i++) {
      if (k[i] < klinear[f]) {
        found = i;
        break;
      }
    }
    idx[f] = found;
  }
  for (int f = 0; f < N; f++) frac[f] = (klinear[f] - k[idx[f]]) / diffk[idx[f]];  // :695
}

void build_barthann(int W, std::vector<double>& win) {
  win.resize(W);
  for (int p = 0; p < W; p++) {
    const float nn = (float)p;       // :940  float, as in the reference
    const float NN = (float)(W - 1); // :941
    const float ratio = nn / NN;     // float division, then promoted
    win[p] = 0.62 - 0.48 * std::abs(ratio - 0.5) + 0.38 * std::cos(2 * kPi * (ratio - 0.5));
  }
}

// cv::applyColorMap(src, dst, COLORMAP_JET) (main:1284) is a look-up in a 256-entry table that OpenCV does not store but BUILDS
// (imgproc/src/colormap.cpp, the same in 2.4 .. 4.x), in float:
//   * the base map is GNU Octave's jet(256) -- x = linspace(0, 1, 256)', r = 4x - 3/2 on [3/8, 5/8), 1 on [5/8, 7/8),
//     -4x + 9/2 above; g and b the same ramp shifted by 1/4 and 1/2 -- written into the source as 256 float literals per
//     channel (values (k + 1/2) / 255: 0.00588235294117645f = 1.5 / 255, ...);
//   * X = linspace(0.f, 1.f, 256): step = 1.f / 255, X[i] = 0.f + i * step;
//   * lut = interp1(X, channel, X) per channel -- for every X[i] a binary search that ends on low = i - 1, high = i and
//     Y[low] + (X[i] - X[low]) * (Y[high] - Y[low]) / (X[high] - X[low]), all in float (i = 0: Y[0]);
//   * lut.convertTo(CV_8U, 255.): saturate(round-half-even(v * 255.f)).
// Every table value sits half-way between two bytes, so which byte an entry becomes is decided by these float roundings: the
// recipe is followed operation by operation (this file is compiled with -ffp-contract=off).  The Octave doubles are
// recomputed here (4 * (i / 255.0) - 1.5 ...): they differ from the printed literals by parts in 1e16, far inside the float
// the literal rounds to.  Pinned by the end points every OpenCV build shows (0 -> (128, 0, 0), 255 -> (0, 0, 128) in B,G,R)
// and, once a maintainer runs `make -C oracle opencv-golden`, by applyColorMap itself (tests/test_octave_crosscheck.py).
void build_opencv_jet(unsigned char* bgr) {
  float X[256], Y[3][256];  // Y[0] = b, Y[1] = g, Y[2] = r
  const float step = (1.f - 0.f) / (float)(256 - 1);
  for (int i = 0; i < 256; i++) {
    X[i] = 0.f + (float)i * step;
    const double x = (double)i * (1.0 / 255.0);  // Octave: linspace(0, 1, 256)
    const double r = (x >= 3.0 / 8 && x < 5.0 / 8) * (4 * x - 3.0 / 2) + (x >= 5.0 / 8 && x < 7.0 / 8) + (x >= 7.0 / 8) * (-4 * x + 9.0 / 2);
    const double g = (x >= 1.0 / 8 && x < 3.0 / 8) * (4 * x - 1.0 / 2) + (x >= 3.0 / 8 && x < 5.0 / 8) + (x >= 5.0 / 8 && x < 7.0 / 8) * (-4 * x + 7.0 / 2);
    const double b = (x < 1.0 / 8) * (4 * x + 1.0 / 2) + (x >= 1.0 / 8 && x < 3.0 / 8) + (x >= 3.0 / 8 && x < 5.0 / 8) * (-4 * x + 5.0 / 2);
    Y[0][i] = (float)b;
    Y[1][i] = (float)g;
    Y[2][i] = (float)r;
  }
  for (int ch = 0; ch < 3; ch++)
    for (int i = 0; i < 256; i++) {
      const float xi = X[i];
      int low = 0, high = 255;
      if (xi < X[low]) high = 1;
      if (xi > X[high]) low = high - 1;
      while (high - low > 1) {
        const int c = low + ((high - low) >> 1);
        if (xi > X[c])
          low = c;
        else
          high = c;
      }
      volatile float num = (xi - X[low]) * (Y[ch][high] - Y[ch][low]);  // (volatile: each step rounded to float, as written)
      volatile float quo = num / (X[high] - X[low]);
      volatile float yi = Y[ch][low] + quo;
      volatile float scaled = yi * 255.f;
      long v = std::lrint((double)scaled);  // round half to even (the default rounding mode), as cvRound / cvtps2dq
      bgr[3 * i + ch] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

}  // namespace fdoct
