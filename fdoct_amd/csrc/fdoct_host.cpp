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

}  // namespace fdoct
