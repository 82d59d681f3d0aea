// make_opencv_golden.cpp -- for a maintainer WHO HAS OpenCV: runs the reference's processing block
// (BscanFFT.cpp:1123-1240, BscanFFTsim.cpp:842-955) with the real cv:: calls on this repo's committed input fixtures and
// writes what cv::dft / cv::magnitude produce, so that the oracle (and through it the HIP path) can be pinned by
// reference-held arithmetic.  Not built or run in this repo's image (no OpenCV there); tests consume its output when present.
//   g++ -O2 tools/make_opencv_golden.cpp -o make_opencv_golden $(pkg-config --cflags --libs opencv4)
//   ./make_opencv_golden tests/golden            -> tests/golden/opencv_magI_96x1024.f32 (+ _bscandb_512x96.f64)
// Settings = tests/test_gpu_parity.py::test_reference_fixture_sim_variant's "main u16" case: imgi/backg 96 x 128 u16,
// numfftpoints 1024, numdisplaypoints 512, lambda 816..884 nm, donotnormalize = 1, full-frame background, averages 1.
#include <cstdio>
#include <opencv2/opencv.hpp>
#include <vector>
using namespace cv;

static Mat load_u16(const std::string& path, int rows, int cols) {
  std::vector<unsigned short> v((size_t)rows * cols);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f || fread(v.data(), 2, v.size(), f) != v.size()) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(1); }
  fclose(f);
  Mat m(rows, cols, CV_16UC1, v.data()), d;
  m.convertTo(d, CV_64F);                                               // main:1125
  return d;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "tests/golden";
  const int H = 96, W = 128, N = 1024, D = 512;
  const double lambdamin = 816e-9, lambdamax = 884e-9, pi = 3.141592653589793;
  Mat data_y = load_u16(dir + "/imgi_u16_96x128.bin", H, W), data_yb = load_u16(dir + "/backg_u16_96x128.bin", H, W);
  // one-time tables, main:615-698
  const double deltalambda = (lambdamax - lambdamin) / W;
  std::vector<double> k(W), diffk(W), klinear(N), fractionalk(N);
  std::vector<int> nearestkindex(N, 0);
  for (int i = 0; i < W; i++) k[i] = 2 * pi / (lambdamin + i * deltalambda);
  const double kmin = 2 * pi / (lambdamax - deltalambda), kmax = 2 * pi / lambdamin, deltak = (kmax - kmin) / N;
  for (int f = 0; f < N; f++) klinear[f] = kmin + (f + 1) * deltak;
  for (int i = 1; i < W; i++) diffk[i] = k[i - 1] - k[i];
  diffk[0] = diffk[1];
  for (int f = 0; f < N; f++)
    for (int i = 0; i < W; i++)
      if (k[i] < klinear[f]) { nearestkindex[f] = i; break; }
  for (int f = 0; f < N; f++) fractionalk[f] = (klinear[f] - k[nearestkindex[f]]) / diffk[nearestkindex[f]];
  Mat win(1, W, CV_64F);                                                // main:936-944 (float nn / NN)
  for (int p = 0; p < W; p++) {
    const float nn = p, NN = W - 1;
    win.at<double>(0, p) = 0.62 - 0.48 * std::abs(nn / NN - 0.5) + 0.38 * std::cos(2 * pi * (nn / NN - 0.5));
  }
  data_y = data_y / data_yb;                                            // main:1132 (data_yp = zeros)
  Mat data_ylin(H, N, CV_64F, Scalar(0));                               // columns 0 and N-1 stay 0 (this repo's definition)
  for (int p = 0; p < H; p++) {
    Mat row = data_y.row(p);
    row = row - mean(row)[0];                                           // main:1138-1139
    multiply(row, win, row);                                            // main:1142
    std::vector<double> slopes(W);                                      // main:1153-1161
    for (int q = 1; q < W; q++) slopes[q] = row.at<double>(0, q) - row.at<double>(0, q - 1);
    slopes[0] = slopes[1];
    for (int q = 1; q < N - 1; q++) {                                   // main:1164-1173
      const int i = nearestkindex[q];
      data_ylin.at<double>(p, q) = row.at<double>(0, i) + fractionalk[i] * slopes[i];
    }
  }
  Mat planes[] = {Mat_<float>(data_ylin), Mat::zeros(data_ylin.size(), CV_32F)}, complexI, magI;   // main:1181-1183
  merge(planes, 2, complexI);
  dft(complexI, complexI, DFT_ROWS | DFT_INVERSE);                      // main:1185
  split(complexI, planes);
  magnitude(planes[0], planes[1], magI);                                // main:1190
  Mat bscan, bscandb, mag64;
  magI.colRange(0, D).convertTo(mag64, CV_64F);
  transpose(mag64, bscan);                                              // main:1220
  bscan += 0.00001;                                                     // main:1222
  log(bscan, bscandb);                                                  // main:1235-1236
  bscandb = 20.0 * bscandb / 2.303;
  bscandb.row(4).copyTo(bscandb.row(1));                                // main:1237-1238
  bscandb.row(4).copyTo(bscandb.row(0));
  FILE* f = fopen((dir + "/opencv_magI_96x1024.f32").c_str(), "wb");
  fwrite(magI.ptr<float>(0), 4, (size_t)H * N, f);
  fclose(f);
  f = fopen((dir + "/opencv_bscandb_512x96.f64").c_str(), "wb");
  fwrite(bscandb.ptr<double>(0), 8, (size_t)D * H, f);
  fclose(f);
  printf("wrote %s/opencv_magI_96x1024.f32 and opencv_bscandb_512x96.f64 (OpenCV %s)\n", dir.c_str(), CV_VERSION);
  return 0;
}
