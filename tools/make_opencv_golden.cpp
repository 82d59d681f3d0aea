// make_opencv_golden.cpp -- for a maintainer WHO HAS OpenCV: runs the reference's processing block
// (BscanFFT.cpp:1123-1240, BscanFFTsim.cpp:842-955) and the OpenCV calls either side of it with the real cv:: functions on
// this repo's committed input fixtures and writes what they produce, so that the oracle (and through it the HIP path) can
// be pinned by reference-held arithmetic.  Not built or run in this repo's image (no OpenCV there); the tests consume every
// file it writes when present (tests/test_octave_crosscheck.py::test_opencv_*), one run is enough.
//   make -C oracle opencv-golden        (= the two lines below, then the tests that consume the files)
//   g++ -O2 tools/make_opencv_golden.cpp -o make_opencv_golden $(pkg-config --cflags --libs opencv4)
//   ./make_opencv_golden tests/golden
// Files (all raw little-endian arrays, row-major):
//   opencv_magI_96x1024.f32, opencv_bscandb_512x96.f64   the block, "main" variant: imgi/backg 96 x 128 u16, numfftpoints 1024,
//       numdisplaypoints 512, lambda 816..884 nm, donotnormalize = 1, full-frame background, averages 1   (main:1123-1240)
//   opencv_sim_magI_96x1024.f32      the same through BscanFFTsim.cpp's block: normalize(NORM_MINMAX) always (sim:845)
//   opencv_zeropad_8x640.f64         zeropadrowwise of 8 rows x 160 samples, multiplier 4: the DFT_REAL_OUTPUT reading of a
//                                    2-channel input, fftshift / copyMakeBorder / ifftshift on an even width (main:180-245)
//   opencv_zeropad_odd_8x507.f64, opencv_zeropad_odd_8x381.f64     the same on 127 columns, multipliers 4 and 3: the fftshift that
//                                    leaves an odd last column in place and the M W - 1 columns an even multiplier returns
//   opencv_normalize_96x128.f64, opencv_normalizerows_96x128.f64     normalize(.., 0, 1, NORM_MINMAX) whole frame / per row
//                                    (main:88-97, 1126-1129)
//   opencv_median{3,5}_u16_96x128.bin, opencv_median{3,5,7}_u8_96x128.bin      medianBlur borders (main:953-956)
//   opencv_resize_2x2_{u8,u16}_48x64.bin, opencv_resize_4x3_{u8,u16}_32x32.bin  resize(INTER_AREA) rounding (main:958)
//   opencv_div0_96x128.f64           Mat / Mat with zeros in the divisor (main:1132)
//   opencv_display_512x96.u8         threshold, min-max normalise, x255 -> CV_8U of the bscandb above (main:1242-1255)
//   opencv_jet_256x3.u8              applyColorMap(COLORMAP_JET) of the 0..255 ramp, B,G,R (main:1284)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <opencv2/opencv.hpp>
#include <string>
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

static void dump(const std::string& path, const void* p, size_t bytes) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(1); }
  fclose(f);
  printf("wrote %s (%zu bytes)\n", path.c_str(), bytes);
}

// Mat must be continuous: every Mat dumped here is freshly allocated by OpenCV (clone() where it is a view).
static void dump(const std::string& path, const Mat& m) {
  Mat c = m.isContinuous() ? m : m.clone();
  dump(path, c.ptr<unsigned char>(0), c.total() * c.elemSize());
}

// The zero-pad spectral upsampling of main:180-245, stated with the same cv:: calls in the same order (forward row DFT
// with DFT_SCALE and complex output of the float row, halves swapped, zero border of floor((M-1) W / 2) columns left and
// right, halves swapped back, inverse row DFT with DFT_REAL_OUTPUT).
static Mat upsample_rows(const Mat& rows64, int mult) {
  Mat f32, spec, padded, up, tmp;
  rows64.convertTo(f32, CV_32F);
  dft(f32, spec, DFT_SCALE | DFT_COMPLEX_OUTPUT | DFT_ROWS);
  auto swap_halves = [&](Mat& m) {
    const int half = m.cols / 2;
    Mat left(m, Rect(0, 0, half, m.rows)), right(m, Rect(half, 0, half, m.rows));
    left.copyTo(tmp);
    right.copyTo(left);
    tmp.copyTo(right);
  };
  swap_halves(spec);
  const int extra = (int)std::floor((rows64.cols * mult - rows64.cols) / 2);
  copyMakeBorder(spec, padded, 0, 0, extra, extra, BORDER_CONSTANT, 0.0);
  swap_halves(padded);
  dft(padded, up, DFT_INVERSE | DFT_REAL_OUTPUT | DFT_ROWS);
  up.convertTo(up, CV_64F);
  return up;
}

// The reconstruction block from data_y (CV_64F, after any normalisation) to magI (CV_32F, H x N); main:1132-1190.
static Mat block_magI(Mat data_y, const Mat& data_yb, const Mat& win, const std::vector<int>& nearestkindex,
                      const std::vector<double>& fractionalk, int N) {
  const int H = data_y.rows, W = data_y.cols;
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
  return magI;
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
  Mat magI = block_magI(data_y.clone(), data_yb, win, nearestkindex, fractionalk, N);
  Mat bscan, bscandb, mag64;
  magI.colRange(0, D).convertTo(mag64, CV_64F);
  transpose(mag64, bscan);                                              // main:1220
  bscan += 0.00001;                                                     // main:1222
  log(bscan, bscandb);                                                  // main:1235-1236
  bscandb = 20.0 * bscandb / 2.303;
  bscandb.row(4).copyTo(bscandb.row(1));                                // main:1237-1238
  bscandb.row(4).copyTo(bscandb.row(0));
  dump(dir + "/opencv_magI_96x1024.f32", magI);
  dump(dir + "/opencv_bscandb_512x96.f64", bscandb);

  // ---- BscanFFTsim.cpp's block: the frame is normalised to [0, 1] first, always (sim:845); background as loaded
  {
    Mat y = data_y.clone();
    normalize(y, y, 0, 1, NORM_MINMAX);
    dump(dir + "/opencv_normalize_96x128.f64", y);
    Mat yb01 = data_yb / 65535.0;                                       // a background of the normalised frame's scale
    dump(dir + "/opencv_sim_magI_96x1024.f32", block_magI(y, yb01, win, nearestkindex, fractionalk, N));
    Mat yr = data_y.clone();                                            // normalizerows (main:88-97): each row on its own
    for (int r = 0; r < yr.rows; r++) {
      Mat row = yr.row(r);
      normalize(row, row, 0, 1, NORM_MINMAX);
    }
    dump(dir + "/opencv_normalizerows_96x128.f64", yr);
  }
  // ---- zeropadrowwise at width 160, multiplier 4: row r = the fixture's row r followed by the first 32 samples of row r + 1
  {
    Mat in(8, 160, CV_64F);
    for (int r = 0; r < 8; r++)
      for (int c = 0; c < 160; c++) in.at<double>(r, c) = c < 128 ? data_y.at<double>(r, c) : data_y.at<double>(r + 1, c - 128);
    dump(dir + "/opencv_zeropad_8x640.f64", upsample_rows(in, 4));
    // ... and on ODD widths (main:215-227 swaps two halves of cols / 2 columns and leaves the last one; main:229 pads
    // floor((M W - W) / 2) columns either side): 127 columns x 4 -> 507 columns (M W - 1), 127 x 3 -> 381 (M W)
    Mat odd = in.colRange(0, 127).clone();
    Mat up4 = upsample_rows(odd, 4), up3 = upsample_rows(odd, 3);
    if (up4.cols != 507 || up3.cols != 381) { std::fprintf(stderr, "unexpected widths %d %d\n", up4.cols, up3.cols); return 2; }
    dump(dir + "/opencv_zeropad_odd_8x507.f64", up4);
    dump(dir + "/opencv_zeropad_odd_8x381.f64", up3);
  }
  // ---- the frame-source tail: medianBlur and resize(INTER_AREA) on the camera's integer types (main:953-958)
  {
    Mat u16, t;
    data_y.convertTo(u16, CV_16U);                                      // the fixture's own integers
    Mat u8(u16.size(), CV_8UC1);                                        // the 8-bit camera's view of it: imgi >> 8
    for (int r = 0; r < u16.rows; r++)
      for (int c = 0; c < u16.cols; c++) u8.at<unsigned char>(r, c) = (unsigned char)(u16.at<unsigned short>(r, c) >> 8);
    for (int n : {3, 5}) {
      medianBlur(u16, t, n);
      dump(dir + "/opencv_median" + std::to_string(n) + "_u16_96x128.bin", t);
    }
    for (int n : {3, 5, 7}) {
      medianBlur(u8, t, n);
      dump(dir + "/opencv_median" + std::to_string(n) + "_u8_96x128.bin", t);
    }
    resize(u16, t, Size(), 1.0 / 2, 1.0 / 2, INTER_AREA);
    dump(dir + "/opencv_resize_2x2_u16_48x64.bin", t);
    resize(u8, t, Size(), 1.0 / 2, 1.0 / 2, INTER_AREA);
    dump(dir + "/opencv_resize_2x2_u8_48x64.bin", t);
    resize(u16, t, Size(), 1.0 / 4, 1.0 / 3, INTER_AREA);              // binvaluex = 4, binvaluey = 3 (BscanFFTspinjnt.cpp:1553)
    dump(dir + "/opencv_resize_4x3_u16_32x32.bin", t);
    resize(u8, t, Size(), 1.0 / 4, 1.0 / 3, INTER_AREA);
    dump(dir + "/opencv_resize_4x3_u8_32x32.bin", t);
  }
  // ---- Mat / Mat with zeros in the divisor (main:1132): every 7th background sample set to 0
  {
    Mat yb0 = data_yb.clone();
    for (int r = 0; r < yb0.rows; r++)
      for (int c = (r % 7); c < yb0.cols; c += 7) yb0.at<double>(r, c) = 0.0;
    Mat q = data_y / yb0;
    dump(dir + "/opencv_div0_96x128.f64", q);
  }
  // ---- display chain of main:1242-1255 on the bscandb above (bscanthreshold = -30, no clampupper), and the JET table
  {
    Mat disp, disp8;
    max(bscandb, -30.0, disp);
    normalize(disp, disp, 0, 1, NORM_MINMAX);
    disp.convertTo(disp8, CV_8UC1, 255.0);
    dump(dir + "/opencv_display_512x96.u8", disp8);
    Mat ramp(1, 256, CV_8UC1), jet;
    for (int i = 0; i < 256; i++) ramp.at<unsigned char>(0, i) = (unsigned char)i;
    applyColorMap(ramp, jet, COLORMAP_JET);                             // main:1284
    dump(dir + "/opencv_jet_256x3.u8", jet);
  }
  printf("OpenCV %s\n", CV_VERSION);
  return 0;
}
