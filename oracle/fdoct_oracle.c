/*
 * fdoct_oracle.c -- CPU restatement of hn-88/FDOCT's reconstruction block.
 *
 * TEST INFRASTRUCTURE ONLY (see fdoct_oracle.h).  PARITY UNPINNED: the
 * reference holds no expected outputs and its OpenCV dependency is absent, so
 * this file follows the reference source literally instead; every function
 * cites the lines it restates.  OpenCV semantics restated here (documented
 * behaviour of the calls on the path):
 *   - cv::dft(DFT_INVERSE) uses the +i exponent and does not scale unless
 *     DFT_SCALE is given; float input => float arithmetic.
 *   - cv::normalize(NORM_MINMAX, a, b): dst = src*s + (a - min*s),
 *     s = (b-a)/(max-min), s = 0 when max-min < DBL_EPSILON.
 *   - Mat / Mat with a zero divisor gives 0 (OpenCV 3.x, the author's build
 *     links 3.3); we adopt that definition.
 *   - Mat_<float>(Mat64F) rounds to nearest; cv::magnitude = sqrt(re^2+im^2)
 *     in float; cv::log is the natural log.
 *
 * Documented deviations from the reference (all are undefined behaviour
 * there): data_ylin columns 0 and N-1 are never written and the Mat is
 * allocated uninitialised (main:553,1164) -- defined as 0 here;
 * fractionalk.at(idx) with idx >= N reads past the table -- defined as 0;
 * `slopes` is allocated with swapped dimensions (main:626/632) -- we keep a
 * per-row slope buffer instead.
 *
 * Build: gcc -O2 -ffp-contract=off [-fopenmp] -shared -fPIC (see Makefile).
 */
#include "fdoct_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.141592653589793 /* main:609 */

/* ------------------------------------------------- display post-chain -- */
/* main:1242-1255 */
void orc_display_u8(const double *db, int rows, int cols, double thr, int clampupper, uint8_t *gray) {
  const size_t n = (size_t)rows * cols;
  double *t = (double *)malloc(n * sizeof(double));
  for (size_t i = 0; i < n; i++) t[i] = db[i] > thr ? db[i] : thr; /* main:1247 */
  if (clampupper && rows > 5 && cols > 5) t[(size_t)5 * cols + 5] = 50.0; /* main:1252 */
  orc_normalize_minmax(t, n, 0.0, 1.0);                               /* main:1254 */
  for (size_t i = 0; i < n; i++) {                                    /* main:1255 */
    double r = rint(t[i] * 255.0);
    gray[i] = (uint8_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
  }
  free(t);
}

/* main:1284 */
void orc_apply_lut(const uint8_t *gray, size_t n, const uint8_t *lut, uint8_t *bgr) {
  for (size_t i = 0; i < n; i++) {
    bgr[3 * i] = lut[3 * gray[i]];
    bgr[3 * i + 1] = lut[3 * gray[i] + 1];
    bgr[3 * i + 2] = lut[3 * gray[i] + 2];
  }
}

/* main:1227-1230 (makeonlypositive 173-178), 1260-1261 */
void orc_lockin_db(const double *bscan, const double *jscan, size_t n, double *out) {
  for (size_t i = 0; i < n; i++) {
    double d = bscan[i] - jscan[i];
    d = d > 0.0 ? d : 0.0;
    d += 0.001;
    out[i] = 20.0 * log(d) / 2.303;
  }
}

const char *orc_version(void) { return "fdoct-oracle 1 (parity unpinned)"; }

/* ------------------------------------------------------------------ A0 -- */
/* main:615-698.  lambdas[i] = lmin + i*dl/M (641); k = 2*pi/lambdas (644);
 * kmin = 2*pi/(lmax-dl) (645); kmax = 2*pi/lmin (646); deltak (647);
 * klinear[f] = kmin + (f+1)*deltak (652); diffk[i] = k[i-1]-k[i],
 * diffk[0] = diffk[1] (663-671); nearestkindex[f] = first i with
 * k[i] < klinear[f], stays 0 if none (673-690); fractionalk[f] =
 * (klinear[f] - k[idx[f]]) / diffk[idx[f]] (692-698). */
int orc_tables(int W, int M, int N, double lambdamin, double lambdamax,
               int32_t *idx, double *frac, double *k_out, double *klin_out,
               double *diffk_out) {
  const int MW = M * W;
  double *k = (double *)malloc(sizeof(double) * (size_t)MW);
  double *diffk = (double *)malloc(sizeof(double) * (size_t)MW);
  double *klinear = (double *)malloc(sizeof(double) * (size_t)N);
  const double pi = ORC_PI;
  const double deltalambda = (lambdamax - lambdamin) / W; /* main:615 */
  for (int i = 0; i < MW; i++) {
    double lam = lambdamin + i * deltalambda / M; /* main:641 */
    k[i] = 2 * pi / lam;                         /* main:644 */
  }
  const double kmin = 2 * pi / (lambdamax - deltalambda); /* main:645 */
  const double kmax = 2 * pi / lambdamin;                 /* main:646 */
  const double deltak = (kmax - kmin) / N;                /* main:647 */
  for (int f = 0; f < N; f++) klinear[f] = kmin + (f + 1) * deltak;
  for (int i = 1; i < MW; i++) diffk[i] = k[i - 1] - k[i];
  diffk[0] = MW > 1 ? diffk[1] : 1.0;
  for (int f = 0; f < N; f++) {
    idx[f] = 0; /* Mat::zeros, main:620 */
    for (int i = 0; i < MW; i++) {
      if (k[i] < klinear[f]) {
        idx[f] = i;
        break;
      }
    }
  }
  for (int f = 0; f < N; f++)
    frac[f] = (klinear[f] - k[idx[f]]) / diffk[idx[f]];
  if (k_out) memcpy(k_out, k, sizeof(double) * (size_t)MW);
  if (diffk_out) memcpy(diffk_out, diffk, sizeof(double) * (size_t)MW);
  if (klin_out) memcpy(klin_out, klinear, sizeof(double) * (size_t)N);
  free(k);
  free(diffk);
  free(klinear);
  return 0;
}

/* ------------------------------------------------------------------ A1 -- */
/* main:936-944: float nn = p, NN = W-1; nn/NN is a FLOAT division, then
 * promoted to double for the 0.62 - 0.48*|.| + 0.38*cos(2*pi*(.)) terms. */
void orc_barthann(int W, double *win) {
  const double pi = ORC_PI;
  for (int p = 0; p < W; p++) {
    float nn = (float)p;
    float NN = (float)(W - 1);
    float r = nn / NN;
    win[p] = 0.62 - 0.48 * fabs(r - 0.5) + 0.38 * cos(2 * pi * (r - 0.5));
  }
}

/* ----------------------------------------------------------------- A11 -- */
void orc_normalize_minmax(double *y, size_t n, double lo, double hi) {
  if (n == 0) return;
  double smin = y[0], smax = y[0];
  for (size_t i = 1; i < n; i++) {
    if (y[i] < smin) smin = y[i];
    if (y[i] > smax) smax = y[i];
  }
  double dmin = lo < hi ? lo : hi, dmax = lo < hi ? hi : lo;
  double scale = (dmax - dmin) * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
  double shift = dmin - smin * scale;
  for (size_t i = 0; i < n; i++) y[i] = y[i] * scale + shift;
}

/* main:88-97 */
void orc_normalizerows(double *y, int H, int W, double lo, double hi) {
  for (int r = 0; r < H; r++) orc_normalize_minmax(y + (size_t)r * W, (size_t)W, lo, hi);
}

/* main:247-304: (2n+1)-tap average, truncated taps replaced by the centre
 * sample, centre counted twice, divisor 2*(n+1). */
void orc_smoothmovavg(const double *src, double *dst, int H, int W, int n) {
  for (int si = 0; si < H; si++) {
    const double *s = src + (size_t)si * W;
    double *d = dst + (size_t)si * W;
    for (int sj = 0; sj < W; sj++) {
      double ssum = 0;
      for (int sk = -n; sk < n + 1; sk++) {
        int ii = sj + sk;
        if (ii > -1 && ii < W)
          ssum = ssum + s[ii];
        else
          ssum = ssum + s[sj];
      }
      ssum = ssum + s[sj];
      d[sj] = ssum / 2 / (n + 1);
    }
  }
}

/* ------------------------------------------------------------------ A7 -- */
/* DFT model.  Power-of-two lengths: iterative radix-2 decimation in time.
 * Other lengths: recursive mixed radix over the smallest prime factor.
 * Twiddles are computed in double and (for the f32 flavour) rounded once. */
static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

#define DEFINE_FFT(SUF, T)                                                         \
  typedef struct {                                                                 \
    int n;                                                                         \
    int inverse;                                                                   \
    T *tw;      /* n (re,im) pairs: exp(sign*2*pi*i*j/n) */                        \
    int *rev;   /* bit reversal (pow2 only) */                                     \
    T *scratch; /* 6*n scalars: n pairs out + 2n pairs butterfly temp */                                                    \
  } plan_##SUF;                                                                    \
  static void plan_init_##SUF(plan_##SUF *pl, int n, int inverse) {                \
    pl->n = n;                                                                     \
    pl->inverse = inverse;                                                         \
    pl->tw = (T *)malloc(sizeof(T) * 2 * (size_t)n);                               \
    pl->scratch = (T *)malloc(sizeof(T) * 6 * (size_t)n);                          \
    pl->rev = NULL;                                                                \
    double sgn = inverse ? 1.0 : -1.0;                                             \
    for (int j = 0; j < n; j++) {                                                  \
      double a = sgn * 2.0 * ORC_PI * (double)j / (double)n;                       \
      pl->tw[2 * j] = (T)cos(a);                                                   \
      pl->tw[2 * j + 1] = (T)sin(a);                                               \
    }                                                                              \
    if (is_pow2(n)) {                                                              \
      pl->rev = (int *)malloc(sizeof(int) * (size_t)n);                            \
      int bits = 0;                                                                \
      while ((1 << bits) < n) bits++;                                              \
      for (int i = 0; i < n; i++) {                                                \
        int r = 0;                                                                 \
        for (int b = 0; b < bits; b++)                                             \
          if (i & (1 << b)) r |= 1 << (bits - 1 - b);                              \
        pl->rev[i] = r;                                                            \
      }                                                                            \
    }                                                                              \
  }                                                                                \
  static void plan_free_##SUF(plan_##SUF *pl) {                                    \
    free(pl->tw);                                                                  \
    free(pl->scratch);                                                             \
    free(pl->rev);                                                                 \
  }                                                                                \
  static void fft_pow2_##SUF(const plan_##SUF *pl, T *x) {                         \
    const int n = pl->n;                                                           \
    for (int i = 0; i < n; i++) {                                                  \
      int r = pl->rev[i];                                                          \
      if (r > i) {                                                                 \
        T tr = x[2 * i], ti = x[2 * i + 1];                                        \
        x[2 * i] = x[2 * r];                                                       \
        x[2 * i + 1] = x[2 * r + 1];                                               \
        x[2 * r] = tr;                                                             \
        x[2 * r + 1] = ti;                                                         \
      }                                                                            \
    }                                                                              \
    for (int len = 2; len <= n; len <<= 1) {                                       \
      const int half = len >> 1, step = n / len;                                   \
      for (int base = 0; base < n; base += len) {                                  \
        for (int j = 0; j < half; j++) {                                           \
          const T wr = pl->tw[2 * j * step], wi = pl->tw[2 * j * step + 1];        \
          T *a = x + 2 * (base + j), *b = x + 2 * (base + j + half);               \
          T br = b[0] * wr - b[1] * wi, bi = b[0] * wi + b[1] * wr;                \
          b[0] = a[0] - br;                                                        \
          b[1] = a[1] - bi;                                                        \
          a[0] = a[0] + br;                                                        \
          a[1] = a[1] + bi;                                                        \
        }                                                                          \
      }                                                                            \
    }                                                                              \
  }                                                                                \
  static int smallest_factor_##SUF(int n) {                                        \
    for (int p = 2; p * p <= n; p++)                                               \
      if (n % p == 0) return p;                                                    \
    return n;                                                                      \
  }                                                                                \
  /* out[0..n) = DFT of in[0], in[stride], ...; twiddle w_n^j = tw[j*twstride] */  \
  static void fft_rec_##SUF(const plan_##SUF *pl, const T *in, T *out, int n,      \
                            int stride, int twstride, T *tmp) {                    \
    if (n == 1) {                                                                  \
      out[0] = in[0];                                                              \
      out[1] = in[1];                                                              \
      return;                                                                      \
    }                                                                              \
    const int p = smallest_factor_##SUF(n), m = n / p;                             \
    for (int r = 0; r < p; r++)                                                    \
      fft_rec_##SUF(pl, in + 2 * (size_t)r * stride, out + 2 * (size_t)r * m, m,   \
                    stride * p, twstride * p, tmp);                                \
    const int N = pl->n;                                                           \
    for (int k = 0; k < m; k++) {                                                  \
      for (int r = 0; r < p; r++) { /* twiddle the sub-results */                  \
        const size_t ti = ((size_t)r * k * twstride) % (size_t)N;                  \
        const T wr = pl->tw[2 * ti], wi = pl->tw[2 * ti + 1];                      \
        const T xr = out[2 * (r * m + k)], xi = out[2 * (r * m + k) + 1];          \
        tmp[2 * r] = xr * wr - xi * wi;                                            \
        tmp[2 * r + 1] = xr * wi + xi * wr;                                        \
      }                                                                            \
      for (int q = 0; q < p; q++) { /* p-point DFT */                              \
        T sr = 0, si = 0;                                                          \
        for (int r = 0; r < p; r++) {                                              \
          const size_t ti = ((size_t)r * q * m * twstride) % (size_t)N;            \
          const T wr = pl->tw[2 * ti], wi = pl->tw[2 * ti + 1];                    \
          sr += tmp[2 * r] * wr - tmp[2 * r + 1] * wi;                             \
          si += tmp[2 * r] * wi + tmp[2 * r + 1] * wr;                             \
        }                                                                          \
        /* store after all q are computed: use second half of tmp */               \
        tmp[2 * (p + q)] = sr;                                                     \
        tmp[2 * (p + q) + 1] = si;                                                 \
      }                                                                            \
      for (int q = 0; q < p; q++) {                                                \
        out[2 * (k + q * m)] = tmp[2 * (p + q)];                                   \
        out[2 * (k + q * m) + 1] = tmp[2 * (p + q) + 1];                           \
      }                                                                            \
    }                                                                              \
  }                                                                                \
  static void fft_exec_##SUF(const plan_##SUF *pl, T *x, int scale) {              \
    const int n = pl->n;                                                           \
    if (pl->rev) {                                                                 \
      fft_pow2_##SUF(pl, x);                                                       \
    } else {                                                                       \
      T *out = pl->scratch, *tmp = pl->scratch + 2 * (size_t)n;                    \
      fft_rec_##SUF(pl, x, out, n, 1, 1, tmp);                                     \
      memcpy(x, out, sizeof(T) * 2 * (size_t)n);                                   \
    }                                                                              \
    if (scale) {                                                                   \
      const T s = (T)(1.0 / n);                                                    \
      for (int i = 0; i < 2 * n; i++) x[i] *= s;                                   \
    }                                                                              \
  }

DEFINE_FFT(f32, float)
DEFINE_FFT(f64, double)

void orc_dft_rows_f32(float *data, int H, int N, int inverse, int scale) {
  plan_f32 pl;
  plan_init_f32(&pl, N, inverse);
  for (int r = 0; r < H; r++) fft_exec_f32(&pl, data + 2 * (size_t)r * N, scale);
  plan_free_f32(&pl);
}

void orc_dft_rows_f64(double *data, int H, int N, int inverse, int scale) {
  plan_f64 pl;
  plan_init_f64(&pl, N, inverse);
  for (int r = 0; r < H; r++) fft_exec_f64(&pl, data + 2 * (size_t)r * N, scale);
  plan_free_f64(&pl);
}

/* ------------------------------------------------------------------ A4 -- */
/* main:180-245.  convertTo(CV_32F) (209); forward dft DFT_SCALE|
 * DFT_COMPLEX_OUTPUT (211); fftshift = swap halves (224-227); copyMakeBorder
 * floor((MW-W)/2) zeros each side (229); ifftshift (233-239); inverse dft
 * DFT_REAL_OUTPUT (241): cv::dft then treats the complex input as the CCS
 * (Hermitian) half spectrum, i.e. only bins 0..n/2 are read and the imaginary
 * parts of bins 0 and n/2 are ignored; convertTo(CV_64F) (242). */
void orc_zeropadrowwise(const double *y, int H, int W, int M, int bandpass,
                        double *out) {
  const int MW = M * W;
  const int pad = (MW - W) / 2; /* floor, main:229 */
  /* The padded spectrum -- and with it the inverse transform and the returned Mat -- has W + 2 pad columns: M W, or M W - 1
   * when W is odd and M even (main:229, 241).  The reference then reads data_y with nearestkindex < M W (main:1167): column
   * M W - 1 of such a row does not exist there (an out-of-bounds read); it is defined as 0 here, like data_ylin[0]. */
  const int zplen = W + 2 * pad;
  plan_f32 fwd, inv;
  plan_init_f32(&fwd, W, 0);
  plan_init_f32(&inv, zplen, 1);
  float *f = (float *)malloc(sizeof(float) * 2 * (size_t)W);
  float *sh = (float *)malloc(sizeof(float) * 2 * (size_t)W);
  float *zp = (float *)calloc(2 * (size_t)(W + 2 * ((MW - W) / 2) + 2), sizeof(float));
  float *g = (float *)malloc(sizeof(float) * 2 * (size_t)MW);
  for (int r = 0; r < H; r++) {
    for (int i = 0; i < W; i++) {
      f[2 * i] = (float)y[(size_t)r * W + i];
      f[2 * i + 1] = 0.f;
    }
    fft_exec_f32(&fwd, f, 1);
    const int cx = W / 2; /* main:215 */
    /* swap LHS/RHS halves of width cx (if W is odd the last column stays) */
    memcpy(sh, f, sizeof(float) * 2 * (size_t)W);
    for (int i = 0; i < cx; i++) {
      sh[2 * i] = f[2 * (i + cx)];
      sh[2 * i + 1] = f[2 * (i + cx) + 1];
      sh[2 * (i + cx)] = f[2 * i];
      sh[2 * (i + cx) + 1] = f[2 * i + 1];
    }
    if (bandpass) { /* dark:218-236 */
      int dcl = W / 2 - (int)floor(W / 10);
      int dcr = W / 2 + (int)floor(W / 10);
      for (int i = 0; i < dcl && i < W; i++) sh[2 * i] = sh[2 * i + 1] = 0.f;
      for (int i = dcr; i < dcr + dcl && i < W; i++) sh[2 * i] = sh[2 * i + 1] = 0.f;
      int dcvals = 3;
      dcl = W / 2 - dcvals;
      for (int i = dcl; i < dcl + 2 * dcvals && i < W; i++)
        if (i >= 0) sh[2 * i] = sh[2 * i + 1] = 0.f;
    }
    memset(zp, 0, sizeof(float) * 2 * (size_t)zplen);
    memcpy(zp + 2 * (size_t)pad, sh, sizeof(float) * 2 * (size_t)W);
    /* ifftshift: swap halves of width zplen/2 (main:233-239) */
    const int cz = zplen / 2;
    for (int i = 0; i < 2 * MW; i++) g[i] = 0.f;
    for (int i = 0; i < cz; i++) {
      g[2 * i] = zp[2 * (i + cz)];
      g[2 * i + 1] = zp[2 * (i + cz) + 1];
      g[2 * (i + cz)] = zp[2 * i];
      g[2 * (i + cz) + 1] = zp[2 * i + 1];
    }
    /* DFT_REAL_OUTPUT: Hermitian-extend bins 0..n/2, drop imag of 0 and n/2 */
    const int n = zplen; /* the inverse transform's length */
    g[1] = 0.f;
    if (n % 2 == 0) g[2 * (n / 2) + 1] = 0.f;
    for (int kk = 1; kk < (n + 1) / 2; kk++) {
      g[2 * (n - kk)] = g[2 * kk];
      g[2 * (n - kk) + 1] = -g[2 * kk + 1];
    }
    fft_exec_f32(&inv, g, 0);
    for (int i = 0; i < MW; i++) out[(size_t)r * MW + i] = i < zplen ? (double)g[2 * i] : 0.0;
  }
  free(f);
  free(sh);
  free(zp);
  free(g);
  plan_free_f32(&fwd);
  plan_free_f32(&inv);
}

/* The same stage with NO float rounding (truth mode, see orc_params.truth): main:209's convertTo(CV_32F) and both float
 * DFTs (main:211, 241) evaluated in double.  Same readings of cv::dft as above (CCS half spectrum, imaginary parts of bins 0
 * and n/2 ignored, Nyquist bin of the W-point spectrum dropped by the fftshift + DFT_REAL_OUTPUT pair). */
static void zeropadrowwise_f64(const double *y, int H, int W, int M, int bandpass, double *out) {
  const int MW = M * W;
  const int pad = (MW - W) / 2;
  const int zplen = W + 2 * pad;
  plan_f64 fwd, inv;
  plan_init_f64(&fwd, W, 0);
  plan_init_f64(&inv, zplen, 1);
  double *f = (double *)malloc(sizeof(double) * 2 * (size_t)W);
  double *sh = (double *)malloc(sizeof(double) * 2 * (size_t)W);
  double *zp = (double *)calloc(2 * (size_t)(zplen + 2), sizeof(double));
  double *g = (double *)malloc(sizeof(double) * 2 * (size_t)MW);
  for (int r = 0; r < H; r++) {
    for (int i = 0; i < W; i++) {
      f[2 * i] = y[(size_t)r * W + i];
      f[2 * i + 1] = 0.0;
    }
    fft_exec_f64(&fwd, f, 1);
    const int cx = W / 2;
    memcpy(sh, f, sizeof(double) * 2 * (size_t)W);
    for (int i = 0; i < cx; i++) {
      sh[2 * i] = f[2 * (i + cx)];
      sh[2 * i + 1] = f[2 * (i + cx) + 1];
      sh[2 * (i + cx)] = f[2 * i];
      sh[2 * (i + cx) + 1] = f[2 * i + 1];
    }
    if (bandpass) { /* dark:218-236 */
      int dcl = W / 2 - (int)floor(W / 10);
      int dcr = W / 2 + (int)floor(W / 10);
      for (int i = 0; i < dcl && i < W; i++) sh[2 * i] = sh[2 * i + 1] = 0.0;
      for (int i = dcr; i < dcr + dcl && i < W; i++) sh[2 * i] = sh[2 * i + 1] = 0.0;
      int dcvals = 3;
      dcl = W / 2 - dcvals;
      for (int i = dcl; i < dcl + 2 * dcvals && i < W; i++)
        if (i >= 0) sh[2 * i] = sh[2 * i + 1] = 0.0;
    }
    memset(zp, 0, sizeof(double) * 2 * (size_t)zplen);
    memcpy(zp + 2 * (size_t)pad, sh, sizeof(double) * 2 * (size_t)W);
    const int cz = zplen / 2;
    for (int i = 0; i < 2 * MW; i++) g[i] = 0.0;
    for (int i = 0; i < cz; i++) {
      g[2 * i] = zp[2 * (i + cz)];
      g[2 * i + 1] = zp[2 * (i + cz) + 1];
      g[2 * (i + cz)] = zp[2 * i];
      g[2 * (i + cz) + 1] = zp[2 * i + 1];
    }
    const int n = zplen;
    g[1] = 0.0;
    if (n % 2 == 0) g[2 * (n / 2) + 1] = 0.0;
    for (int kk = 1; kk < (n + 1) / 2; kk++) {
      g[2 * (n - kk)] = g[2 * kk];
      g[2 * (n - kk) + 1] = -g[2 * kk + 1];
    }
    fft_exec_f64(&inv, g, 0);
    for (int i = 0; i < MW; i++) out[(size_t)r * MW + i] = i < zplen ? g[2 * i] : 0.0;
  }
  free(f);
  free(sh);
  free(zp);
  free(g);
  plan_free_f64(&fwd);
  plan_free_f64(&inv);
}

/* -------------------------------------------------------------- A2..A8 -- */
/* Scratch Mats of one frame.  The reference allocates its temporaries per frame
 * through cv::Mat; here they are allocated once per driver call so the timed CPU
 * baseline is not dominated by page faults of fresh 16-32 MB allocations. */
typedef struct {
  double *data_y, *tmp, *yup, *ylin, *slopes; /* slopes: threads x MW */
  float *cplx;
  double *cplxd; /* truth mode: the complex rows in double */
  int nth;
} frame_ws;

static int ws_init(frame_ws *ws, const orc_params *p) {
  const size_t HW = (size_t)p->H * p->W, MW = (size_t)p->M * p->W;
  ws->nth = p->threads > 1 ? p->threads : 1;
  ws->data_y = (double *)malloc(sizeof(double) * HW);
  ws->tmp = (double *)malloc(sizeof(double) * HW);
  ws->yup = p->M > 1 ? (double *)malloc(sizeof(double) * (size_t)p->H * MW) : NULL;
  ws->ylin = (double *)malloc(sizeof(double) * (size_t)p->H * p->N);
  ws->slopes = (double *)malloc(sizeof(double) * MW * (size_t)ws->nth);
  ws->cplx = p->truth ? NULL : (float *)malloc(sizeof(float) * 2 * (size_t)p->H * p->N);
  ws->cplxd = p->truth ? (double *)malloc(sizeof(double) * 2 * (size_t)p->H * p->N) : NULL;
  if (!ws->data_y || !ws->tmp || (p->M > 1 && !ws->yup) || !ws->ylin || !ws->slopes || (!ws->cplx && !ws->cplxd)) return -1;
  return 0;
}

static void ws_free(frame_ws *ws) {
  free(ws->data_y);
  free(ws->tmp);
  free(ws->yup);
  free(ws->ylin);
  free(ws->slopes);
  free(ws->cplx);
  free(ws->cplxd);
}

static int frame_to_mag_ws(const orc_params *p, frame_ws *ws, const double *data_y_in,
                           const double *yb, const double *yp, const double *yd,
                           const double *win, const int32_t *idx, const double *frac,
                           const float *phase, float *magI, double *magD, double *ylin_dbg) {
  const int W = p->W, H = p->H, N = p->N, M = p->M;
  const int MW = M * W;
  const size_t HW = (size_t)H * W;
  const int nth = ws->nth;
  if (p->truth ? !magD : !magI) return -3;
  (void)nth;
  double *data_y = ws->data_y, *tmp = ws->tmp;

  /* main:1125 data_y.convertTo(data_y, CV_64F) -- one copy pass */
  memcpy(data_y, data_y_in, sizeof(double) * HW);

  /* main:990-991 smoothing by weighted moving average (before the block) */
  if (p->movavgn > 0) {
    orc_smoothmovavg(data_y, tmp, H, W, p->movavgn);
    memcpy(data_y, tmp, sizeof(double) * HW);
  }
  /* dark:1269 data_y = data_y - data_yd */
  if (yd)
    for (size_t i = 0; i < HW; i++) data_y[i] = data_y[i] - yd[i];
  /* main:1126-1129 */
  if (p->rowwisenormalize) orc_normalizerows(data_y, H, W, 0, 1);
  if (!p->donotnormalize) orc_normalize_minmax(data_y, HW, 0, 1);

  /* main:1132 data_y = (data_y - data_yp) / data_yb : two MatExpr passes */
#pragma omp parallel for num_threads(nth) if (nth > 1)
  for (int r = 0; r < H; r++)
    for (int c = 0; c < W; c++) {
      size_t i = (size_t)r * W + c;
      tmp[i] = data_y[i] - yp[i];
    }
#pragma omp parallel for num_threads(nth) if (nth > 1)
  for (int r = 0; r < H; r++)
    for (int c = 0; c < W; c++) {
      size_t i = (size_t)r * W + c;
      data_y[i] = yb[i] != 0.0 ? tmp[i] / yb[i] : 0.0; /* x/0 = 0, OpenCV 3.x */
    }

  /* main:1135-1143 per row: DC removal, windowing */
#pragma omp parallel for num_threads(nth) if (nth > 1)
  for (int r = 0; r < H; r++) {
    double *row = data_y + (size_t)r * W;
    double s = 0;
    for (int c = 0; c < W; c++) s += row[c];
    const double meanval = s / W;
    for (int c = 0; c < W; c++) row[c] = row[c] - meanval;
    for (int c = 0; c < W; c++) row[c] = row[c] * win[c];
  }

  /* main:1146-1147 zero-pad upsample */
  double *yup = data_y;
  if (M > 1) {
    yup = ws->yup;
    if (p->truth)
      zeropadrowwise_f64(data_y, H, W, M, p->bandpass, yup);
    else
      orc_zeropadrowwise(data_y, H, W, M, p->bandpass, yup);
  }

  /* main:1151-1177 interpolate to linear k space */
  double *ylin = ws->ylin;
#pragma omp parallel for num_threads(nth) if (nth > 1)
  for (int r = 0; r < H; r++) {
    const double *row = yup + (size_t)r * MW;
#ifdef _OPENMP
    double *slopes = ws->slopes + (size_t)omp_get_thread_num() * MW;
#else
    double *slopes = ws->slopes;
#endif
    for (int q = 1; q < MW; q++) slopes[q] = row[q] - row[q - 1]; /* main:1156 */
    slopes[0] = slopes[1];                                        /* main:1161 */
    double *lin = ylin + (size_t)r * N;
    lin[0] = 0.0;     /* never written by the reference (main:1164): defined 0 */
    lin[N - 1] = 0.0; /* idem */
    for (int q = 1; q < N - 1; q++) { /* main:1164-1173 */
      const int i = idx[q];
      const double fr = (i < N) ? frac[i] : 0.0; /* fractionalk[nearestkindex[q]] */
      lin[q] = row[i] + fr * slopes[i];
    }
  }
  if (ylin_dbg) memcpy(ylin_dbg, ylin, sizeof(double) * (size_t)H * N);

  if (p->truth) {
    /* Truth mode: main:1181's narrowing, the DFT of main:1185 and main:1190's magnitude in double -- the value of the
     * reference's mathematics on these inputs and tables, free of every float rounding (good to ~1e-15 of the row's norm).
     * The phasors of the A6' extension are the float pairs the caller holds, promoted. */
    double *cd = ws->cplxd;
    for (int r = 0; r < H; r++)
      for (int q = 0; q < N; q++) {
        const size_t i = (size_t)r * N + q;
        cd[2 * i] = phase ? ylin[i] * (double)phase[2 * q] : ylin[i];
        cd[2 * i + 1] = phase ? ylin[i] * (double)phase[2 * q + 1] : 0.0;
      }
    plan_f64 pl;
    plan_init_f64(&pl, N, 1);
    for (int r = 0; r < H; r++) fft_exec_f64(&pl, cd + 2 * (size_t)r * N, 0);
    plan_free_f64(&pl);
    for (size_t i = 0; i < (size_t)H * N; i++) magD[i] = sqrt(cd[2 * i] * cd[2 * i] + cd[2 * i + 1] * cd[2 * i + 1]);
    return 0;
  }

  /* main:1181-1183 Mat_<float>(data_ylin), zeros plane, merge */
  float *cplx = ws->cplx;
#pragma omp parallel for num_threads(nth) if (nth > 1)
  for (int r = 0; r < H; r++)
    for (int q = 0; q < N; q++) {
      const size_t i = (size_t)r * N + q;
      const float v = (float)ylin[i];
      if (phase) { /* A6' extension: ylin * exp(i*phi), float */
        cplx[2 * i] = v * phase[2 * q];
        cplx[2 * i + 1] = v * phase[2 * q + 1];
      } else {
        cplx[2 * i] = v;
        cplx[2 * i + 1] = 0.f;
      }
    }

  /* main:1185 dft(complexI, complexI, DFT_ROWS | DFT_INVERSE) */
  {
    plan_f32 pl;
    plan_init_f32(&pl, N, 1);
    if (nth > 1) {
#pragma omp parallel num_threads(nth)
      {
        plan_f32 mine = pl; /* private scratch for non-pow2 */
        mine.scratch = (float *)malloc(sizeof(float) * 6 * (size_t)N);
#pragma omp for
        for (int r = 0; r < H; r++) fft_exec_f32(&mine, cplx + 2 * (size_t)r * N, 0);
        free(mine.scratch);
      }
    } else {
      for (int r = 0; r < H; r++) fft_exec_f32(&pl, cplx + 2 * (size_t)r * N, 0);
    }
    plan_free_f32(&pl);
  }

  /* main:1189-1190 split, magnitude */
#pragma omp parallel for num_threads(nth) if (nth > 1)
  for (int r = 0; r < H; r++)
    for (int q = 0; q < N; q++) {
      const size_t i = (size_t)r * N + q;
      const float re = cplx[2 * i], im = cplx[2 * i + 1];
      magI[i] = sqrtf(re * re + im * im);
    }
  return 0;
}

int orc_frame_to_mag(const orc_params *p, const double *data_y_in,
                     const double *yb, const double *yp, const double *yd,
                     const double *win, const int32_t *idx, const double *frac,
                     const float *phase, float *magI, double *ylin_dbg) {
  frame_ws ws;
  if (ws_init(&ws, p)) {
    ws_free(&ws);
    return -1;
  }
  int rc = frame_to_mag_ws(p, &ws, data_y_in, yb, yp, yd, win, idx, frac, phase, magI, NULL, ylin_dbg);
  ws_free(&ws);
  return rc;
}

int orc_frame_to_mag_f64(const orc_params *p, const double *data_y_in,
                         const double *yb, const double *yp, const double *yd,
                         const double *win, const int32_t *idx, const double *frac,
                         const float *phase, double *magD, double *ylin_dbg) {
  frame_ws ws;
  if (!p->truth) return -3;
  if (ws_init(&ws, p)) {
    ws_free(&ws);
    return -1;
  }
  int rc = frame_to_mag_ws(p, &ws, data_y_in, yb, yp, yd, win, idx, frac, phase, NULL, magD, ylin_dbg);
  ws_free(&ws);
  return rc;
}

/* ------------------------------------------------------------------ A9 -- */
/* main:1195-1197 (accumulate) / sim:938-941 (copyTo) */
void orc_accumulate(const float *magI, int H, int N, int D, int copy_only,
                    double *acc) {
  for (int r = 0; r < H; r++)
    for (int d = 0; d < D; d++) {
      const double v = (double)magI[(size_t)r * N + d];
      if (copy_only)
        acc[(size_t)r * D + d] = v;
      else
        acc[(size_t)r * D + d] += v;
    }
}

static void accumulate_f64(const double *magD, int H, int N, int D, int copy_only, double *acc) {
  for (int r = 0; r < H; r++)
    for (int d = 0; d < D; d++) {
      const double v = magD[(size_t)r * N + d];
      if (copy_only)
        acc[(size_t)r * D + d] = v;
      else
        acc[(size_t)r * D + d] += v;
    }
}

/* ----------------------------------------------------------------- A10 -- */
/* main:1220-1240: transpose; /averagestoggle; += 1e-5 (sim: 1e-6, no divide);
 * log; 20*ln/2.303; rows 1 and 0 <- row 4. */
void orc_finish(const double *acc, int H, int D, int A, double eps,
                double *bscan, double *bscandb) {
  for (int d = 0; d < D; d++)
    for (int r = 0; r < H; r++) {
      double v = acc[(size_t)r * D + d];
      v = v / A;
      v += eps;
      if (bscan) bscan[(size_t)d * H + r] = v;
      if (bscandb) bscandb[(size_t)d * H + r] = 20.0 * log(v) / 2.303;
    }
  if (bscandb && D > 4) {
    memcpy(bscandb + (size_t)1 * H, bscandb + (size_t)4 * H, sizeof(double) * H);
    memcpy(bscandb + (size_t)0 * H, bscandb + (size_t)4 * H, sizeof(double) * H);
  }
}

/* -------------------------------------------------------------- driver -- */
static int process_u16_impl(const orc_params *p, int A, double eps, int copy_only,
                    const uint16_t *frames, int nframes, const double *yb,
                    const double *yp, const double *yd, const double *win,
                    const int32_t *idx, const double *frac, const float *phase,
                    double *out_mag_rowmajor, double *out_bscan,
                    double *out_db) {
  const int W = p->W, H = p->H, N = p->N, D = p->D;
  const size_t HW = (size_t)H * W, HD = (size_t)H * D;
  if (A < 1 || nframes % A) return -2;
  double *data_y = (double *)malloc(sizeof(double) * HW);
  float *magI = p->truth ? NULL : (float *)malloc(sizeof(float) * (size_t)H * N);
  double *magD = p->truth ? (double *)malloc(sizeof(double) * (size_t)H * N) : NULL;
  double *acc = (double *)malloc(sizeof(double) * HD);
  frame_ws ws;
  if (!data_y || (!magI && !magD) || !acc || ws_init(&ws, p)) return -1;
  int rc = 0;
  for (int g = 0; g < nframes / A && rc == 0; g++) {
    memset(acc, 0, sizeof(double) * HD); /* main:1482 */
    for (int a = 0; a < A && rc == 0; a++) {
      const uint16_t *fr = frames + (size_t)(g * A + a) * HW;
      for (size_t i = 0; i < HW; i++) data_y[i] = (double)fr[i]; /* main:987 */
      rc = frame_to_mag_ws(p, &ws, data_y, yb, yp, yd, win, idx, frac, phase, magI, magD, NULL);
      if (p->truth)
        accumulate_f64(magD, H, N, D, copy_only, acc);
      else
        orc_accumulate(magI, H, N, D, copy_only, acc);
    }
    /* sim:947-949: bscantransposed holds the LAST frame's magnitudes (copyTo) and is not divided */
    const int div = copy_only ? 1 : A;
    if (out_mag_rowmajor)
      for (size_t i = 0; i < HD; i++) out_mag_rowmajor[(size_t)g * HD + i] = acc[i] / div;
    orc_finish(acc, H, D, div, eps, out_bscan ? out_bscan + (size_t)g * HD : NULL,
               out_db ? out_db + (size_t)g * HD : NULL);
  }
  ws_free(&ws);
  free(data_y);
  free(magI);
  free(magD);
  free(acc);
  return rc;
}

int orc_process_u16(const orc_params *p, int A, double eps,
                    const uint16_t *frames, int nframes, const double *yb,
                    const double *yp, const double *yd, const double *win,
                    const int32_t *idx, const double *frac, const float *phase,
                    double *out_mag_rowmajor, double *out_bscan,
                    double *out_db) {
  return process_u16_impl(p, A, eps, 0, frames, nframes, yb, yp, yd, win, idx, frac, phase, out_mag_rowmajor, out_bscan, out_db);
}

/* BscanFFTsim.cpp with averages = A (sim:936-947): for indextemp < A every frame's cropped magnitudes are COPIED into
 * bscantransposed (the accumulate is commented out, sim:940-941); on the next frame the else branch emits the transpose of
 * what is there -- the magnitudes of frame A - 1 of the group -- plus 1e-6, undivided (sim:947-949), and that frame's own
 * magnitudes are dropped.  Groups of A frames here: the dropped (A + 1)-th frame of the reference's loop reaches no output. */
int orc_process_u16_sim(const orc_params *p, int A, double eps,
                        const uint16_t *frames, int nframes, const double *yb,
                        const double *yp, const double *yd, const double *win,
                        const int32_t *idx, const double *frac, const float *phase,
                        double *out_mag_rowmajor, double *out_bscan,
                        double *out_db) {
  return process_u16_impl(p, A, eps, 1, frames, nframes, yb, yp, yd, win, idx, frac, phase, out_mag_rowmajor, out_bscan, out_db);
}

/* ------------------------------------------------------- frame-source tail -- */
static int cmp_u16(const void *a, const void *b) {
  return (int)*(const uint16_t *)a - (int)*(const uint16_t *)b;
}

void orc_median_blur_u16(const uint16_t *src, uint16_t *dst, int w, int h, int n) {
  const int r = n / 2;
  uint16_t v[49 * 4];
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int c = 0;
      for (int dy = -r; dy <= r; dy++) {
        int yy = y + dy;
        yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
        for (int dx = -r; dx <= r; dx++) {
          int xx = x + dx;
          xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
          v[c++] = src[(size_t)yy * w + xx];
        }
      }
      qsort(v, (size_t)c, sizeof(uint16_t), cmp_u16);
      dst[(size_t)y * w + x] = v[c / 2];
    }
}

void orc_resize_area_u16(const uint16_t *src, uint16_t *dst, int w, int h, int binx, int biny) {
  const int ow = w / binx, oh = h / biny;
  const float scale = 1.f / (float)(binx * biny);
  for (int y = 0; y < oh; y++)
    for (int x = 0; x < ow; x++) {
      unsigned s = 0;
      for (int dy = 0; dy < biny; dy++)
        for (int dx = 0; dx < binx; dx++) s += src[(size_t)(y * biny + dy) * w + (x * binx + dx)];
      unsigned o;
      if (binx == 2 && biny == 2)
        o = (s + 2) >> 2;
      else
        o = (unsigned)nearbyintf((float)s * scale); /* cvRound: round half to even */
      dst[(size_t)y * ow + x] = (uint16_t)o;
    }
}
