/*
 * fdoct_oracle.h -- CPU restatement of the FD-OCT reconstruction block of
 * hn-88/FDOCT (BscanFFT.cpp / BscanFFTsim.cpp).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under fdoct_amd/ (the product) may
 * include, link or dlopen this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU
 * baseline.
 *
 * PARITY UNPINNED: the reference ships no expected-output vectors, and its
 * arithmetic lives in OpenCV (cv::dft, cv::normalize, ...) which is absent
 * from this image, so the reference cannot be built or run here.  This oracle
 * follows the reference source line by line (citations on every function) and
 * is pinned only by (1) the reference's input fixtures imgi.png / backg.png,
 * (2) an analytic known-answer test on those fixtures (reflector depth ->
 * peak bin), (3) numpy/scipy cross-checks of every stage.
 *
 * All file:line citations are into /root/reference ("main" = BscanFFT.cpp,
 * "sim" = BscanFFTsim.cpp, "dark" = BscanDark.cpp).
 */
#ifndef FDOCT_ORACLE_H
#define FDOCT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A0: one-time k tables, main:615-698 (sim:451-534).
 * idx[N] (nearestkindex), frac[N] (fractionalk); optional debug outputs
 * k[M*W], klinear[N], diffk[M*W] may be NULL.  Returns 0. */
int orc_tables(int W, int M, int N, double lambdamin, double lambdamax,
               int32_t *idx, double *frac, double *k, double *klinear,
               double *diffk);

/* A1: modified Bartlett-Hann window, main:936-944 (sim:765-773). */
void orc_barthann(int W, double *win);

/* A11: normalizerows main:88-97; cv::normalize(NORM_MINMAX) semantics. */
void orc_normalize_minmax(double *y, size_t n, double lo, double hi);
void orc_normalizerows(double *y, int H, int W, double lo, double hi);

/* A11: smoothmovavg main:247-304. */
void orc_smoothmovavg(const double *src, double *dst, int H, int W, int n);

/* A4: zeropadrowwise main:180-245; bandpass != 0 adds dark:218-236. */
void orc_zeropadrowwise(const double *y, int H, int W, int M, int bandpass,
                        double *out /* H x (M*W) */);

/* A7: batched row DFT, cv::dft(DFT_ROWS [|DFT_INVERSE] [|DFT_SCALE]) model.
 * data: H rows of N interleaved (re,im) pairs.  f32 = float arithmetic with
 * twiddles rounded from double; f64 = double arithmetic (error bounding). */
void orc_dft_rows_f32(float *data, int H, int N, int inverse, int scale);
void orc_dft_rows_f64(double *data, int H, int N, int inverse, int scale);

typedef struct {
  int W, H, N, D, M;
  int rowwisenormalize; /* main:1126 */
  int donotnormalize;   /* main:1128; sim:845 always normalises => 0 */
  int movavgn;          /* main:990 */
  int bandpass;         /* dark:218 */
  int threads;          /* 1 = reference-faithful single thread; >1 = OpenMP over rows */
  int truth;            /* 0 = the reference's arithmetic (fp64 elementwise, fp32 zero-pad DFTs, fp32 narrowing + cv::dft +
                         * magnitude).  1 = TRUTH: the same operations in the same order with every float step (main:209-242,
                         * 1181, 1185, 1190) carried out in double -- the exact value of the reference's MATHEMATICS on the same
                         * inputs and tables, to ~1e-15.  It is the adjudicator of the parity tests: a result within 0.5 x the
                         * tolerance of truth is within the tolerance of ANY correctly rounded float cv::dft of the same chain,
                         * whichever radix decomposition OpenCV picks -- the part of "parity unpinned" arithmetic can close. */
} orc_params;

/* A2..A8 for one frame: main:1123-1190 (sim:842-933).
 * data_y: H x W doubles (the frame after convertTo(CV_64F), main:987).
 * yb, yp: H x W doubles (background, pi frame); yd: H x W dark frame or NULL
 * (dark:1269).  win[W], idx[N], frac[N] from A0/A1.
 * phase: N interleaved (cos,sin) float pairs or NULL (A6', extension).
 * magI: H x N float output (main:1190).  ylin_dbg: H x N double or NULL. */
int orc_frame_to_mag(const orc_params *p, const double *data_y,
                     const double *yb, const double *yp, const double *yd,
                     const double *win, const int32_t *idx, const double *frac,
                     const float *phase, float *magI, double *ylin_dbg);

/* The same in truth mode (p->truth must be 1): magD is H x N doubles. */
int orc_frame_to_mag_f64(const orc_params *p, const double *data_y,
                         const double *yb, const double *yp, const double *yd,
                         const double *win, const int32_t *idx, const double *frac,
                         const float *phase, double *magD, double *ylin_dbg);

/* A9: crop to D, convert to f64, accumulate (main:1195-1197) or copy
 * (sim:938-941) into acc (H x D). */
void orc_accumulate(const float *magI, int H, int N, int D, int copy_only,
                    double *acc);

/* A10: main:1220-1240 (sim:947-955): bscan = transpose(acc)/A + eps,
 * bscandb = 20*ln(bscan)/2.303, rows 0,1 <- row 4.  Outputs D x H. */
void orc_finish(const double *acc, int H, int D, int A, double eps,
                double *bscan, double *bscandb);

/* Convenience driver used for golden vectors and the CPU baseline timing:
 * nframes frames of u16 (H x W each) -> groups of A averaged outputs.
 * frames: nframes*H*W uint16.  out_bscan/out_db: (nframes/A) x D x H doubles
 * (either may be NULL).  out_mag_rowmajor: (nframes/A) x H x D doubles
 * (averaged linear magnitude before transpose/eps; may be NULL). */
int orc_process_u16(const orc_params *p, int A, double eps,
                    const uint16_t *frames, int nframes, const double *yb,
                    const double *yp, const double *yd, const double *win,
                    const int32_t *idx, const double *frac, const float *phase,
                    double *out_mag_rowmajor, double *out_bscan,
                    double *out_db);

/* The same for BscanFFTsim.cpp (sim:936-947): copyTo instead of accumulate, no division -- every group of A frames
 * yields its LAST frame's magnitudes. */
int orc_process_u16_sim(const orc_params *p, int A, double eps,
                        const uint16_t *frames, int nframes, const double *yb,
                        const double *yp, const double *yd, const double *win,
                        const int32_t *idx, const double *frac, const float *phase,
                        double *out_mag_rowmajor, double *out_bscan,
                        double *out_db);

/* Frame-source tail, on u16 samples (u8 data embeds exactly):
 * cv::medianBlur(src, dst, n) main:953-956 -- n x n median, n odd, BORDER_REPLICATE;
 * cv::resize(.., 1/binx, 1/biny, INTER_AREA) main:958 for integer factors -- box sum, then
 * (s+2)>>2 for 2x2 (OpenCV's vectorised 8u/16u path) or round-half-even of s*(1.f/area). */
void orc_median_blur_u16(const uint16_t *src, uint16_t *dst, int w, int h, int n);
void orc_resize_area_u16(const uint16_t *src, uint16_t *dst, int w, int h, int binx, int biny);

/* Display post-chain, main:1242-1255: max(db, thr); optional (5,5) <- 50; cv::normalize
 * NORM_MINMAX to [0,1] (dst = src*s + (0 - min*s), s = 1/(max-min), 0 if the range is < DBL_EPSILON);
 * convertTo(CV_8U, 255.0) = saturate(round-half-even(v*255.0)), all in double.  (OpenCV builds whose SIMD
 * path converts 64f->8u through float may differ by 1 LSB at rounding ties; unpinned like the rest.) */
void orc_display_u8(const double *db, int rows, int cols, double thr, int clampupper, uint8_t *gray);
/* applyColorMap main:1284 as a table look-up: bgr[3i..] = lut[3*gray[i]..]. */
void orc_apply_lut(const uint8_t *gray, size_t n, const uint8_t *lut_bgr256, uint8_t *bgr);
/* J0 lock-in main:1227-1230,1260-1261: 20*ln(max(bscan-jscan,0)+0.001)/2.303. */
void orc_lockin_db(const double *bscan, const double *jscan, size_t n, double *out);

const char *orc_version(void);

#ifdef __cplusplus
}
#endif
#endif
