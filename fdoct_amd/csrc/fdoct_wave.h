// fdoct_wave.h -- interface of the wave-per-row kernels (fdoct_wave.hip): the acquisition configurations the reference
// ships (build/*.ini: numfftpoints 2560 / 2880 / 640, zero-pad multiplier 4 or 1), one 64-lane wave per A-scan.
#pragma once
#ifndef __HIPCC_RTC__  // (device part also compiled at run time, fdoct_jit.cpp: no system headers there)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace fdoct {

// Radix plan of an in-LDS Stockham DFT run by ONE wave: a fused 20 where it fits, then odd radices first (the first pass writes butterfly j's outputs R
// apart; an odd R keeps those stores spread over the LDS banks), then 8s, then what is left of the power of two.
// One definition for the kernel (compile time) and the host (twiddle tables).
struct WavePlan {
  int npass = 0;
  int R[12] = {};
  int Ns[12] = {};  // product of the radices of the earlier passes
};
constexpr WavePlan wave_plan(int n) {
  WavePlan p{};
  const int n_all = n;
  int ns = 1;
  auto push = [&](int r) {
    p.R[p.npass] = r;
    p.Ns[p.npass] = ns;
    p.npass++;
    ns *= r;
  };
  // a radix-5 pass followed by a radix-4 pass stays inside one lane when the length is a multiple of 64*20: fused
  // into one in-register 20-point pass (fft_reg20), no LDS round trip between them
  if (n % 1280 == 0) { push(20); n /= 20; }
  // a single factor 3 next to a 5: one in-register 3 x 5 pass instead of a radix-5 and a radix-3 round trip (960 = 15 8 8)
#ifndef FDOCT_WAVE_R15_MIN
#define FDOCT_WAVE_R15_MIN 240   // smallest transform that takes a radix-15 pass: 120 = 15 8 keeps eight lanes busy and loses 2.5 % on 240 x 4 rows (tools/bench_plans.py)
#endif
  while (n % 15 == 0 && n % 9 != 0 && n_all >= FDOCT_WAVE_R15_MIN) { push(15); n /= 15; }
  while (n % 5 == 0) { push(5); n /= 5; }
#ifndef FDOCT_WAVE_R9_MIN
#define FDOCT_WAVE_R9_MIN 0   // smallest transform that takes a radix-9 pass: it pays on every built-in shape with a factor 9 (+ 2.6 ... 12 %, tools/bench_plans.py)
#endif
  while (n % 9 == 0 && n_all >= FDOCT_WAVE_R9_MIN) { push(9); n /= 9; }   // (one in-register 3 x 3 pass instead of two radix-3 round trips: 1440 = 5 9 8 4)
  while (n % 3 == 0) { push(3); n /= 3; }
  // the power of two that is left: radix-8 passes, and as many radix-16 ones as make the pass count smaller (640 = 5 16 8
  // instead of 5 8 8 2, 2560 = 20 16 8 instead of 20 8 8 2: + 5 ... 24 % on such shapes; a tie goes to the 8s, whose
  // butterflies hold half the registers).  Not below 512 points: a radix-16 pass of an 80-point transform keeps 5 lanes busy
  // with 16 values each and costs more than the round trip it saves (BscanFFT.ini - 6 %, measured).
  int bits = 0;
  for (int m = n; m > 1 && m % 2 == 0; m /= 2) bits++;
  int n16 = 0, fewest = 1 << 20;
#ifndef FDOCT_WAVE_R16_MIN
#define FDOCT_WAVE_R16_MIN 512   // (128 instead: 256 / 512 x 4 rows - 4 ... 7 %; none at all: the built-in shapes +- 1 %, 640 x 4 -> 5120 - 20 %; tools/bench_plans.py)
#endif
  for (int a16 = 0; 4 * a16 <= bits && (a16 == 0 || n_all >= FDOCT_WAVE_R16_MIN); a16++) {
    const int passes = a16 + (bits - 4 * a16 + 2) / 3;
    if (passes < fewest) { fewest = passes; n16 = a16; }
  }
  for (int i = 0; i < n16; i++) { push(16); n /= 16; }
  while (n % 8 == 0) { push(8); n /= 8; }
  if (n % 4 == 0) { push(4); n /= 4; }
  if (n % 2 == 0) { push(2); n /= 2; }
  if (n != 1) p.npass = -1;  // another prime factor: no plan
  return p;
}
// entries of the per-pass twiddle tables of one transform: pass p (Ns > 1) holds exp(+2 pi i k / (Ns R)), k < Ns
constexpr int wave_plan_twiddles(int n) {
  const WavePlan p = wave_plan(n);
  int c = 0;
  for (int i = 0; i < p.npass; i++)
    if (p.Ns[i] > 1) c += p.Ns[i];
  return c;
}

// A transform whose plan starts with the fused radix 20 keeps its first pass's output padded (22 complex values per butterfly
// instead of 20: wave_pass, OPAD / IPAD), so its buffer holds n + n / 10 values between the first two passes.  One rule for the
// kernel's passes, the buffer length (kernel and host) and the run-time compiler.
#ifndef FDOCT_WAVE_R20PAD
#define FDOCT_WAVE_R20PAD 1
#endif
constexpr bool wave_r20_padded(int n) {
  const WavePlan p = wave_plan(n);
  return FDOCT_WAVE_R20PAD && p.npass >= 2 && p.R[0] == 20 && p.Ns[1] == 20 && (n / p.R[1]) % 20 == 0;
}
constexpr int wave_fft_extent(int n) { return wave_r20_padded(n) ? n + n / 10 : n; }

// The slope step gives lane l the M*W/64 consecutive upsampled samples l*SPL ..: with SPL a multiple of 8 the lanes' 16-byte
// LDS accesses start 8 banks apart and collide four ways (a third of the LDS time of the 640 x 4 shapes).  Four pad floats
// per lane make the stride 4 * odd: sample s of the upsampled row lives at s + 4 (s / SPL).  Kernel and host (LDS sizes) share
// the rule.
constexpr int wave_row_pad_floats(int W, int M) { return (M > 1 && (M * W) % 64 == 0 && ((M * W / 64) % 8) == 0) ? 4 : 0; }

// The compiled shapes: {width after binning, zero-pad multiplier, numfftpoints}.
//   160 x4 -> 2560   build/BscanFFT.ini (320-wide ROI, 2x2 binning)
//   640 x4 -> 2560   build/BscanFFTspin.ini, BscanFFTpeak.ini, BscanDark.ini (1280-wide cameras, 2x2 binning)
//   720 x4 -> 2880   build/BscanFFTspinj.ini (720-wide, no binning)
//   640 x1 ->  640   build/BscanFFTwebcam.ini
//   320 x4 -> 2560   1280-wide cameras at 4x4 binning
#ifndef FDOCT_WAVE_SHAPES
#define FDOCT_WAVE_SHAPES(X) X(160, 4, 2560) X(640, 4, 2560) X(720, 4, 2880) X(640, 1, 640) X(320, 4, 2560)
#endif
// Other regions of interest / bin factors an operator may type into the ini (build/BscanFFT.ini:9-12 width, 25-26 binvalue)
// with the shipped numfftpoints 2560 and zero-pad multiplier 4: every row width from 192 to 1280 that is a multiple of 16
// and factors into 2, 3 and 5.  Compiled for 8/16-bit samples and numdisplaypoints <= 512 (two kernels per shape, in two
// translation units); other widths, sample types and depths stay on the workgroup-per-row kernel.
#ifndef FDOCT_WAVE_SHAPES_EXTRA_1
#define FDOCT_WAVE_SHAPES_EXTRA_1(X) \
  X(192, 4, 2560) X(240, 4, 2560) X(256, 4, 2560) X(288, 4, 2560) X(384, 4, 2560) X(400, 4, 2560) X(432, 4, 2560) X(480, 4, 2560) \
  X(512, 4, 2560) X(576, 4, 2560)
#endif
#ifndef FDOCT_WAVE_SHAPES_EXTRA_2
#define FDOCT_WAVE_SHAPES_EXTRA_2(X) \
  X(768, 4, 2560) X(800, 4, 2560) X(864, 4, 2560) X(960, 4, 2560) X(1024, 4, 2560) X(1152, 4, 2560) X(1200, 4, 2560) X(1280, 4, 2560)
#endif
#define FDOCT_WAVE_SHAPES_EXTRA(X) FDOCT_WAVE_SHAPES_EXTRA_1(X) FDOCT_WAVE_SHAPES_EXTRA_2(X)

// Bytes of one wave's private row buffer: L complex values (L = max(N/2, padded M W / 2)) and the zero slot; when 64 does
// not divide M W the slope step's last lanes read up to 63 floats past the row (values it does not use), so the buffer
// carries that much slack and no read leaves the workgroup's allocation.
constexpr int wave_private_bytes(int L, int MW) { return (((L + 2) * 8 + 15) & ~15) + ((MW % 64) != 0 ? 256 : 0); }

// Shapes whose register budget (8 waves per workgroup, 256 registers) has room for the two row-invariant tables the slope
// step and the gather read for every input row -- fractionalk by sample (M W / 64 floats per lane) and the packed gather
// sources (N / 128 words per lane): they stay in registers there and 15 KB of LDS reads per row go away (the 640 x 4 -> 2560
// kernels use 181 of 256 registers without them; the 720 x 4 -> 2880 ones have none to spare).
constexpr bool wave_resident_tables(int W, int M, int N) {
#ifdef FDOCT_WAVE_NO_RESG  // tuning: tables from LDS everywhere
  return false;
#else
  return M > 1 && (M * W) % 64 == 0 && M * W >= 2560 && M * W <= 2560 && N <= 2560;
#endif
}

// acquisition options a wave_kernel instantiation is compiled with (template parameter OPT)
// (defined below; wave_rows_of reads two of them)
#define FDOCT_WAVE_OPT_CPLX 64
#define FDOCT_WAVE_OPT_DEEP 128
// Rows a wave works on side by side: with -DFDOCT_WAVE_ROWS2=1, 2 on the short zero-padded real rows -- M W <= 1280 upsampled
// samples, a whole number per lane: BscanFFT.ini's 160 x 4, 320 x 4, ... -- whose small transforms leave most lanes idle; 1
// everywhere else.  One definition for kernel, host (LDS per wave, waves per workgroup, rows per slot) and run-time compiler.
// OFF by default: built and measured in round 6 after being costed in rounds 4 and 5 (EXPERIMENTS.md section 5): 11 % fewer
// instructions per row, and 285 against 374 M input A-scans/s on BscanFFT.ini -- six waves of 201 registers hide the LDS latency
// of the dependent phases worse than twelve of 140 (profiles/r06_wave_rows2_ab.txt).  Kept as the measured prototype.
#ifndef FDOCT_WAVE_ROWS2
#define FDOCT_WAVE_ROWS2 0
#endif
constexpr int wave_rows_of(int W, int M, int N, int opt) {
  return (FDOCT_WAVE_ROWS2 && M > 1 && M * W <= 1280 && (M * W) % 64 == 0 && !(opt & (FDOCT_WAVE_OPT_CPLX | FDOCT_WAVE_OPT_DEEP | 4 /* FDOCT_WAVE_OPT_BANDPASS: the one-row body's double evaluation */)) && N % 2 == 0) ? 2 : 1;
}

#define FDOCT_WAVE_OPT_PI 1        // data_yp: pi-shifted / J0 frame subtracted before the division (main:1132)
#define FDOCT_WAVE_OPT_DARK 2      // data_yd: dark frame subtracted first (BscanDark.cpp:1269)
#define FDOCT_WAVE_OPT_BANDPASS 4  // band-pass inside the zero-pad stage (BscanDark.cpp:218-236)
#define FDOCT_WAVE_OPT_ROWNORM 8   // normalizerows: every row min-max normalised to [0, 1] (main:88-97, 1126-1127)
#define FDOCT_WAVE_OPT_BIN2 32      // the frames are RAW camera frames (2 H x 2 W): 2 x 2 software binning (main:958) inside the loads
#define FDOCT_WAVE_OPT_FRAMENORM 16  // whole-frame min-max normalisation to [0, 1] (main:1128-1129, sim:845); min/max from a pre-pass
//      FDOCT_WAVE_OPT_CPLX 64       // dispersion phase (fdoct_set_dispersion_phase; wangOCTrec4.m:130-131, 169): data_ylin[q] times a unit phasor, then the
                                     // FULL numfftpoints-point complex transform (no real-input untangle), any numdisplaypoints <= numfftpoints
//      FDOCT_WAVE_OPT_DEEP 128      // real rows displayed beyond numfftpoints / 2: bins above it mirror (|X[b]| = |X[N - b]|), bin N/2 from Z[0]
// the final transform's length in complex points: the whole row for complex rows, half of it for real ones
constexpr int wave_final_points(int N, int opt) { return (opt & FDOCT_WAVE_OPT_CPLX) ? N : N / 2; }

// The depth bound a kernel is compiled for, next to TD (depth bins per lane): numdisplaypoints rounded up to a whole output block
// of the final transform's last pass (block r = outputs r Ns .. (r + 1) Ns - 1).  Blocks no depth <= the bound reads are never
// stored, so their arithmetic is not compiled either (wave_pass, DK): BscanFFT.ini shows 320 of 1280 bins, four of the last
// radix-8 pass's eight blocks feed nothing.  One definition for kernel, launch and run-time compiler.
constexpr int wave_depth_bound(int N, int opt, int D) {
  const int nc = wave_final_points(N, opt);
  const WavePlan p = wave_plan(nc);
  if (p.npass < 1 || D >= nc) return nc;
  const int ns = p.Ns[p.npass - 1];
  const int dk = ((D + ns - 1) / ns) * ns;
  return dk < nc ? dk : nc;
}

struct WaveArgs {
  const void* frames;
  long long pitch_bytes;
  long long total_out_rows;
  int dtype;  // FDOCT_K_*
  int H, D, A;
  const float* ib;  // [W] or [H*W] 1/background
  const float* il;  // its low word, indexed like ib (fdoct_capi.cpp::reciprocal_words)
  int ib_2d;
  const float* win;      // [W] window
  const float* win_lo;   // [W] what the float window leaves of the double one (OPT & FDOCT_WAVE_OPT_BANDPASS: the row is formed in double)
  const float *yp_lo, *yd_lo;  // the same of the pi and dark frames (laid out like yp, yd)
  const float* g;        // [M*W] fractionalk by sample (0 past numfftpoints)
  const uint32_t* gidx;  // [N/2] packed float indices of the sources of data_ylin[2n] (low half) and [2n+1]; M*W = the zero slot
                         // (OPT & FDOCT_WAVE_OPT_CPLX: [N], the source of data_ylin[n] in the low half)
  const float2* tw;      // twiddle blob (see build_wave_tables in fdoct_capi.cpp); offsets in float2 units below
  int tw_count;
  int off_nc, off_lh, off_wh;      // per-pass tables of the N/2-, M*W/2- and W/2-point transforms
  int off_tww, off_twmw, off_twn;  // exp(+2 pi i k/W), k < W/2; exp(+2 pi i k/(M W)), k < W/2; exp(+2 pi i k/N), k < D
  int dcmask;
  float inv_A, eps, db_scale;
  float* out_mag;
  float* out_db;
  const float* yp;  // [W] or [H*W], OPT & FDOCT_WAVE_OPT_PI
  const float* yd;  // [W] or [H*W], OPT & FDOCT_WAVE_OPT_DARK
  int yp_2d, yd_2d;
  const void* minmax;  // float2 (min, max) per input frame, OPT & FDOCT_WAVE_OPT_FRAMENORM
  const float2* phase;  // [N] (cos, sin), OPT & FDOCT_WAVE_OPT_CPLX
  unsigned long long* probe;  // measurement builds (-DFDOCT_WAVE_PROBE): cycles per phase of one wave; null otherwise
};

#ifndef __HIPCC_RTC__
bool wave_shape_compiled(int W, int M, int N);   // one of FDOCT_WAVE_SHAPES: every sample type, any numdisplaypoints <= N/2
// a wave-per-row kernel exists for this shape, sample type (FDOCT_K_*) and depth (FDOCT_WAVE_SHAPES or _EXTRA)
bool wave_kernel_available(int W, int M, int N, int dtype, int D);
int wave_max_waves(int W, int M, int N, int opt = 0);  // waves per workgroup the shape is compiled for (register budget)
// LDS bytes: the tables every wave of a workgroup shares, and the private buffer of one wave
size_t wave_shared_lds_bytes(int tw_count, int W, int M, int N, bool ib_2d, int opt = 0);
size_t wave_private_lds_bytes(int W, int M, int N, int opt = 0);   // per WAVE: wave_rows_of row buffers
int wave_rows_per_wave(int W, int M, int N, int opt = 0);           // output rows a wave's slot names (wave_rows_of)
hipError_t launch_wave(int W, int M, int N, const WaveArgs& a, int grid, int waves, size_t lds, hipStream_t st);
#endif  // !__HIPCC_RTC__

}  // namespace fdoct
