// fdoct_wave.hip -- wave-per-row kernels for the acquisition configurations the reference ships (gfx950).
//
// Every build/*.ini of the reference uses a numfftpoints that is not a power of two (2560 = 2^9*5, 2880 = 2^6*3^2*5, 640)
// and all but the webcam one the x4 zero-pad upsampling (zeropadrowwise, BscanFFT.cpp:180-245), so none of them fits the
// power-of-two fused plans of fdoct_kernels.hip.  Rows are short there (160 .. 720 samples), which makes the
// workgroup-per-row kernel of fdoct_generic.hip a chain of ~25 workgroup barriers per A-scan with a handful of
// instructions between them.  Here ONE 64-lane wave owns an A-scan from the camera samples to the depth profile:
//   * no workgroup barrier at all: a wave's LDS operations execute in program order, so every hand-off inside the
//     row is a compiler fence;
//   * the row lives in one private LDS buffer of max(N/2, M*W/2) complex values; each Stockham pass reads all of its
//     butterflies' inputs into registers, then writes the outputs back IN PLACE (no ping-pong buffer), so 11-12 rows are
//     in flight per CU (3 waves per SIMD);
//   * radices, strides, trip counts and predicates of every pass are compile-time constants of the shape
//     (FDOCT_WAVE_SHAPES in fdoct_wave.h), per-pass twiddle tables are contiguous in LDS (one read per butterfly, the
//     powers by products), the resample tables are shared by the workgroup's waves;
//   * the lambda->k gather feeds the first pass's registers directly, the last pass stores only the bins the real-input
//     untangle will read (numdisplaypoints <= N/2 of them and their mirror partners).
// Same arithmetic as the any-option kernels: f32, row mean in f64, float DFTs; M > 1 reproduces cv::dft's DFT_REAL_OUTPUT
// reading (Nyquist bin dropped, imaginary part of bin 0 ignored) exactly as fdoct_generic.hip does, at half length.
// Scope: rows whose lengths factor into 2, 3 and 5.  The instantiations of this file are the plain acquisition set-up (real rows,
// numdisplaypoints <= N/2, no pi/dark frame, no normalisation, no band-pass: OPT = 0) of the shapes in fdoct_wave.h; the options
// -- those, the dispersion phase (complex rows) and a display beyond N/2 -- and every other shape are instantiated at run time
// from the same source (fdoct_wave_dev.h, fdoct_jit.cpp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fdoct_fft_reg.h"
#include "fdoct_kernels.h"
#include "fdoct_wave.h"

#include "fdoct_wave_dev.h"

namespace fdoct {

// ---------------------------------------------------------------- dispatch --
#ifndef FDOCT_WAVE_EXTRA_TU
int wave_max_waves(int W, int M, int N, int opt) { return wave_block_of(W, M, N, opt) / 64; }

bool wave_shape_compiled(int W, int M, int N) {
#define FDOCT_WAVE_HAS(W_, M_, N_) \
  if (W == W_ && M == M_ && N == N_) return true;
  FDOCT_WAVE_SHAPES(FDOCT_WAVE_HAS)
#undef FDOCT_WAVE_HAS
  return false;
}

static bool wave_shape_extra(int W, int M, int N) {
#define FDOCT_WAVE_HAS(W_, M_, N_) \
  if (W == W_ && M == M_ && N == N_) return true;
  FDOCT_WAVE_SHAPES_EXTRA(FDOCT_WAVE_HAS)
#undef FDOCT_WAVE_HAS
  return false;
}

bool wave_kernel_available(int W, int M, int N, int dtype, int D) {
  if (wave_shape_compiled(W, M, N)) return D <= N / 2;
  return wave_shape_extra(W, M, N) && (dtype == FDOCT_K_U8 || dtype == FDOCT_K_U16) && D <= 512 && D <= N / 2;
}
#endif  // !FDOCT_WAVE_EXTRA_TU

#ifndef FDOCT_WAVE_EXTRA_TU
size_t wave_private_lds_bytes(int W, int M, int N, int opt) {
  const int L = imax(imax(wave_fft_extent(wave_final_points(N, opt)), (M * W + 64 * wave_row_pad_floats(W, M)) / 2), M > 1 ? wave_fft_extent(M * W / 2) : 0);   // as the kernel's L
  return (size_t)wave_private_bytes(L, M * W) * (size_t)wave_rows_of(W, M, N, opt);   // (a wave of a two-row shape holds two row buffers)
}
int wave_rows_per_wave(int W, int M, int N, int opt) { return wave_rows_of(W, M, N, opt); }

size_t wave_shared_lds_bytes(int tw_count, int W, int M, int N, bool ib_2d, int opt) {
  const size_t words = (size_t)tw_count * 2 + wave_final_points(N, opt) + (size_t)M * W + 64 * wave_row_pad_floats(W, M) + W + (ib_2d ? 0 : 2 * W) +
                       ((opt & FDOCT_WAVE_OPT_CPLX) ? 2 * (size_t)N : 0);  // as the kernel lays them out
  return (words * 4 + 15) & ~(size_t)15;
}

// EXTRA shapes (FDOCT_WAVE_SHAPES_EXTRA) are compiled for the cameras' integer samples and numdisplaypoints <= 512 only: two
// kernels per shape instead of six.
#endif  // !FDOCT_WAVE_EXTRA_TU

// One LdsGrant per KERNEL (the kernel is a non-type template argument: every instantiation has its own static; the kernels of
// a shape share one function-pointer type, so a generic lambda over `auto k` would share one table -- ADVICE r4).
template <auto K>
static hipError_t launch_wave_one(const WaveArgs& a, int grid, int waves, size_t lds, hipStream_t st) {
  static LdsGrant grant;
  if (hipError_t e = grant.ensure(K, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(K, dim3(grid), dim3(64 * waves), lds, st, a);
  return hipGetLastError();
}
#define FDOCT_WAVE_GO(...) launch_wave_one<__VA_ARGS__>(a, grid, waves, lds, st)

template <int W, int M, int N, bool EXTRA = false>
static hipError_t launch_wave_typed(const WaveArgs& a, int grid, int waves, size_t lds, hipStream_t st) {
  constexpr int TDF = (N / 2 + 63) / 64;  // any numdisplaypoints <= N/2
  constexpr int TDS = TDF < 8 ? TDF : 8;  // numdisplaypoints <= 512
  const bool small = a.D <= 64 * TDS;
  if constexpr (EXTRA) {
    if (!small) return hipErrorNotSupported;
    switch (a.dtype) {
      case FDOCT_K_U8: return FDOCT_WAVE_GO(wave_kernel<W, M, N, uint8_t, TDS>);
      case FDOCT_K_U16: return FDOCT_WAVE_GO(wave_kernel<W, M, N, uint16_t, TDS>);
      default: return hipErrorNotSupported;
    }
  }
  // The shipped configurations show a quarter of the half transform or less (320 of 1280 bins, 360 of 1440): a third kernel
  // of the primary shapes is compiled for that depth, without the blocks of the last pass it never reads (wave_depth_bound).
  constexpr int DKQ = wave_depth_bound(N, 0, N / 8), TDQ = (DKQ + 63) / 64;
  if constexpr (TDF > 8 && DKQ < N / 2) {
    if (a.D <= DKQ) {
      switch (a.dtype) {
        case FDOCT_K_U8: return FDOCT_WAVE_GO(wave_kernel<W, M, N, uint8_t, TDQ, 0, DKQ>);
        case FDOCT_K_U16: return FDOCT_WAVE_GO(wave_kernel<W, M, N, uint16_t, TDQ, 0, DKQ>);
        case FDOCT_K_F32: return FDOCT_WAVE_GO(wave_kernel<W, M, N, float, TDQ, 0, DKQ>);
        default: return hipErrorInvalidValue;
      }
    }
  }
  switch (a.dtype) {
    case FDOCT_K_U8: return small ? FDOCT_WAVE_GO(wave_kernel<W, M, N, uint8_t, TDS>) : FDOCT_WAVE_GO(wave_kernel<W, M, N, uint8_t, TDF>);
    case FDOCT_K_U16: return small ? FDOCT_WAVE_GO(wave_kernel<W, M, N, uint16_t, TDS>) : FDOCT_WAVE_GO(wave_kernel<W, M, N, uint16_t, TDF>);
    case FDOCT_K_F32: return small ? FDOCT_WAVE_GO(wave_kernel<W, M, N, float, TDS>) : FDOCT_WAVE_GO(wave_kernel<W, M, N, float, TDF>);
    default: return hipErrorInvalidValue;
  }
}

#ifdef FDOCT_WAVE_EXTRA_TU
// the extra shapes, one translation unit per part of the list (FDOCT_WAVE_EXTRA_TU = 1, 2)
#if FDOCT_WAVE_EXTRA_TU == 1
#define FDOCT_WAVE_EXTRA_PART FDOCT_WAVE_SHAPES_EXTRA_1
hipError_t launch_wave_extra1(int W, int M, int N, const WaveArgs& a, int grid, int waves, size_t lds, hipStream_t st) {
#else
#define FDOCT_WAVE_EXTRA_PART FDOCT_WAVE_SHAPES_EXTRA_2
hipError_t launch_wave_extra2(int W, int M, int N, const WaveArgs& a, int grid, int waves, size_t lds, hipStream_t st) {
#endif
#define FDOCT_WAVE_CASE(W_, M_, N_) \
  if (W == W_ && M == M_ && N == N_) return launch_wave_typed<W_, M_, N_, true>(a, grid, waves, lds, st);
  FDOCT_WAVE_EXTRA_PART(FDOCT_WAVE_CASE)
#undef FDOCT_WAVE_CASE
  return hipErrorNotSupported;
}
#else
hipError_t launch_wave_extra1(int W, int M, int N, const WaveArgs& a, int grid, int waves, size_t lds, hipStream_t st);
hipError_t launch_wave_extra2(int W, int M, int N, const WaveArgs& a, int grid, int waves, size_t lds, hipStream_t st);

hipError_t launch_wave(int W, int M, int N, const WaveArgs& a, int grid, int waves, size_t lds, hipStream_t st) {
  if (waves < 1 || waves > wave_max_waves(W, M, N)) return hipErrorInvalidValue;
#define FDOCT_WAVE_CASE(W_, M_, N_) \
  if (W == W_ && M == M_ && N == N_) return launch_wave_typed<W_, M_, N_>(a, grid, waves, lds, st);
  FDOCT_WAVE_SHAPES(FDOCT_WAVE_CASE)
#undef FDOCT_WAVE_CASE
  const hipError_t e = launch_wave_extra1(W, M, N, a, grid, waves, lds, st);
  return e != hipErrorNotSupported ? e : launch_wave_extra2(W, M, N, a, grid, waves, lds, st);
}
#endif

}  // namespace fdoct
