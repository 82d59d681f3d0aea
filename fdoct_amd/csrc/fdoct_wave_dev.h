// fdoct_wave_dev.h -- device code of the wave-per-row kernels (see fdoct_wave.hip for the design).  Compiled twice: by hipcc
// into the library for the shapes of FDOCT_WAVE_SHAPES*, and at run time by hipRTC (fdoct_jit.cpp) for any other shape a
// handle is created with -- hence no host code and no system headers here (__HIPCC_RTC__: the runtime compiler brings
// its own HIP declarations).
#pragma once
#include "fdoct_fft_reg.h"
#include "fdoct_wave.h"

namespace fdoct {

namespace {

// Register budget per shape (measured both ways on every shape, `tools/bench_generic.py`): rows whose upsampled length
// reaches 2560 samples (40 samples per lane in the slope step, two 1280- or 1440-point transforms) spill at 168
// registers and run faster with 8 waves per workgroup and 256 registers; the short rows (160 / 320 x4, 640 x1) are
// faster with 12 waves at 168.
// (The short zero-padded rows use 139-143 registers: 14 waves would still fit the LDS, but a workgroup of 14 puts four waves on
// two of the SIMDs, i.e. a 128-register budget, and the spills cost more than the extra waves give: 2.9e8 against 3.0e8 on
// BscanFFT.ini.  The webcam shape -- no zero-pad stage, 102 registers, a 2.6 KB buffer -- runs 16 waves: +3 %.)
// (Complex rows gather a whole numfftpoints-point transform into registers -- twice the real rows' -- and their buffers are
// twice as large, so the LDS holds few of them anyway: 8 waves, 256 registers.)
#ifndef FDOCT_WAVE_BLOCK_SHORT
#define FDOCT_WAVE_BLOCK_SHORT 768   // threads per workgroup of the short zero-padded rows (160 / 320 x 4 ...)
#endif
constexpr int wave_block_of(int w, int m, int n, int opt = 0) {
  if (wave_rows_of(w, m, n, opt) == 2) return FDOCT_WAVE_BLOCK_SHORT / 2;   // two rows per wave: half the waves hold the same rows
  // (the band-pass forms the row and a few of its spectral bins in double: 44 spilled registers at 168 on the 160 x 4 shape)
  return ((opt & (FDOCT_WAVE_OPT_CPLX | FDOCT_WAVE_OPT_BANDPASS)) || (w * m >= 2560 && m > 1)) ? 512 : (m == 1 ? 1024 : FDOCT_WAVE_BLOCK_SHORT);
}

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int R, bool INV, bool MIDZERO = false>
__device__ __forceinline__ void dft_reg(v2f* v) {
  if constexpr (R == 3)
    fft_reg3<INV>(v);
  else if constexpr (R == 5)
    fft_reg5<INV>(v);
  else if constexpr (R == 9)
    fft_reg9<INV>(v);
  else if constexpr (R == 15)
    fft_reg15<INV>(v);
  else if constexpr (R == 20)
    fft_reg20<INV, MIDZERO>(v);
  else
    fft_reg<R, INV>(v);
}

constexpr int plan_table_offset(const WavePlan& p, int pass) {
  int o = 0;
  for (int i = 0; i < pass; i++)
    if (p.Ns[i] > 1) o += p.Ns[i];
  return o;
}

// One Stockham pass of the n-point transform, in place: butterfly j (lanes stride over j) takes buf[j + r*nb], r < R,
// multiplies by exp(+-2 pi i r k / (Ns R)), k = j mod Ns, and writes buf[(j div Ns) Ns R + k + r Ns].  All reads of the
// pass are issued before any write (same wave: program order), which is what makes the single buffer safe.
// FROM_REGS: the inputs arrive in `rin` (element (lane + 64 t) + r*nb at rin[t*R + r]) instead of the buffer.
// FILTER: only outputs e < keep_lo or e > keep_hi are stored (last pass of the final transform).
// OCH > 0 (last pass of the zero-pad stage's inverse transform): the output row is laid out for the slope step, two pad
// elements after every OCH (= samples per lane / 2): element e goes to e + 2 (e / OCH).
// DK > 0 (with FILTER): keep_lo <= DK and keep_hi >= DKH whatever numdisplaypoints the launch has (DK = 64 lanes x the depth
// bins per lane the kernel is compiled for), so an output block r Ns .. (r + 1) Ns - 1 that lies inside [DK, DKH] is never
// stored by any launch: its store is left out at compile time and the butterfly's arithmetic behind it goes with it (dead
// code: BscanFFT.ini shows 320 of 1280 bins, blocks 2 .. 5 of the last radix-8 pass feed nothing).
#ifndef FDOCT_WAVE_PRUNE
#define FDOCT_WAVE_PRUNE 1
#endif
// ZLO <= ZHI (first pass, inputs from the buffer): elements ZLO .. ZHI of the input are zero by construction (the zero-pad
// stage's inverse transform: M W / 2 points of which the W / 2 lowest and the W / 2 - 1 highest are set).  An input block
// r nb .. (r + 1) nb - 1 that lies inside is not read -- and not written by the re-packing (wave_zero_block_*) -- and a radix-20
// butterfly whose blocks 5 .. 14 are such runs its first stage on two inputs instead of four.
constexpr bool wave_zero_block(int nb, int r, int zlo, int zhi) { return zhi >= zlo && r * nb >= zlo && (r + 1) * nb - 1 <= zhi; }
// first and one-past-last element of the run of whole zero blocks of the first pass (empty: both 0)
constexpr int wave_zero_run_begin(int n, int zlo, int zhi) {
  const WavePlan p = wave_plan(n);
  const int R = p.R[0], nb = n / R;
  for (int r = 0; r < R; r++)
    if (wave_zero_block(nb, r, zlo, zhi)) return r * nb;
  return 0;
}
constexpr int wave_zero_run_end(int n, int zlo, int zhi) {
  const WavePlan p = wave_plan(n);
  const int R = p.R[0], nb = n / R;
  int e = 0;
  for (int r = 0; r < R; r++)
    if (wave_zero_block(nb, r, zlo, zhi)) e = (r + 1) * nb;
  return e;
}
#ifndef FDOCT_WAVE_ZPRUNE
#define FDOCT_WAVE_ZPRUNE 1
#endif
// ROWS = 2 (round 6, wave_rows_of): the same pass of TWO rows' transforms side by side -- butterflies 0 .. nb - 1 work on `buf`,
// nb .. 2 nb - 1 on `buf1` -- so that a pass whose butterfly count does not fill whole rounds of 64 lanes wastes lanes once per
// PAIR of rows (an 80-point transform's radix-5 pass: 16 butterflies per row, one round of 32 lanes per pair instead of two of 16;
// the radix-8 passes of a 1280-point one: 160 per row, five rounds per pair instead of six).  Not with FROM_REGS (the gather
// fills one row's registers at a time).
template <int n, int PASS, bool INV, bool FROM_REGS, bool FILTER, int OCH = 0, int DK = 0, int DKH = 0, int ZLO = 0, int ZHI = -1, int ROWS = 1>
__device__ __forceinline__ void wave_pass(v2f* buf, const v2f* twp, int lane, const v2f* rin, int keep_lo, int keep_hi, v2f* buf1 = nullptr) {
  constexpr WavePlan plan = wave_plan(n);
  constexpr int R = plan.R[PASS], Ns = plan.Ns[PASS], nb1 = n / R, nb = ROWS * nb1, NBL = (nb + 63) / 64;
  static_assert(ROWS == 1 || (ROWS == 2 && !FROM_REGS), "two rows side by side: inputs from the buffers");
  static_assert(ZHI < ZLO || (PASS == 0 && !FROM_REGS), "known-zero inputs: first pass, from the buffer");
  constexpr bool MIDZ = R == 20 && wave_zero_block(nb1, 5, ZLO, ZHI) && wave_zero_block(nb1, 14, ZLO, ZHI);
  constexpr bool FULL = (nb % 64) == 0;
  constexpr int toff = plan_table_offset(plan, PASS);
  // The fused radix-20 first pass writes butterfly j's twenty outputs to 20 j ..: 160 bytes from lane to lane, i.e. 16-byte stores
  // that reach sixteen of the 32 banks (two cycles become four).  With wave_r20_padded(n) its output is laid out 22 complex values per
  // butterfly -- element e at e + 2 (e div 20): 176 bytes from lane to lane, all 32 banks -- and the second pass, whose butterfly j
  // reads elements j + r nb, finds them at (j + 2 (j div 20)) + r (nb + nb / 10): its own q = j div Ns is that quotient (Ns = 20).
  constexpr bool OPAD = wave_r20_padded(n) && PASS == 0, IPAD = wave_r20_padded(n) && PASS == 1;
  static_assert(!OPAD || (R == 20 && Ns == 1 && OCH == 0 && !FILTER), "padded first pass: the fused radix 20");
  static_assert(!IPAD || (Ns == 20 && nb1 % 20 == 0 && !FROM_REGS && ZHI < ZLO), "second pass behind a padded first pass");
  constexpr int istride = IPAD ? nb1 + nb1 / 10 : nb1;
  v2f v[NBL * R];
  // A pass whose butterfly count is not a multiple of 64 has idle lanes in its last round.  Long transforms (CLAMP): they
  // repeat the last butterfly (clamped index) and only their stores are masked off -- no divergent region around the
  // arithmetic; short ones, where most rounds are partial, branch around the whole butterfly instead (measured both ways).
  constexpr bool CLAMP = n >= 640;
  // butterfly jj of the pass (of the pair of rows): which row's buffer, and its index j inside that row's transform
  auto row_buf = [&](int jj) -> v2f* {
    if constexpr (ROWS == 2) return jj >= nb1 ? buf1 : buf;
    return buf;
  };
  auto row_j = [&](int jj) -> int {
    if constexpr (ROWS == 2) return jj >= nb1 ? jj - nb1 : jj;
    return jj;
  };
  static_for<0, NBL>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const int j = lane + 64 * t;
    constexpr bool PARTIAL = !FULL && 64 * t + 63 >= nb;
    const int jcc = (PARTIAL && CLAMP) ? (j < nb ? j : nb - 1) : j;
    if (!PARTIAL || CLAMP || j < nb) {
      const v2f* bp = row_buf(jcc);
      const int jc = row_j(jcc);
      static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (FROM_REGS)
          v[t * R + r] = rin[t * R + r];
        else if constexpr (wave_zero_block(nb1, r, ZLO, ZHI))
          v[t * R + r] = mk(0.f, 0.f);
        else
          v[t * R + r] = bp[(IPAD ? jc + 2 * (int)((unsigned)jc / 20u) : jc) + r * istride];
      });
    }
  });
  wave_fence();
  static_for<0, NBL>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const int j = lane + 64 * t;
    constexpr bool PARTIAL = !FULL && 64 * t + 63 >= nb;
    const int jcc = (PARTIAL && CLAMP) ? (j < nb ? j : nb - 1) : j;
    if constexpr (t > 0) __builtin_amdgcn_sched_barrier(0);  // one butterfly at a time: interleaving them costs registers
    if (!PARTIAL || CLAMP || j < nb) {
    v2f* const bp = row_buf(jcc);
    const int jc = row_j(jcc);
    int k = 0, q = jc;
    if constexpr (Ns > 1) {
      q = (int)((unsigned)jc / (unsigned)Ns);
      k = jc - q * Ns;
      v2f w[R];
      w[1] = twp[toff + k];
      if (!INV) w[1].y = -w[1].y;
      static_for<2, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        w[r] = (r & 1) ? cmul(w[r - 1], w[1]) : cmul(w[r / 2], w[r / 2]);
      });
      static_for<1, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        v[t * R + r] = cmul(v[t * R + r], w[r]);
      });
    }
    dft_reg<R, INV, MIDZ>(v + t * R);
    const int e0 = q * (Ns * R) + k;
    static_assert(OCH == 0 || (Ns % OCH) == 0, "padded rows: the pass's output stride must be whole lane chunks");
    constexpr int ostride = OCH > 0 ? Ns + 2 * (Ns / (OCH > 0 ? OCH : 1)) : Ns;
    v2f* d = bp + (OCH > 0 ? e0 + 2 * (int)((unsigned)e0 / (unsigned)(OCH > 0 ? OCH : 1)) : (OPAD ? e0 + 2 * jc : e0));   // (OPAD: e0 = 20 j)
    if (!PARTIAL || j < nb) {
      static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (FILTER) {
          constexpr bool NEVER = FDOCT_WAVE_PRUNE && DK > 0 && Ns * R == n && r * Ns >= DK && (r + 1) * Ns - 1 <= DKH;
          if constexpr (!NEVER) {
            const int e = e0 + r * Ns;
            if (e < keep_lo || e > keep_hi) d[r * Ns] = v[t * R + r];
          }
        } else {
          d[r * ostride] = v[t * R + r];
        }
      });
    }
    }
  });
  wave_fence();
}

// ROWS = 2: every pass on two rows' buffers side by side (wave_pass); P0: the first pass to run (1: the caller has run pass 0
// from registers, one row at a time).
template <int n, bool INV, bool FROM_REGS, bool FILTER_LAST, int OCH_LAST = 0, int DK = 0, int DKH = 0, int ZLO = 0, int ZHI = -1, int ROWS = 1, int P0 = 0>
__device__ __forceinline__ void wave_fft(v2f* buf, const v2f* twp, int lane, const v2f* rin, int keep_lo, int keep_hi, v2f* buf1 = nullptr) {
  constexpr WavePlan plan = wave_plan(n);
  static_assert(plan.npass > 0, "length must factor into 2, 3 and 5");
  static_for<P0, plan.npass>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    constexpr bool last = p == plan.npass - 1;
    wave_pass<n, p, INV, FROM_REGS && p == 0, FILTER_LAST && last, last ? OCH_LAST : 0, last ? DK : 0, last ? DKH : 0, p == 0 ? ZLO : 0, p == 0 ? ZHI : -1, ROWS>(
        buf, twp, lane, rin, keep_lo, keep_hi, buf1);
  });
}

constexpr int wave_pow2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
// wave-wide f64 sum by DPP (the row mean of main:1138), total returned to every lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double wdpp_add_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int tlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  const int thi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(thi, tlo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v = wdpp_add_f64<0x111, 0xf>(v);
  v = wdpp_add_f64<0x112, 0xf>(v);
  v = wdpp_add_f64<0x114, 0xf>(v);
  v = wdpp_add_f64<0x118, 0xf>(v);
  v = wdpp_add_f64<0x142, 0xa>(v);
  v = wdpp_add_f64<0x143, 0xc>(v);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// wave-wide f32 sum by DPP, total returned to every lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float wdpp_add_f32(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
  v = wdpp_add_f32<0x111, 0xf>(v);
  v = wdpp_add_f32<0x112, 0xf>(v);
  v = wdpp_add_f32<0x114, 0xf>(v);
  v = wdpp_add_f32<0x118, 0xf>(v);
  v = wdpp_add_f32<0x142, 0xa>(v);
  v = wdpp_add_f32<0x143, 0xc>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// wave-wide min / max by DPP (row-wise normalisation), result in every lane
template <int CTRL, int ROW_MASK, bool MAX>
__device__ __forceinline__ float wdpp_minmax_f32(float v) {
  // lanes the row mask leaves out keep their own value (bound_ctrl off, old = v): min/max with itself
  const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
  return MAX ? __builtin_fmaxf(v, o) : __builtin_fminf(v, o);
}
template <bool MAX>
__device__ __forceinline__ float wave_minmax_f32(float v) {
  v = wdpp_minmax_f32<0x111, 0xf, MAX>(v);
  v = wdpp_minmax_f32<0x112, 0xf, MAX>(v);
  v = wdpp_minmax_f32<0x114, 0xf, MAX>(v);
  v = wdpp_minmax_f32<0x118, 0xf, MAX>(v);
  v = wdpp_minmax_f32<0x142, 0xa, MAX>(v);
  v = wdpp_minmax_f32<0x143, 0xc, MAX>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min_f32(float v) { return wave_minmax_f32<false>(v); }
__device__ __forceinline__ float wave_max_f32(float v) { return wave_minmax_f32<true>(v); }

constexpr int imax(int a, int b) { return a > b ? a : b; }

// std::conditional without the header
template <bool C, class A, class B>
struct wave_select { typedef A type; };
template <class A, class B>
struct wave_select<false, A, B> { typedef B type; };

}  // namespace

// One wave per output A-scan (persistent: waves stride over the rows).  See the file header.
// TD: depth bins per lane (numdisplaypoints <= 64 TD): the shipped configurations display 320 / 360 bins, so the
// accumulators are 8 registers, not N/128.
// OPT (FDOCT_WAVE_OPT_*): the acquisition options beyond the plain set-up -- pi-shifted / J0 frame (main:1132), dark frame
// (BscanDark.cpp:1269), band-pass inside the zero-pad stage (BscanDark.cpp:218-236), row-wise / whole-frame min-max
// normalisation (main:1126-1129; BscanFFTsim.cpp:845 always normalises).  The library's own instantiations are
// OPT = 0; a handle that uses an option gets its kernel from the run-time compiler (fdoct_jit.cpp).
// FDOCT_WAVE_OPT_CPLX: the dispersion phase (complex rows: full-length final transform, no untangle); FDOCT_WAVE_OPT_DEEP: real
// rows displayed beyond numfftpoints / 2.
// DKP: numdisplaypoints <= DKP is guaranteed by the launch (0: no bound beyond 64 TD; wave_depth_bound)
template <int W, int M, int N, typename IN_T, int TD, int OPT = 0, int DKP = 0>
__global__ __launch_bounds__(wave_block_of(W, M, N, OPT)) void wave_kernel(const WaveArgs a) {
  constexpr bool CPLX = (OPT & FDOCT_WAVE_OPT_CPLX) != 0, DEEP = (OPT & FDOCT_WAVE_OPT_DEEP) != 0;
  static_assert(!(CPLX && DEEP), "complex rows take any depth as they are");
  constexpr int MW = M * W, NC = wave_final_points(N, OPT), WH = W / 2, LH = MW / 2;
  constexpr int SPL = (MW + 63) / 64;   // upsampled samples per lane in the slope step (contiguous; the last lanes own fewer, or none, when 64 does not divide M W)
  constexpr bool RAGGED = (MW % 64) != 0;
  constexpr int PADF = wave_row_pad_floats(W, M);  // pad floats after every lane's samples (see fdoct_wave.h)
  constexpr int SPLP = SPL + PADF, MWP = MW + 64 * PADF;  // lane stride and extent of the padded row; MWP = the zero slot
  constexpr int L = imax(imax(wave_fft_extent(NC), MWP / 2), M > 1 ? wave_fft_extent(LH) : 0);
  auto rp = [](int smp) { return PADF ? smp + PADF * (int)((unsigned)smp / (unsigned)SPL) : smp; };  // sample -> float index of the row
  constexpr int NSAMP = (W + 63) / 64;  // camera samples per lane (strided)
  static_assert(MW >= 128 && (!RAGGED || PADF == 0), "at least two upsampled samples per lane; padded rows split evenly");
  static_assert((CPLX || N % 2 == 0) && (M == 1 || W % 2 == 0), "half-length transforms need even lengths");
  static_assert(MW < 65536, "gather sources are 16-bit float indices");
  constexpr WavePlan pnc = wave_plan(NC);
  constexpr int R0 = pnc.R[0], NB0 = NC / R0, NBL0 = (NB0 + 63) / 64;
  constexpr bool FULL0 = (NB0 % 64) == 0;

  extern __shared__ __align__(16) unsigned char wsm[];
  v2f* s_tw = reinterpret_cast<v2f*>(wsm);                              // [tw_count]
  v2f* s_ph = s_tw + a.tw_count;                                        // [N] dispersion phasors (CPLX only)
  uint32_t* s_gi = reinterpret_cast<uint32_t*>(s_ph + (CPLX ? N : 0));  // [NC]
  float* s_g = reinterpret_cast<float*>(s_gi + NC);                     // [MWP], laid out like the row
  float* s_win = s_g + MWP;                                             // [W]
  float* s_ib = s_win + W;                                              // [W] 1/background, high word (1-row background only)
  float* s_il = s_ib + W;                                               // [W] ... low word (fdoct_capi.cpp::reciprocal_words)
  const int nshared = a.tw_count * 2 + NC + MWP + W + (a.ib_2d ? 0 : 2 * W) + (CPLX ? 2 * N : 0);  // in 4-byte words
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
#ifndef FDOCT_WAVE_TICKETS
#define FDOCT_WAVE_TICKETS 1   // rows handed out by ticket (round 6); 0: the static deal of rounds 2-5 (tools/ab_jit.sh)
#endif
  // Row slots are CLAIMED, not dealt (round 6; as fused_kernel does since round 1): slot s of this workgroup is output row
  // (s div nw) gridDim nw + blockIdx nw + (s mod nw) -- the static deal's rows, so the chip still sweeps the batch front to back
  // and neighbouring waves hold neighbouring rows -- and a wave that finishes takes the next slot from a workgroup-wide counter.
  // With the static deal every wave of a SIMD had the same number of rows while the hardware issues oldest-wave-first: the
  // oldest wave ran ahead, finished its share early and left its SIMD with two waves for the rest of the launch.
  __shared__ unsigned s_next_slot;   // (4 bytes of static LDS: inside the 64 bytes the launch leaves free)
  if (FDOCT_WAVE_TICKETS && tid == 0) s_next_slot = (unsigned)nw;   // slots 0 .. nw - 1 are the waves' first rows
  {
    const v2f* gtw = reinterpret_cast<const v2f*>(a.tw);
    for (int i = tid; i < a.tw_count; i += blockDim.x) s_tw[i] = gtw[i];
    for (int i = tid; i < NC; i += blockDim.x) {  // gather sources as float indices of the (padded) row; M*W = the zero slot
      const uint32_t g = a.gidx[i];
      const int lo = (int)(g & 0xffffu), hi = (int)(g >> 16);
      s_gi[i] = (uint32_t)(lo == MW ? MWP : rp(lo)) | ((uint32_t)(hi == MW ? MWP : rp(hi)) << 16);
    }
    if constexpr (CPLX) {
      const v2f* gph = reinterpret_cast<const v2f*>(a.phase);
      for (int i = tid; i < N; i += blockDim.x) s_ph[i] = gph[i];
    }
    for (int i = tid; i < MW; i += blockDim.x) s_g[rp(i)] = a.g[i];
    for (int i = tid; i < W; i += blockDim.x) s_win[i] = a.win[i];
    if (!a.ib_2d)
      for (int i = tid; i < W; i += blockDim.x) {
        s_ib[i] = a.ib[i];
        s_il[i] = a.il[i];
      }
  }
  __syncthreads();  // the only workgroup barrier: the shared tables
  constexpr int PRIV = wave_private_bytes(L, MW);  // bytes of one row's buffer (fdoct_wave.h: one rule for kernel and host)
  constexpr int ROWS = wave_rows_of(W, M, N, OPT);  // rows a wave works on side by side (2 on the short zero-padded shapes, round 6)
  v2f* buf = reinterpret_cast<v2f*>(wsm + (((size_t)nshared * 4 + 15) & ~(size_t)15) + (size_t)wave * (ROWS * PRIV));
  float* bf = reinterpret_cast<float*>(buf);

  const v2f* tw_nc = s_tw + a.off_nc;
  const v2f* tw_lh = s_tw + a.off_lh;
  const v2f* tw_wh = s_tw + a.off_wh;
  const v2f* tw_w = s_tw + a.off_tww;
  const v2f* tw_mw = s_tw + a.off_twmw;
  const v2f* tw_n = s_tw + a.off_twn;
  const int D = a.D;
  const unsigned char* frames = static_cast<const unsigned char*>(a.frames);
  // rows are wave-uniform (one wave, one A-scan) and fewer than 2^31 (host): 32-bit scalar arithmetic
  const unsigned total = (unsigned)a.total_out_rows, stride = gridDim.x * (unsigned)nw;
  const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (unsigned)nw + (unsigned)wave));
  const unsigned wg_base = blockIdx.x * (unsigned)nw;
  // the row of slot s (a slot past the end of the batch maps to a row >= total: rows grow with the slot); saturating, so that
  // 32 bits hold it whatever the slot
  auto slot_row = [&](unsigned s_) -> unsigned {
    const unsigned q_ = s_ / (unsigned)nw, m_ = s_ - q_ * (unsigned)nw;
    const unsigned long long o_ = (unsigned long long)q_ * stride + wg_base + m_;
    return o_ < (unsigned long long)total ? (unsigned)o_ : total;
  };

  if constexpr (ROWS == 2) {
    // ================= two rows per wave (round 6; EXPERIMENTS.md section 5) =================
    // The short zero-padded shapes (M W <= 1280: BscanFFT.ini's 160 x 4, 320 x 4 ...) leave most lanes idle in their small
    // transforms: an 80-point transform's radix-5 pass has 16 butterflies.  Here a wave owns TWO consecutive output rows, each
    // in a buffer of its own; every phase runs for both between the same pair of fences (twice the independent work in flight per
    // wave, half the waves per workgroup: the LDS holds the same twelve rows), and the transform passes run the two rows'
    // butterflies side by side (wave_pass, ROWS = 2), so that a partial round is wasted once per pair.
    static_assert(M > 1 && !CPLX && !DEEP && !RAGGED, "two rows per wave: short zero-padded real rows");
    constexpr bool BIN2 = (OPT & FDOCT_WAVE_OPT_BIN2) != 0;
    typedef typename wave_select<BIN2, typename wave_select<sizeof(IN_T) == 1, unsigned short, unsigned int>::type, IN_T>::type RAW_T;
    v2f* const bufr[2] = {buf, buf + PRIV / 8};
    float* const bfr[2] = {bf, bf + PRIV / 4};
    // a slot is a PAIR of rows: the slot's rows are 2 (row of the one-row deal), + 1
    auto slot_pair = [&](unsigned s_) -> unsigned {
      const unsigned q_ = s_ / (unsigned)nw, m_ = s_ - q_ * (unsigned)nw;
      const unsigned long long o_ = 2ull * ((unsigned long long)q_ * stride + wg_base + m_);
      return o_ < (unsigned long long)total ? (unsigned)o_ : total;
    };
    RAW_T rawn[2][NSAMP], rawn2[2][BIN2 ? NSAMP : 1];
    auto load_raw2 = [&](unsigned o_, int ai_) {
      static_for<0, 2>([&](auto rhc) {
        constexpr int rho = decltype(rhc)::value;
        const unsigned orow = (o_ + rho < total) ? o_ + rho : o_;   // (an odd batch's last pair: the second row repeats the first, its results are not stored)
        const unsigned g_ = orow / (unsigned)a.H;
        const unsigned r_ = orow - g_ * (unsigned)a.H;
        if constexpr (BIN2) {
          const unsigned char* row0 = frames + ((long long)(g_ * (unsigned)a.A + (unsigned)ai_) * (2 * a.H) + 2 * r_) * a.pitch_bytes;
          const RAW_T* p0 = reinterpret_cast<const RAW_T*>(row0);
          const RAW_T* p1 = reinterpret_cast<const RAW_T*>(row0 + a.pitch_bytes);
#pragma unroll
          for (int c = 0; c < NSAMP; c++) {
            const int i = lane + 64 * c;
            rawn[rho][c] = ((W % 64) == 0 || i < W) ? p0[i] : RAW_T(0);
            rawn2[rho][c] = ((W % 64) == 0 || i < W) ? p1[i] : RAW_T(0);
          }
        } else {
          const IN_T* row = reinterpret_cast<const IN_T*>(frames + ((long long)(g_ * (unsigned)a.A + (unsigned)ai_) * a.H + r_) * a.pitch_bytes);
#pragma unroll
          for (int c = 0; c < NSAMP; c++) {
            const int i = lane + 64 * c;
            rawn[rho][c] = ((W % 64) == 0 || i < W) ? row[i] : IN_T(0);
          }
        }
      });
    };
    const unsigned first2 = (unsigned)__builtin_amdgcn_readfirstlane((int)slot_pair((unsigned)wave));
    if (first2 < total) load_raw2(first2, 0);
    constexpr bool NORMED = (OPT & (FDOCT_WAVE_OPT_ROWNORM | FDOCT_WAVE_OPT_FRAMENORM)) != 0;
    constexpr bool LOWW = NORMED || (OPT & (FDOCT_WAVE_OPT_DARK | FDOCT_WAVE_OPT_PI)) != 0;
    constexpr int DK0 = 64 * TD < NC ? 64 * TD : NC, DK = (DKP > 0 && DKP < DK0) ? DKP : DK0;
    constexpr int ZLO = FDOCT_WAVE_ZPRUNE ? WH : 0, ZHI = FDOCT_WAVE_ZPRUNE ? LH - WH : -1;
    constexpr int ZB = wave_zero_run_begin(LH, ZLO, ZHI), ZE = wave_zero_run_end(LH, ZLO, ZHI);   // ZB == ZE: no such block
    for (unsigned o = first2, o_next = 0; o < total; o = o_next) {
      if constexpr (FDOCT_WAVE_TICKETS != 0) {
        unsigned tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(&s_next_slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        o_next = slot_pair((unsigned)__builtin_amdgcn_readfirstlane((int)tk));
      } else {
        o_next = o + 2 * stride < o ? total : (o + 2 * stride < total ? o + 2 * stride : total);
      }
      const bool valid1 = o + 1 < total;
      unsigned gg[2];
      int rr2[2];
      static_for<0, 2>([&](auto rhc) {
        constexpr int rho = decltype(rhc)::value;
        const unsigned orow = (o + rho < total) ? o + rho : o;
        gg[rho] = orow / (unsigned)a.H;
        rr2[rho] = (int)(orow - gg[rho] * (unsigned)a.H);
      });
      float acc[2][TD];
#pragma unroll
      for (int t = 0; t < TD; t++) acc[0][t] = acc[1][t] = 0.f;

      for (int ai = 0; ai < a.A; ai++) {
        IN_T raw[2][NSAMP];
        static_for<0, 2>([&](auto rhc) {
          constexpr int rho = decltype(rhc)::value;
#pragma unroll
          for (int c = 0; c < NSAMP; c++) {
            if constexpr (BIN2) {
              constexpr unsigned SH = 8 * sizeof(IN_T), MK = (1u << SH) - 1u;
              const unsigned p_ = (unsigned)rawn[rho][c], q_ = (unsigned)rawn2[rho][c];
              raw[rho][c] = (IN_T)(((p_ & MK) + (p_ >> SH) + (q_ & MK) + (q_ >> SH) + 2u) >> 2);
            } else {
              raw[rho][c] = rawn[rho][c];
            }
          }
        });
        {
          unsigned on = o;
          int an = ai + 1;
          if (an == a.A) {
            an = 0;
            on = o_next;
          }
          if (on < total) load_raw2(on, an);
        }
        // ---- A2/A3 of both rows (the expressions of the one-row body below, row by row)
        static_for<0, 2>([&](auto rhc) {
          constexpr int rho = decltype(rhc)::value;
          const int r = rr2[rho];
          float* const bf_ = bfr[rho];
          float y[NSAMP], ibv[NSAMP], ilv[NSAMP];
          if (a.ib_2d) {
            const float* ibr = a.ib + (size_t)r * W;
#pragma unroll
            for (int c = 0; c < NSAMP; c++) {
              const int i = lane + 64 * c;
              ibv[c] = ((W % 64) == 0 || i < W) ? ibr[i] : 0.f;
              ilv[c] = ((W % 64) == 0 || i < W) ? a.il[(size_t)r * W + i] : 0.f;
            }
          } else {
#pragma unroll
            for (int c = 0; c < NSAMP; c++) {
              const int i = lane + 64 * c;
              ibv[c] = ((W % 64) == 0 || i < W) ? s_ib[i] : 0.f;
              ilv[c] = ((W % 64) == 0 || i < W) ? s_il[i] : 0.f;
            }
          }
          float vs[NSAMP], vlo[LOWW ? NSAMP : 1];
#pragma unroll
          for (int c = 0; c < NSAMP; c++) {
            const int i = lane + 64 * c;
            vs[c] = (float)raw[rho][c];
            if constexpr (LOWW) vlo[c] = 0.f;
            if constexpr ((OPT & FDOCT_WAVE_OPT_DARK) != 0)
              if ((W % 64) == 0 || i < W) vs[c] = two_diff(vs[c], a.yd[(a.yd_2d ? (size_t)r * W : 0) + i], vlo[c]);
          }
          if constexpr (NORMED) {
            float mn, mx;
            if constexpr ((OPT & FDOCT_WAVE_OPT_ROWNORM) != 0) {
              mn = __builtin_inff();
              mx = -__builtin_inff();
#pragma unroll
              for (int c = 0; c < NSAMP; c++) {
                const int i = lane + 64 * c;
                if ((W % 64) == 0 || i < W) {
                  mn = __builtin_fminf(mn, vs[c]);
                  mx = __builtin_fmaxf(mx, vs[c]);
                }
              }
              mn = wave_min_f32(mn);
              mx = wave_max_f32(mx);
            } else {
              const v2f mmx = reinterpret_cast<const v2f*>(a.minmax)[(size_t)gg[rho] * (unsigned)a.A + (unsigned)ai];
              mn = mmx.x;
              mx = mmx.y;
            }
            const float sc = (mx - mn > 2.220446049250313e-16f) ? 1.f / (mx - mn) : 0.f;
#pragma unroll
            for (int c = 0; c < NSAMP; c++) {
              float e;
              const float vm = two_diff(vs[c], mn, e);
              vs[c] = vm * sc;
              vlo[c] = fmaf(vlo[c] + e, sc, fmaf(vm, sc, -vs[c]));
            }
          }
          if constexpr ((OPT & FDOCT_WAVE_OPT_PI) != 0) {
#pragma unroll
            for (int c = 0; c < NSAMP; c++) {
              const int i = lane + 64 * c;
              if ((W % 64) == 0 || i < W) {
                float e;
                vs[c] = two_diff(vs[c], a.yp[(a.yp_2d ? (size_t)r * W : 0) + i], e);
                vlo[c] += e;
              }
            }
          }
          constexpr int CM = (NSAMP - 1) / 2;
          constexpr int CMN = W >= 64 * (CM + 1) ? 64 : W - 64 * CM;
          const float c0 = wave_sum_f32(vs[CM] * ibv[CM]) * (1.f / (float)CMN);
          float sum = 0.f;
#pragma unroll
          for (int c = 0; c < NSAMP; c++) {
            const int i = lane + 64 * c;
            y[c] = 0.f;
            if ((W % 64) == 0 || i < W) {
              y[c] = fmaf(vs[c], ilv[c], fmaf(vs[c], ibv[c], -c0));
              if constexpr (LOWW) y[c] = fmaf(vlo[c], ibv[c], y[c]);
              sum += y[c];
            }
          }
          const float md = wave_sum_f32(sum) * (1.f / (float)W);
#pragma unroll
          for (int c = 0; c < NSAMP; c++) {
            const int i = lane + 64 * c;
            if ((W % 64) == 0 || i < W) bf_[i] = (y[c] - md) * s_win[i];  // main:1139, 1142
          }
        });
        wave_fence();
        // ---- A4 (main:180-245) at half length, both rows: forward W/2-point transforms side by side, the re-packing of each
        // spectrum, inverse M W/2-point transforms side by side
        wave_fft<WH, false, false, false, 0, 0, 0, 0, -1, 2>(bufr[0], tw_wh, lane, nullptr, 0, 0, bufr[1]);
        constexpr int NK = (WH + 63) / 64;
        v2f zk[2][NK], zp[2][NK];
        static_for<0, 2>([&](auto rhc) {
          constexpr int rho = decltype(rhc)::value;
#pragma unroll
          for (int t = 0; t < NK; t++) {
            const int k = lane + 64 * t;
            if ((WH % 64) == 0 || k < WH) {
              zk[rho][t] = bufr[rho][k];
              zp[rho][t] = bufr[rho][k == 0 ? 0 : WH - k];
            }
          }
        });
        wave_fence();
        constexpr float inv_w = 1.f / (float)W;
        static_for<0, 2>([&](auto rhc) {
          constexpr int rho = decltype(rhc)::value;
          v2f* const b_ = bufr[rho];
#pragma unroll
          for (int t = 0; t < NK; t++) {
            const int k = lane + 64 * t;
            if ((WH % 64) == 0 || k < WH) {
              const float ax = zk[rho][t].x + zp[rho][t].x, ay = zk[rho][t].y - zp[rho][t].y, bx = zk[rho][t].x - zp[rho][t].x, by = zk[rho][t].y + zp[rho][t].y;
              const v2f tc = tw_w[k];
              const float qx = fmaf(tc.y, by, tc.x * bx), qy = fmaf(-tc.y, bx, tc.x * by);
              float xx = 0.5f * (ax + qy) * inv_w, xy = (k == 0) ? 0.f : 0.5f * (ay - qx) * inv_w;
              static_assert((OPT & FDOCT_WAVE_OPT_BANDPASS) == 0, "the band-pass runs in the one-row body (wave_rows_of)");
              const v2f w = tw_mw[k];
              const float px = fmaf(-xy, w.y, xx * w.x), py = fmaf(xy, w.x, xx * w.y);
              b_[k] = mk(xx - py, xy + px);
              if (k > 0) {
                const float cx = xx, cy = -xy;
                const float q2x = fmaf(-cy, w.y, cx * -w.x), q2y = fmaf(cy, -w.x, cx * w.y);
                b_[LH - k] = mk(cx + q2y, cy - q2x);
              }
            }
          }
          constexpr int NZ1 = (ZE > ZB ? ZB : LH - WH + 1) - WH;
#pragma unroll
          for (int t = 0; t < (NZ1 + 63) / 64; t++) {
            const int k = WH + lane + 64 * t;
            if (k < WH + NZ1) b_[k] = mk(0.f, 0.f);
          }
          if constexpr (ZE > ZB) {
            constexpr int NZ2 = LH - WH + 1 - ZE;
#pragma unroll
            for (int t = 0; t < (NZ2 + 63) / 64; t++) {
              const int k = ZE + lane + 64 * t;
              if (k <= LH - WH) b_[k] = mk(0.f, 0.f);
            }
          }
        });
        wave_fence();
        wave_fft<LH, true, false, false, PADF ? SPL / 2 : 0, 0, 0, ZLO, ZHI, 2>(bufr[0], tw_lh, lane, nullptr, 0, 0, bufr[1]);

        // ---- A5 (first half): the slope step of both rows, in place
        {
          float mylast[2], y1[2];
          static_for<0, 2>([&](auto rhc) {
            constexpr int rho = decltype(rhc)::value;
            mylast[rho] = bfr[rho][lane * SPLP + SPL - 1];
            y1[rho] = bfr[rho][lane * SPLP + 1];
          });
          wave_fence();
          constexpr int CH = SPL % 10 == 0 ? 10 : (SPL % 9 == 0 ? 9 : (SPL % 8 == 0 ? 8 : (SPL <= 12 ? SPL : 8)));
          static_for<0, 2>([&](auto rhc) {
            constexpr int rho = decltype(rhc)::value;
            const float* src = bfr[rho] + lane * SPLP;
            const float* gs = s_g + lane * SPLP;
            float* dst = bfr[rho] + lane * SPLP;
            float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mylast[rho]), 0x138, 0xf, 0xf, false));
            static_for<0, (SPL + CH - 1) / CH>([&](auto cc) {
              constexpr int c0 = decltype(cc)::value * CH;
              constexpr int CN = SPL - c0 < CH ? SPL - c0 : CH;
              float yy[CN], ggv[CN];
#pragma unroll
              for (int c = 0; c < CN; c++) {
                yy[c] = src[c0 + c];
                ggv[c] = gs[c0 + c];
              }
#pragma unroll
              for (int c = 0; c < CN; c++) {
                float slope = yy[c] - (c == 0 ? prev : yy[c - 1]);
                if (c0 + c == 0) slope = (lane == 0) ? (y1[rho] - yy[0]) : slope;
                dst[c0 + c] = fmaf(ggv[c], slope, yy[c]);
              }
              prev = yy[CN - 1];
            });
            if (lane == 0) bfr[rho][MWP] = 0.f;  // source of data_ylin[0] and data_ylin[N-1] (never written by the reference: 0)
          });
        }
        wave_fence();

        // ---- A5 (second half) + A6: the gather fills the first pass's registers, one row at a time; the remaining passes of the
        // two final transforms run side by side
        static_for<0, 2>([&](auto rhc) {
          constexpr int rho = decltype(rhc)::value;
          v2f zin[NBL0 * R0];
          static_for<0, NBL0>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int j = lane + 64 * t;
            if (FULL0 || j < NB0) {
              static_for<0, R0>([&](auto rc) {
                constexpr int rr = decltype(rc)::value;
                const uint32_t gi = s_gi[j + rr * NB0];
                zin[t * R0 + rr] = mk(bfr[rho][gi & 0xffffu], bfr[rho][gi >> 16]);
              });
            }
          });
          wave_fence();
          wave_pass<NC, 0, true, true, pnc.npass == 1, 0, pnc.npass == 1 ? DK : 0, pnc.npass == 1 ? NC - DK : 0>(bufr[rho], tw_nc, lane, zin, D, NC - D);
        });
        wave_fft<NC, true, false, true, 0, DK, NC - DK, 0, -1, 2, 1>(bufr[0], tw_nc, lane, nullptr, D, NC - D, bufr[1]);

        // ---- A8: untangle, magnitude
        static_for<0, 2>([&](auto rhc) {
          constexpr int rho = decltype(rhc)::value;
#pragma unroll
          for (int t = 0; t < TD; t++) {
            const int b = lane + 64 * t;
            if (b < D) {
              const v2f zkk = bufr[rho][b];
              const v2f zpp = bufr[rho][b == 0 ? 0 : NC - b];
              const v2f w = tw_n[b];
              const float ax = zkk.x + zpp.x, ay = zkk.y - zpp.y, bx = zkk.x - zpp.x, by = zkk.y + zpp.y;
              const float qx = fmaf(-w.y, by, w.x * bx), qy = fmaf(w.y, bx, w.x * by);
              const float xr = ax + qy, xi = ay - qx;
              acc[rho][t] += 0.5f * fast_sqrt(fmaf(xr, xr, xi * xi));
            }
          }
        });
        wave_fence();
      }

      // ---- A9/A10 of both rows
      static_for<0, 2>([&](auto rhc) {
        constexpr int rho = decltype(rhc)::value;
        if (rho == 0 || valid1) {
          float* om = a.out_mag ? a.out_mag + (size_t)(o + rho) * D : nullptr;
          float* od = a.out_db ? a.out_db + (size_t)(o + rho) * D : nullptr;
          float db4 = 0.f;
          if (od && a.dcmask && D > 4) db4 = a.db_scale * fast_log2(fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc[rho][0]), 4)), a.inv_A, a.eps));
#pragma unroll
          for (int t = 0; t < TD; t++) {
            const int b = lane + 64 * t;
            if (b < D) {
              const float v = fmaf(acc[rho][t], a.inv_A, a.eps);
              if (om) __builtin_nontemporal_store(v, om + b);
              if (od) __builtin_nontemporal_store((a.dcmask && D > 4 && b < 2) ? db4 : a.db_scale * fast_log2(v), od + b);
            }
          }
        }
      });
    }
    return;
  }

  // camera samples are loaded one input row ahead (the row's own work hides the latency): sample i = lane + 64 c
  // OPT & FDOCT_WAVE_OPT_BIN2: `frames` are the RAW camera frames, 2 H rows of 2 W samples, and the 2 x 2 software binning of
  // main:958 (cv::resize INTER_AREA by 1/2 = the rounded mean of each 2 x 2 block, in the sample type) happens in these
  // loads instead of in a pass of its own: the two horizontal neighbours arrive as one PAIR from each of the two raw rows.
  constexpr bool BIN2 = (OPT & FDOCT_WAVE_OPT_BIN2) != 0;
  static_assert(!BIN2 || sizeof(IN_T) <= 2, "software binning is defined on the camera's integer samples");
  typedef typename wave_select<BIN2, typename wave_select<sizeof(IN_T) == 1, unsigned short, unsigned int>::type, IN_T>::type RAW_T;
  RAW_T rawn[NSAMP], rawn2[BIN2 ? NSAMP : 1];
  auto load_raw = [&](unsigned o_, int ai_) {
    const unsigned g_ = o_ / (unsigned)a.H;
    const unsigned r_ = o_ - g_ * (unsigned)a.H;
    if constexpr (BIN2) {
      const unsigned char* row0 = frames + ((long long)(g_ * (unsigned)a.A + (unsigned)ai_) * (2 * a.H) + 2 * r_) * a.pitch_bytes;
      const RAW_T* p0 = reinterpret_cast<const RAW_T*>(row0);
      const RAW_T* p1 = reinterpret_cast<const RAW_T*>(row0 + a.pitch_bytes);
#pragma unroll
      for (int c = 0; c < NSAMP; c++) {
        const int i = lane + 64 * c;
        rawn[c] = ((W % 64) == 0 || i < W) ? p0[i] : RAW_T(0);
        rawn2[c] = ((W % 64) == 0 || i < W) ? p1[i] : RAW_T(0);
      }
    } else {
      const IN_T* row = reinterpret_cast<const IN_T*>(frames + ((long long)(g_ * (unsigned)a.A + (unsigned)ai_) * a.H + r_) * a.pitch_bytes);
#pragma unroll
      for (int c = 0; c < NSAMP; c++) {
        const int i = lane + 64 * c;
        rawn[c] = ((W % 64) == 0 || i < W) ? row[i] : IN_T(0);
      }
    }
  };
  if (first < total) load_raw(first, 0);

  // row-invariant tables in registers where the budget allows (wave_resident_tables)
  constexpr bool RESG = wave_resident_tables(W, M, N) && TD <= 8;   // fractionalk by sample
#ifndef FDOCT_WAVE_RESGI
#define FDOCT_WAVE_RESGI 0   // measured on BscanFFT.ini (tools/ab_jit.sh): 294 against 322 M input A-scans/s with the sources resident -- the registers are not to spare after all
#endif
  // the gather sources too (one word per transform point): where the short rows' 168-register budget has them to spare (the
  // 160 x 4 and 320 x 4 shapes use ~140); the long-row shapes are 20 registers short
  constexpr bool RESGI = FDOCT_WAVE_RESGI && !CPLX && M > 1 && wave_block_of(W, M, N, OPT) == 768 && NBL0 * R0 <= 24;
  float g_res[RESG ? SPL : 1];
  uint32_t gi_res[RESGI ? NBL0 * R0 : 1];
  if constexpr (RESG) {
#pragma unroll
    for (int c = 0; c < SPL; c++) g_res[c] = s_g[lane * SPLP + c];
  }
  if constexpr (RESGI) {
    static_for<0, NBL0>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const int j = lane + 64 * t;
      static_for<0, R0>([&](auto rc) {
        constexpr int rr = decltype(rc)::value;
        gi_res[t * R0 + rr] = (FULL0 || j < NB0) ? s_gi[j + rr * NB0] : 0u;
      });
    });
  }

#ifdef FDOCT_WAVE_PROBE  // where a row's cycles go: s_memtime at the phase boundaries of one wave, summed over its rows
  unsigned long long pr_acc[12] = {}, pr_t = 0;
#define FDOCT_PR(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); pr_acc[i] += t_ - pr_t; pr_t = t_; } while (0)
#else
#define FDOCT_PR(i) do {} while (0)
#endif
  for (unsigned o = first, o_next = 0; o < total; o = o_next) {
    // the next row: claimed now (one LDS round trip, a whole output row before the prefetch of its samples needs it)
    if constexpr (FDOCT_WAVE_TICKETS != 0) {
      unsigned tk = 0;
      if (lane == 0) tk = __hip_atomic_fetch_add(&s_next_slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      o_next = slot_row((unsigned)__builtin_amdgcn_readfirstlane((int)tk));
    } else {
      o_next = o + stride < o ? total : o + stride;
    }
    const unsigned g = o / (unsigned)a.H;
    const int r = (int)(o - g * (unsigned)a.H);
    float acc[TD];
#pragma unroll
    for (int t = 0; t < TD; t++) acc[t] = 0.f;

    for (int ai = 0; ai < a.A; ai++) {
#ifdef FDOCT_WAVE_PROBE
      pr_t = __builtin_readcyclecounter();
#endif
      IN_T raw[NSAMP];
#pragma unroll
      for (int c = 0; c < NSAMP; c++) {
        if constexpr (BIN2) {
          constexpr unsigned SH = 8 * sizeof(IN_T), MK = (1u << SH) - 1u;
          const unsigned p = (unsigned)rawn[c], q = (unsigned)rawn2[c];
          raw[c] = (IN_T)(((p & MK) + (p >> SH) + (q & MK) + (q >> SH) + 2u) >> 2);   // as bin2x2_kernel (fdoct_generic.hip)
        } else {
          raw[c] = rawn[c];
        }
      }
      {
        unsigned on = o;
        int an = ai + 1;
        if (an == a.A) {
          an = 0;
          on = o_next;
        }
        if (on < total) load_raw(on, an);
      }
      // ---- A2/A3: 1/background, row mean (f64), window.  Sample i = lane + 64 c.
      float y[NSAMP], ibv[NSAMP], ilv[NSAMP];
      // 1/background: a full frame comes from global memory, one spectrum from the shared LDS copy (two separate loops:
      // one loop over a selected pointer would turn both into flat loads)
      if (a.ib_2d) {
        const float* ibr = a.ib + (size_t)r * W;
#pragma unroll
        for (int c = 0; c < NSAMP; c++) {
          const int i = lane + 64 * c;
          ibv[c] = ((W % 64) == 0 || i < W) ? ibr[i] : 0.f;
          ilv[c] = ((W % 64) == 0 || i < W) ? a.il[(size_t)r * W + i] : 0.f;
        }
      } else {
#pragma unroll
        for (int c = 0; c < NSAMP; c++) {
          const int i = lane + 64 * c;
          ibv[c] = ((W % 64) == 0 || i < W) ? s_ib[i] : 0.f;
          ilv[c] = ((W % 64) == 0 || i < W) ? s_il[i] : 0.f;
        }
      }
      // the camera sample less the dark frame (BscanDark.cpp:1269), less the pi-shifted / J0 frame (main:1132): v = (y - yd) - yp
      // in the reference's order: dark, row-wise / whole-frame min-max normalisation to [0, 1] (main:1126-1129, normalizerows
      // main:88-97; the whole-frame pass is the identity after the row-wise one), pi frame
      constexpr bool NORMED = (OPT & (FDOCT_WAVE_OPT_ROWNORM | FDOCT_WAVE_OPT_FRAMENORM)) != 0;
      // (vlo: the sample's second word -- the exact residuals of the dark, normalisation and pi differences (two_diff) and the
      // normalised sample's own; round 6: with any of the three options, not only the normalisations)
      constexpr bool LOWW = NORMED || (OPT & (FDOCT_WAVE_OPT_DARK | FDOCT_WAVE_OPT_PI)) != 0;
      float vs[NSAMP], vlo[LOWW ? NSAMP : 1];
#pragma unroll
      for (int c = 0; c < NSAMP; c++) {
        const int i = lane + 64 * c;
        vs[c] = (float)raw[c];
        if constexpr (LOWW) vlo[c] = 0.f;
        if constexpr ((OPT & FDOCT_WAVE_OPT_DARK) != 0)
          if ((W % 64) == 0 || i < W) vs[c] = two_diff(vs[c], a.yd[(a.yd_2d ? (size_t)r * W : 0) + i], vlo[c]);
      }
      [[maybe_unused]] float norm_sc = 1.f;   // the normalisation's scale (what a dark frame's second word is multiplied by, below)
      if constexpr ((OPT & (FDOCT_WAVE_OPT_ROWNORM | FDOCT_WAVE_OPT_FRAMENORM)) != 0) {
        float mn, mx;
        if constexpr ((OPT & FDOCT_WAVE_OPT_ROWNORM) != 0) {
          mn = __builtin_inff();
          mx = -__builtin_inff();
#pragma unroll
          for (int c = 0; c < NSAMP; c++) {
            const int i = lane + 64 * c;
            if ((W % 64) == 0 || i < W) {
              mn = __builtin_fminf(mn, vs[c]);
              mx = __builtin_fmaxf(mx, vs[c]);
            }
          }
          mn = wave_min_f32(mn);
          mx = wave_max_f32(mx);
        } else {
          const v2f mmx = reinterpret_cast<const v2f*>(a.minmax)[(size_t)g * (unsigned)a.A + (unsigned)ai];  // of the whole frame, from the pre-pass
          mn = mmx.x;
          mx = mmx.y;
        }
        // cv::normalize(NORM_MINMAX, 0, 1): scale = 1 / (max - min), 0 when the range is below DBL_EPSILON.  The normalised
        // sample (v - min) * scale is not a float, and rounding it would be a rounding at the size of the DC level (random from
        // sample to sample): it goes into the division as two floats, the rounded product and its exact residual
        const float sc = (mx - mn > 2.220446049250313e-16f) ? 1.f / (mx - mn) : 0.f;
        norm_sc = sc;
#pragma unroll
        for (int c = 0; c < NSAMP; c++) {
          float e;
          const float vm = two_diff(vs[c], mn, e);   // (exact on the camera's integers; after a dark frame it is not)
          vs[c] = vm * sc;
          vlo[c] = fmaf(vlo[c] + e, sc, fmaf(vm, sc, -vs[c]));
        }
      }
      if constexpr ((OPT & FDOCT_WAVE_OPT_PI) != 0) {
#pragma unroll
        for (int c = 0; c < NSAMP; c++) {
          const int i = lane + 64 * c;
          if ((W % 64) == 0 || i < W) {
            float e;
            vs[c] = two_diff(vs[c], a.yp[(a.yp_2d ? (size_t)r * W : 0) + i], e);
            vlo[c] += e;
          }
        }
      }
      // BscanDark's band-pass (dark:218-236) keeps 3 <= k < W/10 of the row's spectrum: what is displayed is the little the
      // window leaks into those few bins, and every float rounding in front of the blanking is a rounding at the size of the
      // WHOLE row (the f32 restatement itself sits up to 3 x the tolerance from the chain in double on such rows).  With the
      // band-pass on, the row is formed in double -- sample + residual word times the two reciprocal words, the mean and the
      // window in double (main:1132-1142) -- and the kept bins are evaluated directly in double below (no float forward DFT).
      constexpr bool BP = M > 1 && (OPT & FDOCT_WAVE_OPT_BANDPASS) != 0;
      if constexpr (BP) {
        double xdv[NSAMP], sd = 0.0;
#pragma unroll
        for (int c = 0; c < NSAMP; c++) {
          const int i = lane + 64 * c;
          xdv[c] = 0.0;
          if ((W % 64) == 0 || i < W) {
            double num = (double)vs[c];
            if constexpr (LOWW) num += (double)vlo[c];
            // (the dark and pi frames' second words: the planes are floats, the reference's frames doubles)
            if constexpr ((OPT & FDOCT_WAVE_OPT_DARK) != 0) num -= (double)a.yd_lo[(a.yd_2d ? (size_t)r * W : 0) + i] * (double)norm_sc;
            if constexpr ((OPT & FDOCT_WAVE_OPT_PI) != 0) num -= (double)a.yp_lo[(a.yp_2d ? (size_t)r * W : 0) + i];
            xdv[c] = num * ((double)ibv[c] + (double)ilv[c]);
            sd += xdv[c];
          }
        }
        const double meand = wave_sum_f64(sd) * (1.0 / (double)W);
        double* const xd = reinterpret_cast<double*>(buf);   // [W] (the buffer holds M W / 2 >= W complex floats)
#pragma unroll
        for (int c = 0; c < NSAMP; c++) {
          const int i = lane + 64 * c;
          if ((W % 64) == 0 || i < W) xd[i] = (xdv[c] - meand) * ((double)s_win[i] + (double)a.win_lo[i]);
        }
      } else {
#ifdef FDOCT_WAVE_OLD_MEAN  // tuning: f64 sum of the rounded products, f64 division
      double sum = 0.0;
#pragma unroll
      for (int c = 0; c < NSAMP; c++) {
        const int i = lane + 64 * c;
        y[c] = 0.f;
        if ((W % 64) == 0 || i < W) {
          y[c] = fmaf(vs[c], ilv[c], vs[c] * ibv[c]);  // main:1132, x/0 = 0 through the host-side reciprocal
          sum += (double)y[c];
        }
      }
      sum = wave_sum_f64(sum);
      const double mean = sum / (double)W;  // main:1138
      const float mh = (float)mean, ml = (float)(mean - (double)mh);
#pragma unroll
      for (int c = 0; c < NSAMP; c++) {
        const int i = lane + 64 * c;
        if ((W % 64) == 0 || i < W) bf[i] = ((y[c] - mh) - ml) * s_win[i];  // main:1139, 1142
      }
#else
      // main:1132, 1138: x = v / yb (x/0 = 0 through the host-side reciprocal) and its row mean, with no DC-sized rounding
      // and no f64: c0, the average of 64 consecutive samples from the middle of the row, is a wave-uniform estimate of the
      // mean; d = fma(v, 1/yb, -c0) is the exact product minus c0, rounded at the size of the deviation from it, and
      // x - mean = d - mean(d) (as the fast path of fdoct_kernels.hip does; the f64 sum and division this replaces were
      // ~5 % of the row's instructions).  c0 has to be a GOOD estimate: the f32 sum of the d is exact to an ulp of its own
      // size, and what it is off by reaches depth bin 0 multiplied by the row length (a single sample as c0 -- the mean
      // plus that sample's fringe -- left 1e-5 of the background level there: seen when only a few bins near DC are displayed).
      constexpr int CM = (NSAMP - 1) / 2;                            // samples 64 CM .. 64 CM + 63 (fewer in a row shorter than that)
      constexpr int CMN = W >= 64 * (CM + 1) ? 64 : W - 64 * CM;    // (ibv is 0 past the end of the row)
      const float c0 = wave_sum_f32(vs[CM] * ibv[CM]) * (1.f / (float)CMN);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < NSAMP; c++) {
        const int i = lane + 64 * c;
        y[c] = 0.f;
        if ((W % 64) == 0 || i < W) {
          // 1/yb = ibv + ilv: the second fma adds what the f32 reciprocal alone leaves out (<= 6e-8 of the quotient, a fixed
          // DC-sized pattern), rounded at the size of the deviation like the first
          y[c] = fmaf(vs[c], ilv[c], fmaf(vs[c], ibv[c], -c0));
          if constexpr (LOWW) y[c] = fmaf(vlo[c], ibv[c], y[c]);
          sum += y[c];
        }
      }
      const float md = wave_sum_f32(sum) * (1.f / (float)W);
#pragma unroll
      for (int c = 0; c < NSAMP; c++) {
        const int i = lane + 64 * c;
        if ((W % 64) == 0 || i < W) bf[i] = (y[c] - md) * s_win[i];  // main:1139, 1142
      }
#endif
      }  // !BP
      wave_fence();
      FDOCT_PR(0);   // loads' tail, A2 / A3

      if constexpr (M > 1) {
        // ---- A4: zero-pad spectral upsampling (main:180-245) at half length, as fdoct_generic.hip states it:
        //   forward: z[n] = y[2n] + i y[2n+1] (the row read as complex), Zf = DFT_{W/2}(z),
        //            F[k] = ((Zf[k] + conj Zf[W/2-k]) - i e^(-2 pi i k/W) (Zf[k] - conj Zf[W/2-k]))/2, X = F/W (DFT_SCALE), Im X[0] dropped;
        //   inverse: Z[k] = X[k] (1 + i w^k), Z[L/2-k] = conj(X[k]) (1 - i w^(L/2-k)), w = e^(+2 pi i/(M W)), zeros between;
        //            IDFT_{M W/2}(Z) read as floats IS the upsampled row.
        constexpr int NK = (WH + 63) / 64;
        v2f zk[NK], zp[NK];   // (band-pass: zk[t] = X[lane + 64 t] itself)
        if constexpr (BP) {
          // BscanDark.cpp:218-236 blanks the shifted spectrum's outer 40 % on both sides and 3 bins either side of DC: of the bins
          // that survive the Hermitian read, 3 <= k < floor(W/10) remain.  X[k] = (1/W) sum_m x[m] e^(-2 pi i k m / W) for those,
          // in double (DftBinF64, fdoct_fft_reg.h): a lane owns bin 3 + (lane mod KP) (+ 64 b) and one of G = 64 / KP slices of the
          // samples (every lane of a slice reads the same x[m]); slices add by shuffles.
          constexpr int KB = W / 10 - 3 > 0 ? W / 10 - 3 : 0;
#pragma unroll
          for (int t = 0; t < NK; t++) zk[t] = mk(0.f, 0.f);
          if constexpr (KB > 0) {
            constexpr int KP = KB >= 64 ? 64 : wave_pow2ceil(KB), G = 64 / KP, NB = (KB + 63) / 64, MS = (W + G - 1) / G;
            const int jl = lane & (KP - 1), m0 = (lane / KP) * MS;
            const double* const xd = reinterpret_cast<const double*>(buf);
            constexpr double inv_wd = 1.0 / (double)W;
            constexpr int T = NB > 1 ? 4 : 8;
            DftBinF64<T> bin[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) bin[b].init(3 + jl + 64 * b, m0, W);
            const int m1 = m0 + MS < W ? m0 + MS : W;
            for (int m = m0; m < m1; m += T) {
              double x[T];
#pragma unroll
              for (int t = 0; t < T; t++) x[t] = m + t < m1 ? xd[m + t] : 0.0;
#pragma unroll
              for (int b = 0; b < NB; b++) bin[b].chunk(x);
            }
            double ar[NB], ai_[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) {
              ar[b] = bin[b].ar;
              ai_[b] = bin[b].ai;
            }
#pragma unroll
            for (int s = KP; s < 64; s <<= 1) {
#pragma unroll
              for (int b = 0; b < NB; b++) {
                ar[b] += __shfl_xor(ar[b], s, 64);
                ai_[b] += __shfl_xor(ai_[b], s, 64);
              }
            }
            wave_fence();   // (every lane has read its samples: the head of the buffer takes the bins)
#pragma unroll
            for (int b = 0; b < NB; b++) {
              const int j = jl + 64 * b;
              if (lane < KP && j < KB) buf[j] = mk((float)(ar[b] * inv_wd), (float)(ai_[b] * inv_wd));
            }
            wave_fence();
#pragma unroll
            for (int t = 0; t < NK; t++) {
              const int k = lane + 64 * t;
              if (64 * t < W / 10 && k >= 3 && k < W / 10) zk[t] = buf[k - 3];
            }
            wave_fence();
          }
        } else {
          wave_fft<WH, false, false, false>(buf, tw_wh, lane, nullptr, 0, 0);
          FDOCT_PR(1);   // forward W/2-point transform
#pragma unroll
          for (int t = 0; t < NK; t++) {
            const int k = lane + 64 * t;
            if ((WH % 64) == 0 || k < WH) {
              zk[t] = buf[k];
              zp[t] = buf[k == 0 ? 0 : WH - k];
            }
          }
          wave_fence();
        }
        constexpr float inv_w = 1.f / (float)W;
#pragma unroll
        for (int t = 0; t < NK; t++) {
          const int k = lane + 64 * t;
          if ((WH % 64) == 0 || k < WH) {
            float xx, xy;
            if constexpr (BP) {
              xx = zk[t].x;
              xy = zk[t].y;
            } else {
              const float ax = zk[t].x + zp[t].x, ay = zk[t].y - zp[t].y, bx = zk[t].x - zp[t].x, by = zk[t].y + zp[t].y;
              const v2f tc = tw_w[k];                                                          // e^(+2 pi i k/W): its conjugate is needed
              const float qx = fmaf(tc.y, by, tc.x * bx), qy = fmaf(-tc.y, bx, tc.x * by);     // q = conj(t) * B
              xx = 0.5f * (ax + qy) * inv_w;
              xy = (k == 0) ? 0.f : 0.5f * (ay - qx) * inv_w;
            }
            const v2f w = tw_mw[k];
            const float px = fmaf(-xy, w.y, xx * w.x), py = fmaf(xy, w.x, xx * w.y);         // X * w
            buf[k] = mk(xx - py, xy + px);                                                   // X (1 + i w)
            if (k > 0) {
              const float cx = xx, cy = -xy;                                                 // c = conj X[k], w' = (-w.x, w.y)
              const float q2x = fmaf(-cy, w.y, cx * -w.x), q2y = fmaf(cy, -w.x, cx * w.y);   // c * w'
              buf[LH - k] = mk(cx + q2y, cy - q2x);                                          // c (1 - i w')
            }
          }
        }
        // zeros at WH .. LH - WH -- except the whole input blocks of the inverse transform's first pass that lie inside: that
        // pass does not read them (wave_pass, ZLO / ZHI), so they are not written either
        constexpr int ZLO = FDOCT_WAVE_ZPRUNE ? WH : 0, ZHI = FDOCT_WAVE_ZPRUNE ? LH - WH : -1;
        constexpr int ZB = wave_zero_run_begin(LH, ZLO, ZHI), ZE = wave_zero_run_end(LH, ZLO, ZHI);   // ZB == ZE: no such block
        constexpr int NZ1 = (ZE > ZB ? ZB : LH - WH + 1) - WH;   // WH .. ZB - 1 (or the whole band)
#pragma unroll
        for (int t = 0; t < (NZ1 + 63) / 64; t++) {
          const int k = WH + lane + 64 * t;
          if (k < WH + NZ1) buf[k] = mk(0.f, 0.f);
        }
        if constexpr (ZE > ZB) {
          constexpr int NZ2 = LH - WH + 1 - ZE;                  // ZE .. LH - WH
#pragma unroll
          for (int t = 0; t < (NZ2 + 63) / 64; t++) {
            const int k = ZE + lane + 64 * t;
            if (k <= LH - WH) buf[k] = mk(0.f, 0.f);
          }
        }
        wave_fence();
        FDOCT_PR(2);   // spectrum re-packing
        wave_fft<LH, true, false, false, PADF ? SPL / 2 : 0, 0, 0, ZLO, ZHI>(buf, tw_lh, lane, nullptr, 0, 0);
        FDOCT_PR(3);   // inverse M W/2-point transform
      }

      // ---- A5 (first half): s_i = y_i + g_i (y_i - y_(i-1)) on the (upsampled) row, in place; lane l owns the SPL
      // consecutive samples l*SPL .. (the reference weights by fractionalk[nearestkindex[q]], a per-SAMPLE quantity)
      {
        const float* src = bf + lane * SPLP;
        const float* gs = s_g + lane * SPLP;
        float* dst = bf + lane * SPLP;
        // the left neighbour of this lane's first sample is the LAST sample of lane - 1: fetched before anything is
        // overwritten (wave_shr:1; lane 0 has none: slopes[0] = slopes[1], main:1161)
        const float mylast = src[SPL - 1], y1 = src[1];
        wave_fence();
        float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mylast), 0x138, 0xf, 0xf, false));
        // samples per step: a divisor of SPL where there is one (every step alike), else steps of 8 and a shorter last one
        constexpr int CH = SPL % 10 == 0 ? 10 : (SPL % 9 == 0 ? 9 : (SPL % 8 == 0 ? 8 : (SPL <= 12 ? SPL : 8)));
        // RAGGED rows: this lane owns samples lane SPL .. lane SPL + mine - 1 (a lane past the end of the row none); what it
        // reads beyond them lies inside the workgroup's LDS and is not stored
        const int mine = RAGGED ? (MW - lane * SPL < 0 ? 0 : (MW - lane * SPL < SPL ? MW - lane * SPL : SPL)) : SPL;
        static_for<0, (SPL + CH - 1) / CH>([&](auto cc) {
          constexpr int c0 = decltype(cc)::value * CH;
          constexpr int CN = SPL - c0 < CH ? SPL - c0 : CH;
          float yy[CN], gg[CN];
#pragma unroll
          for (int c = 0; c < CN; c++) {
            yy[c] = src[c0 + c];
            if constexpr (RESG)
              gg[c] = g_res[c0 + c];
            else
              gg[c] = gs[c0 + c];
          }
#pragma unroll
          for (int c = 0; c < CN; c++) {
            float slope = yy[c] - (c == 0 ? prev : yy[c - 1]);
            if (c0 + c == 0) slope = (lane == 0) ? (y1 - yy[0]) : slope;
            if (!RAGGED || c0 + c < mine) dst[c0 + c] = fmaf(gg[c], slope, yy[c]);
          }
          prev = yy[CN - 1];
          if constexpr (c0 + CH < SPL) __builtin_amdgcn_sched_barrier(0);
        });
        if (lane == 0) bf[MWP] = 0.f;  // source of data_ylin[0] and data_ylin[N-1] (never written by the reference: 0)
      }
      wave_fence();
      FDOCT_PR(4);   // slope step

      // ---- A5 (second half) + A6: the gather fills the first pass's registers: FFT point e packs
      // (data_ylin[2e], data_ylin[2e+1]); element (lane + 64 t) + r*NB0 goes to zin[t*R0 + r]
      v2f zin[NBL0 * R0];
      static_for<0, NBL0>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const int j = lane + 64 * t;
        if (FULL0 || j < NB0) {
          static_for<0, R0>([&](auto rc) {
            constexpr int rr = decltype(rc)::value;
            uint32_t gi;
            if constexpr (RESGI)
              gi = gi_res[t * R0 + rr];
            else
              gi = s_gi[j + rr * NB0];
            if constexpr (CPLX) {
              const float y = bf[gi & 0xffffu];                    // data_ylin[e] times its phasor, e = j + rr NB0
              const v2f ph = s_ph[j + rr * NB0];
              zin[t * R0 + rr] = mk(y * ph.x, y * ph.y);
            } else {
              zin[t * R0 + rr] = mk(bf[gi & 0xffffu], bf[gi >> 16]);
            }
          });
        }
      });
      wave_fence();
      FDOCT_PR(5);   // gather
      // ---- A7: N/2-point inverse DFT of the packed row; the last pass keeps what the untangle reads (complex rows: the
      // N-point transform of the row itself, bins below numdisplaypoints kept)
      // (D <= 64 TD: blocks of the last pass that no depth this kernel serves reads are not computed)
      constexpr int DK0 = 64 * TD < NC ? 64 * TD : NC, DK = (DKP > 0 && DKP < DK0) ? DKP : DK0;
      wave_fft<NC, true, true, true, 0, DK, (CPLX || DEEP) ? NC : NC - DK>(buf, tw_nc, lane, zin, (CPLX || DEEP) ? (D < NC ? D : NC) : D, (CPLX || DEEP) ? NC : NC - D);

      FDOCT_PR(6);   // final transform
      // ---- A8: untangle X[k] = (A - i w^k B)/2, A = Z[k] + conj Z[N/2-k], B = Z[k] - conj Z[N/2-k], magnitude
#pragma unroll
      for (int t = 0; t < TD; t++) {
        const int b = lane + 64 * t;
        if constexpr (CPLX) {
          if (b < D) {
            const v2f z = buf[b];
            acc[t] += fast_sqrt(fmaf(z.x, z.x, z.y * z.y));   // main:1190
          }
        } else if (b < D) {
          // (DEEP: bins above N/2 mirror, |X[b]| = |X[N - b]|; bin N/2 pairs Z[0] with itself)
          const int k = DEEP ? (b <= NC ? b : N - b) : b;
          const v2f zkk = buf[(DEEP && k == NC) ? 0 : k];
          const v2f zpp = buf[(k == 0 || (DEEP && k == NC)) ? 0 : NC - k];
          const v2f w = tw_n[k];
          const float ax = zkk.x + zpp.x, ay = zkk.y - zpp.y, bx = zkk.x - zpp.x, by = zkk.y + zpp.y;
          const float qx = fmaf(-w.y, by, w.x * bx), qy = fmaf(w.y, bx, w.x * by);
          const float xr = ax + qy, xi = ay - qx;
          acc[t] += 0.5f * fast_sqrt(fmaf(xr, xr, xi * xi));
        }
      }
      wave_fence();
      FDOCT_PR(7);   // untangle, magnitude
    }
#ifdef FDOCT_WAVE_PROBE
    pr_t = __builtin_readcyclecounter();
#endif

    // ---- A9/A10: average, epsilon, dB (2.303), DC mask; bins lane + 64 t: coalesced stores
    float* om = a.out_mag ? a.out_mag + (size_t)o * D : nullptr;
    float* od = a.out_db ? a.out_db + (size_t)o * D : nullptr;
    float db4 = 0.f;
    if (od && a.dcmask && D > 4) db4 = a.db_scale * fast_log2(fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc[0]), 4)), a.inv_A, a.eps));
#pragma unroll
    for (int t = 0; t < TD; t++) {
      const int b = lane + 64 * t;
      if (b < D) {
        const float v = fmaf(acc[t], a.inv_A, a.eps);
        if (om) __builtin_nontemporal_store(v, om + b);
        if (od) __builtin_nontemporal_store((a.dcmask && D > 4 && b < 2) ? db4 : a.db_scale * fast_log2(v), od + b);
      }
    }
    FDOCT_PR(8);     // epilogue
  }
#ifdef FDOCT_WAVE_PROBE
  if (a.probe && lane == 0 && blockIdx.x < 4 && wave < 16) {
    for (int i = 0; i < 9; i++) a.probe[(blockIdx.x * 16 + wave) * 12 + i] = pr_acc[i];
  }
#endif
}

}  // namespace fdoct
