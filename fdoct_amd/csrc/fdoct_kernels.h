// fdoct_kernels.h -- host/device interface between the C-ABI layer and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#ifndef FDOCT_MAX_BLOCK
#define FDOCT_MAX_BLOCK 768
#endif

namespace fdoct {

// Threads per workgroup the fused kernel is compiled for (register budget = 512 / (threads/256) VGPRs per
// lane).  Plans with 32 FFT points per lane hold twice the per-lane state (the 2048-point ones are LDS-limited to
// <= 8 waves per CU anyway),
// the general (predicated, every-option) kernel carries more live state than the fast-path one, and the
// fast-path row-swap plan (kind 1) keeps its row-invariant tables in registers: all of these trade
// occupancy for registers instead of spilling.
constexpr int fused_max_block(int nc, int T, bool lean, int kind) {
#ifdef FDOCT_X_BLOCK  // tuning: threads per workgroup of the fast-path row-swap plan (768 = 3 waves per SIMD, <= 168 VGPRs)
  if (lean && kind == 1) return FDOCT_X_BLOCK;
#endif
  return (nc / T >= 32 || !lean || kind == 1) ? 512 : FDOCT_MAX_BLOCK;  // nc/T = FFT points held per lane
}

// Fast-path kernels of the row-swap plans keep the per-column constants (1/background, window, slope weights) in
// registers; the host then leaves those three planes out of the workgroup's LDS (FusedArgs::lds_planes = 0), which is
// what lets the 2048-point plans run 7 instead of 5 waves per CU.  One definition for kernel and host.
constexpr bool fused_resident_consts(int kind, bool lean, bool avg, int wch, int stage) {
#ifdef FDOCT_X_NO_RESC  // tuning: constant planes from LDS on the 1024-point plan
  if (kind == 1) return false;
#endif
  return lean && (kind == 1 || (kind == 2 && !avg)) && wch <= 4 && stage != 2;
}

enum { FDOCT_K_U8 = 0, FDOCT_K_U16 = 1, FDOCT_K_F32 = 2 };

// The fast-path kernels exist in two instantiations: multiplying by one word of the reciprocal background or by both
// (fdoct_capi.cpp::reciprocal_words; fdoct_set_precise_division).  The any-option kernel always uses both.
// FDOCT_PREC_T2: how many of the 12 step-3 twiddles of the 1024-point plan stay in registers in the kernels that can
// multiply by both words (the rest come from LDS every row: the low words' 32 registers are in flight at the row top).
#ifndef FDOCT_PREC_T2
#define FDOCT_PREC_T2 6
#endif
// FDOCT_PREC16: the form of the second word on the fast-path kernels with at most 32 samples per lane (fused_kernel's ILX).
// The missing part of the quotient, v * il, is (v * ib) * (il / ib) = (c0 + d) * rho with rho = il / ib, |rho| <= 2^-24, and
// d * rho lies below the rounding of d itself: the correction is c0 * rho_i -- a row-dependent scalar times a column-dependent
// pattern that needs no more than ~10 bits.  1: the pattern is a plane of HALF floats (rho * 2^38: 2 W bytes of LDS, 16
// registers in flight instead of 32, every step-3 twiddle resident again) applied by v_fma_mix_f32, which converts its f16
// operand inside the fma.  0: round 4's form, il as floats multiplied by the samples.
#ifndef FDOCT_PREC16
#define FDOCT_PREC16 1
#endif
#ifndef FDOCT_PREC16_T2
#define FDOCT_PREC16_T2 10       // step-3 twiddles resident in the half-float form (16 registers in flight at the row top): 516 M A-scans/s with 10 or 8, 513 with 12
#endif
#ifndef FDOCT_PREC16_T2_IB2D
#define FDOCT_PREC16_T2_IB2D 0   // ... with a full-frame background (16 more registers hold the next row's pattern): 443 against 422 M A-scans/s with 4
#endif
#ifndef FDOCT_PREC16_T2_DMA
#define FDOCT_PREC16_T2_DMA 8    // ... with a full-frame background whose pattern row is prefetched into LDS (FDOCT_IL16_DMA): 478 M A-scans/s against 438 with 12 (spills) and 442 with the pattern row in registers
#endif
#ifndef FDOCT_IL16_RESIDENT
#define FDOCT_IL16_RESIDENT 1    // 1: the averaging kernels with more than 32 samples per lane keep the half-float pattern in registers
#endif
#ifndef FDOCT_IL16_DMA
#define FDOCT_IL16_DMA 1         // 1: a full-frame background's pattern row is prefetched into LDS by global_load_lds_dwordx4 (no registers)
#endif
// LDS bytes per computing wave of that prefetch slot (one definition for kernel and host)
constexpr size_t fused_il16_dma_bytes(bool ib2d, bool both_words_half, bool tro, int wc) { return (FDOCT_IL16_DMA && ib2d && both_words_half && !tro) ? (size_t)2 * wc : 0; }
#ifndef FDOCT_TRO_IB2D_RES3
#define FDOCT_TRO_IB2D_RES3 0    // 1: the transposed-store variant of that kernel keeps its 15 step-5 twiddles in registers (spills)
#endif
constexpr int kPrec16Shift = 38;  // rho * 2^38: at most 2^14 in magnitude
constexpr bool fused_il_half(bool lean, int wch) { return FDOCT_PREC16 != 0 && lean && wch <= 4; }
// The averaging fast-path kernels with more than 32 samples per lane keep their planes in LDS and are bound by its capacity (a
// fourth plane would cost C4 a wave per CU): they read the low words from a global plane in the same order (FusedArgs::prec = 3).
// (round 6: so do ALL fast-path kernels of the 512-point plan -- C1, 16 lanes per row, four rows per wave -- averaging or not, row-major
// or transposed store: the 4 W bytes of the plane are a sixth computing wave next to the transposed store's ring, and both
// layouts apply the second word in the same form, so their images stay bit-identical)
constexpr bool fused_il_global(bool lean, bool avg, int wch, int T = 64) { return lean && (avg || T == 16) && wch > 4; }

// Rows per tile of the fused transposed store: a workgroup owns FUSED_TR_ROWS consecutive A-scans of one B-scan at a time, so
// the depth-major output is written in segments of FUSED_TR_ROWS * 4 bytes.
#ifndef FUSED_TR_ROWS
#define FUSED_TR_ROWS 16
#endif
// Slots of the LDS ring of finished rows (FUSED_TR_ROWS < slots <= 2 FUSED_TR_ROWS): what 160 KB of LDS hold of 1024-bin
// rows next to seven computing waves' buffers.
#ifndef FUSED_TR_RING
#define FUSED_TR_RING 20
#endif
// Who writes a complete tile out: 1 = every wave takes steps of it at its hand-over points (all waves compute); 2 = the wave
// whose row completes it, all steps at once (all waves compute; one LDS round trip per row); 0 = the last wave of the
// workgroup does nothing else (one wave less computes).  Measured (profiles/r03_tro_final_probe.txt): at 1024
// depth bins 1 is 0-4 % ahead of 2, up to 512 bins (40-slot ring) 2 is 4 % ahead of 1; 1 ships because the reference's
// configurations display 1024 bins.
#ifndef FDOCT_TRO_DW
#define FDOCT_TRO_DW 1
#endif
constexpr int fused_tro_writer_waves() { return FDOCT_TRO_DW ? 0 : 1; }
// Ring slots for numdisplaypoints = d: up to 512 depth bins a second tile fits (rows of the next tile go in while a tile is
// written out: + 7 %), above that FUSED_TR_RING is what the LDS holds.  One definition for kernel and host.
constexpr unsigned fused_tro_ring_slots(int d) { return d <= 512 ? 2u * FUSED_TR_RING : (unsigned)FUSED_TR_RING; }
// LDS bytes of the ring (a slot is d + 4 floats).
constexpr size_t fused_tro_ring_bytes(int d) { return (size_t)fused_tro_ring_slots(d) * (size_t)(d + 4) * 4; }
// Round 5: the ring takes what the LDS has left.  The transposed store is bound by how much of the next tile fits into the
// ring while a tile drains, so its kernels keep out of LDS what they only read once -- the step-5 twiddle
// table (7.7 KB: those kernels hold its 15 entries per lane in registers) and, without averaging, the gather table (4 KB:
// the addresses are resident too) -- and the launch picks the LARGEST ring of this list that fits next to eight computing
// waves (23 slots at 1024 depth bins, 22 with the half-float plane of the second word; the moduli are compile-time constants
// of the kernel: FusedArgs::tr_ring selects one).  Any value above FUSED_TR_ROWS works for the hand-over protocol; at most
// 3 FUSED_TR_ROWS, so that no more than four tiles are open at once (tr_arrived / tr_done are indexed by tile mod 4).
constexpr unsigned kTroRingChoices[] = {20, 21, 22, 23, 24, 26, 28, 32, 40, 44, 48};
// (rpw: rows per wave of the plan -- a wave's rows take consecutive slots, so the ring is a whole number of them)
constexpr unsigned fused_tro_ring_pick(size_t lds_left, int d, int rpw = 1) {
  unsigned best = 0;
  for (unsigned c : kTroRingChoices)
    if ((c % (unsigned)rpw) == 0 && (size_t)c * (size_t)(d + 4) * 4 <= lds_left) best = c;
  return best;
}
// Which tables a fused kernel stages in LDS (one rule for kernel and host).  tw3: the step-5 table of the 1024-point row-swap plan.
// (ib2d_both_words: the variant with a full-frame background and both words re-reads its step-5 twiddles every row: FDOCT_TRO_IB2D_RES3)
constexpr bool fused_tw3_in_lds(int kind, bool lean, int stage, bool tro, bool ib2d_both_words) {
  return !(tro && lean && kind == 1 && stage != 1 && !(ib2d_both_words && !FDOCT_TRO_IB2D_RES3));
}
// Transposed store, row-swap plan, 16-bit samples, no per-row prefetch besides the samples (one-spectrum background, no frame
// normalisation): the samples are prefetched TWO rows ahead (fused_kernel, PF2) -- a wave's vector-memory operations return in
// order, so with one row of distance the prefetched samples wait behind the write-out stores issued a row earlier, and those take
// 2-8 us to be acknowledged (EXPERIMENTS.md section 5).  The second set of sample registers takes the place of the resident gather
// addresses: the gather table goes back to LDS in this variant.
#ifndef FDOCT_TRO_PF2
#define FDOCT_TRO_PF2 0
#endif
constexpr bool fused_tro_pf2(int kind, bool lean, int stage, bool cplx, bool avg, bool tro, bool ib2d, int norm, int sample_bytes) {
  return FDOCT_TRO_PF2 && tro && lean && stage == 0 && kind == 1 && !cplx && !avg && !ib2d && norm == 0 && sample_bytes == 2;
}
constexpr bool fused_gi_in_lds(int kind, bool lean, int stage, bool cplx, bool avg, bool tro, bool pf2 = false) {
  return pf2 || !(tro && lean && stage != 2 && kind == 1 && !cplx && !avg);
}
#ifndef FDOCT_TRO_SPIN_LIMIT
#define FDOCT_TRO_SPIN_LIMIT (1u << 21)  // x s_sleep(8) = 512 cycles each: about half a second
#endif
// Depth bins one iteration of the tile write-out covers (numdisplaypoints must be a multiple of it).
// Four rows per wave (the 512-point plan's in-place tiles): waves per group = rows per tile / 4.  Groups of EIGHT waves own tiles of
// 32 rows and write the D x H image in 128-byte segments -- whole cache lines, which the memory system takes at its row-major rate
// where 64-byte segments stop at 3.7 TB/s (profiles/r06_rw_mix.txt) -- at no cost in LDS (the rows lie in the waves' own buffers).
// Built and measured (round 6, profiles/r06_c1_group_ab.txt, bit-identical results): 790 against 855 M A-scans/s on C1 -- a workgroup
// that is ONE group meets twice per tile with all of its waves, and what they idle there outweighs the segments.  Four it stays.
#ifndef FDOCT_TRO_GROUP_WAVES
#define FDOCT_TRO_GROUP_WAVES 4
#endif
constexpr int fused_tro_group_waves() { return FDOCT_TRO_GROUP_WAVES; }
constexpr int fused_tro_tile_rows(int rpw) { return rpw == 4 ? 4 * FDOCT_TRO_GROUP_WAVES : FUSED_TR_ROWS; }
constexpr int fused_tro_step_bins(int rpw = 1) { return 4 * (64 / (fused_tro_tile_rows(rpw) / 4)); }
// Which plans have the fused transposed store compiled (the fast-path row-swap 1024-point plan, one row per wave).
// Round 6: also the 512-point Stockham plan (C1: 1024 samples -> numfftpoints 1024; 16 lanes per row, FOUR rows per wave) --
// without the ring: tiles are owned by groups of four waves and the finished rows wait in the waves' own row buffers
// (fused_kernel, TRO_INPLACE).
constexpr bool fused_tro_compiled(int kind, int T, int wch) { return (kind == 1 && T == 64 && wch <= 4) || (kind == 0 && T == 16 && wch == 8); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember what has been granted
// on each device (one `LdsGrant` per kernel; the attribute only ever needs to grow).
struct LdsGrant {
  size_t granted[32] = {};
  std::mutex mu;  // handles on different devices may launch the same kernel from different host threads
  template <typename K>
  hipError_t ensure(K kernel, size_t lds) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& g = granted[dev & 31];
    if (lds > g) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      g = lds;
    }
    return hipSuccess;
  }
};

constexpr int FUSED_PROBE_PHASES = 12;
// Arguments of the fused kernel.  All pointers are device pointers.
struct FusedArgs {
  const void* frames;        // camera samples, row pitch in bytes
  const float* frames_lo;    // data_y handed over as doubles (main:987): frames = the f32 high words, this = the low words (same pitch), or null
  long long pitch_bytes;
  long long total_out_rows;  // groups * H
  int W, H, D, A;            // samples/row, rows/frame, output bins, frames averaged per output
  int need_rc;               // kernel must know (group,row): 2-D reference frames or min-max scalars
  int split;                 // staging layout: 1 = even/odd sample planes (see gather)
  int scratch_bytes;         // LDS bytes per row in flight
  int tw_count;              // entries in tw
  int lds_planes;            // 1: the three constant planes are staged in LDS; 0: resident-constant kernel, planes left out
  const float* ib;           // [W] 1/background (1-row mode) or null
  const float* ib2d;         // [H*W] 1/background (2-D mode) or null
  const float* il;           // [W] low word of 1/background (fdoct_capi.cpp::reciprocal_words), 1-row mode, or null
  const float* il2d;         // [H*WC] the same for the 2-D mode, laid out like ib2d
  const float* ilp;          // [WC] the 1-row low words in the order of the kernels' LDS planes (prec == 3)
  const uint32_t* il16;      // [WC/2] fused_il_half kernels: rho = il / ib scaled by 2^38 as half-float pairs, in the order the lanes read them
  const uint32_t* il16_2d;   // [H*WC/2] the same for a full-frame background (rows in frame order)
  int prec;                  // 0: one word (fast path without fdoct_set_precise_division); 1: both words, low words staged in LDS;
                             // 2: both words, full-frame background (il2d); 3: both words, low words read from ilp in global memory
  const float* yp; int yp_2d;  // pi frame or null
  const float* yd; int yd_2d;  // dark frame or null
  const float* win;          // [W] plane a_i = (1 + g_i) w_i  (window and slope weight folded; a_0, b_0: see fdoct_capi.cpp)
  const float* g;            // [W] plane b_i = -g_i w_(i-1), g = fractionalk indexed by sample
  const uint32_t* gidx;      // [NC] packed LDS byte offsets of the gather sources
  const float2* tw;          // Stockham twiddle tables
  const float2* utw;         // [T] exp(2*pi*i*l/N) (real path)
  const float2* phase;       // [N] dispersion phasors (complex path) or null
  const float2* minmax;      // [frames] whole-frame (min,max) or null
  int rowwisenormalize;
  int dcmask;
  int stage;                 // 0 = fused chain, 1 = resample stage (-> ylin), 2 = FFT stage (ylin ->)
  float2* ylin;              // [rows*NC] packed k-linear rows between the stages (staged mode only)
  int ablate;                // profiling aid: bit mask of stages to skip (results are then wrong); 0 in production
  float inv_A, eps, db_scale;
  float* out_mag;            // [groups*H*D] linear (bscan, row-major) or null
  float* out_db;             // [groups*H*D] dB or null
  // Transposed output written by the chain itself (TRO kernels): out_mag / out_db are then [groups][D][H] (the reference's
  // bscan layout, main:1220); finished rows go through a ring in LDS, tiles of FUSED_TR_ROWS rows (see fused_kernel)
  int tro;                   // 1: launch the TRO instantiation
  unsigned tr_ring;          // slots of the LDS ring of finished rows (one of kTroRingChoices)
  unsigned tr_tpf;           // tiles per frame = ceil(H / FUSED_TR_ROWS)
  unsigned tr_tpf_magic;     // floor(2^32 / tr_tpf)
  unsigned tr_total_tiles;   // groups * tr_tpf
  unsigned* tr_fault;        // one word of pinned host memory, set to 1 (a plain system-scope store) if a wave gave up waiting for a tile buffer (never, unless the protocol is broken)
#ifdef FDOCT_CLOCKPROBE
  unsigned long long* probe;  // tuning aid: {shader cycles, 100 MHz ticks} one wave spent in the kernel
#endif
#ifdef FDOCT_FUSED_PROBE
  unsigned long long* phase_probe;  // measurement build: cycles per phase of the row loop, [workgroup < 4][wave < 16][FUSED_PROBE_PHASES]
#endif
};

// Arguments of the any-configuration kernel (fdoct_generic.hip).  All pointers are device pointers.
constexpr int GENERIC_MAX_PASSES = 16;
#ifndef GENERIC_MAX_RADIX
#define GENERIC_MAX_RADIX 8  // largest power-of-two butterfly of the generic kernel (8 or 16)
#endif
// One in-LDS +i DFT of any length for generic_kernel's full-length zero-pad stage: Stockham radices of the length itself
// (blu_m == 0; tw = exp(+2 pi i j / n)) or, for a length with a prime factor above 5, Bluestein around two blu_m-point
// transforms (radices and tw of blu_m; chirp[n], bhat[blu_m] as fdoct_state.cpp::build_bluestein_tables makes them).
struct GenericDft {
  int n, blu_m, npass;
  int rad[GENERIC_MAX_PASSES];
  unsigned mag[GENERIC_MAX_PASSES];
  const float2 *tw, *chirp, *bhat;
};

struct GenericArgs {
  const void* frames;
  const float* frames_lo;    // low words of f64 frames (frames = their f32 high words, same pitch) or null
  long long pitch_bytes;
  long long total_out_rows;
  int dtype;                 // FDOCT_K_*
  int W, H, N, D, M, A;
  int L;                     // max(N, M*W, W): length of each DFT ping-pong buffer
  int ybuf_len;              // floats reserved for the row buffer (>= max(W, M*W), multiple of 4)
  const float* ib; int ib_2d;
  const float* il;           // low word of 1/background, indexed like ib (fdoct_capi.cpp::reciprocal_words)
  const float* yp; int yp_2d;
  const float* yd; int yd_2d;
  const float* win;          // [W] window (unscaled)
  const float* win_lo;       // [W] what the float window leaves of the double one (the band-pass forms the row in double)
  const float *yp_lo, *yd_lo;  // the same of the pi and dark frames (laid out like yp, yd)
  const float* g;            // [M*W] fractionalk indexed by sample (0 past numfftpoints)
  const int32_t* idx;        // [N] nearestkindex
  const float2* phase;       // [N] or null
  const float2* minmax;      // per input frame (min,max) or null
  const float2 *tw_n, *tw_w, *tw_mw;  // exp(+2*pi*i*j/n) for n = N, W, M*W (the last two only when M > 1)
  int rad_n[GENERIC_MAX_PASSES];
  unsigned mag_n[GENERIC_MAX_PASSES];  // ceil(2^32 / Ns) per pass
  int npass_n;
  // zero-pad upsampling (M > 1): the row is real and its padded spectrum Hermitian, so both DFTs run at half length
  const float2 *tw_wh, *tw_mwh;        // exp(+2*pi*i*j/n) for n = W/2, M*W/2
  int rad_wh[GENERIC_MAX_PASSES], rad_mwh[GENERIC_MAX_PASSES];
  unsigned mag_wh[GENERIC_MAX_PASSES], mag_mwh[GENERIC_MAX_PASSES];
  int npass_wh, npass_mwh;
  // zero-pad upsampling at FULL length (round 6): odd widths (the reference's fftshift leaves the last spectrum column in place
  // and an even multiplier pads to M W - 1 bins, main:215-241) and widths whose half-length transforms have a prime factor
  // above 5 -- W-point +i transform of the row, re-packing by pad_source's rule, zn-point +i transform, real parts.  In LDS as
  // long as two buffers of max(these transforms' lengths) fit; beyond that the long-row path (fdoct_big.hip) keeps the rows in HBM.
  int zp_full, zn;           // zn = W + 2 floor((M W - W) / 2), the padded spectrum's length
  GenericDft zf, zi;         // the W-point and the zn-point transform
  int radix16;               // 1: the pass plans hold radix-16 butterflies (the 1024-thread kernels only)
  int inplace;               // 1: ONE DFT buffer of L values (rows whose two buffers do not fit the LDS): generic_kernel<1024, 1, true>
  int bandpass;              // BscanDark.cpp:218-236 inside the zero-pad: keep spectrum bins 3 <= k < floor(W/10) only
  // real rows (no dispersion phase, even N): the N-point DFT is done as an N/2-point complex DFT + untangle
  int real_half;
  const float2* tw_nh;  // exp(+2*pi*i*j/(N/2))
  int rad_nh[GENERIC_MAX_PASSES];
  unsigned mag_nh[GENERIC_MAX_PASSES];
  int npass_nh;
  // Bluestein (numfftpoints with a prime factor above 5): the final +i transform of length n (N, or N/2 for real rows) as
  // chirp multiply -> forward DFT_Mb -> multiply by bhat -> inverse DFT_Mb -> chirp multiply; blu_m = Mb >= 2n - 1 (0: off)
  int blu_m;
  const float2 *blu_chirp, *blu_bhat, *tw_blu;  // e^(+i pi m^2/n), m < n; DFT_Mb(wrapped conj chirp)/Mb; e^(+2 pi i j/Mb)
  int rad_blu[GENERIC_MAX_PASSES];
  unsigned mag_blu[GENERIC_MAX_PASSES];
  int npass_blu;
  int rowwisenormalize, dcmask;
  float inv_A, eps, db_scale;
  float* out_mag;
  float* out_db;
  unsigned* row_ticket;      // launch-wide row counter, zero at launch (rows beyond the workgroups' first are claimed from it), or null
};

hipError_t launch_generic(const GenericArgs& a, int grid, size_t lds, hipStream_t st);
hipError_t launch_movavg(const void* frames, int dtype, long long pitch_bytes, int W, long long rows, int n, float* out,
                         hipStream_t st);
// smoothmovavg of f64 frames: tap sums in double, out as two f32 planes (hi + lo)
hipError_t launch_movavg_f64(const double* in, long long pitch_elems, int W, long long rows, int n, float* hi, float* lo, hipStream_t st);
// ... of f32 frames (their samples need not be integers: a float tap sum would round at the size of the DC level)
hipError_t launch_movavg_f32_wide(const float* in, long long pitch_elems, int W, long long rows, int n, float* hi, float* lo, hipStream_t st);

hipError_t launch_median(const void* in, long long in_pitch, void* out, long long out_pitch, int dtype, int w, int h, int n,
                         int nframes, hipStream_t st);
hipError_t launch_bin(const void* in, long long in_pitch, void* out, long long out_pitch, int dtype, int ow, int oh, int binx,
                      int biny, int nframes, hipStream_t st);

// display post-chain (fdoct_display.hip); part: nbscans * display_parts(count) * 2 doubles of scratch
int display_parts(long long count);
hipError_t launch_display(const float* db, long long count, int nbscans, double thr, long long clamp_at, double* part,
                          const unsigned char* lut, unsigned char* gray, unsigned char* bgr, hipStream_t st);
hipError_t launch_lockin_db(const float* bscan, const float* jscan, long long count, long long jcount, float* out,
                            hipStream_t st);

struct FusedPlan {
  int id, nc, T, R1, R2, R3, WCH, kind;
};

int fused_plan_count();
bool fused_plan_get(int id, FusedPlan* p);
// lean = the unpredicated fast-path kernel (see fused_kernel); the caller guarantees its conditions.
hipError_t launch_fused(const FusedPlan& p, const FusedArgs& a, int dtype, bool cplx, bool lean, int grid,
                        int block, size_t lds, hipStream_t st);
// whole-frame min/max; partial: scratch of minmax_partial_count(nframes) float2 (may be null: slow path only)
int minmax_partial_count(int nframes);
hipError_t launch_minmax(const void* frames, int dtype, long long pitch_bytes, int W, int H, int nframes,
                         const float* yd, int yd_2d, float2* out, float2* partial, hipStream_t st);
hipError_t launch_transpose(const float* in, float* out, int rows, int cols, int groups, hipStream_t st);
// data_y as doubles (main:987) -> two f32 planes, hi = fl32(x) and lo = fl32(x - hi): the chain carries both into the division
hipError_t launch_f64_split(const double* in, long long pitch_elems, float* hi, float* lo, int W, long long rows,
                            hipStream_t st);

}  // namespace fdoct
