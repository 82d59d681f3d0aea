// fdoct_kernels.hip -- CDNA4 (gfx950) kernels of the FD-OCT reconstruction path.
//
// One fused kernel replaces the reference's per-frame OpenCV block
// (BscanFFT.cpp:1123-1240 / BscanFFTsim.cpp:842-955): camera samples in,
// B-scan magnitudes (and dB) out, one HBM read and one HBM write per A-scan.
//
// Work decomposition (no workgroup barrier inside the row loop):
//   * a group of T lanes (T = 16/32/64, so 4/2/1 rows per 64-wide wave) owns one
//     output A-scan at a time and loops over rows (persistent waves);
//   * the row's W samples are loaded as 16-byte vectors, 8 samples per lane per
//     chunk, coalesced; normalise / background / DC removal / window /
//     the reference's slope step are done in registers (A2, A3, A5);
//   * the lambda->k gather goes through a per-row LDS staging buffer (A5);
//   * the IDFT keeps P = NC/T complex points per lane, runs radix-R butterflies in
//     registers and exchanges data between passes through a padded LDS buffer --
//     or, for the 1024-point plan, swaps 16-lane rows with v_permlane*_swap for one
//     of the two exchanges (A7);
//   * real input uses the N/2-point complex FFT + untangle; the partner bin
//     lives in lane (T - l) and is fetched with ds_bpermute (no LDS memory);
//   * magnitude, crop, averaging, epsilon, dB and the DC mask are the epilogue
//     (A8-A10); only D floats per A-scan are written.
//   * waves claim their rows from a workgroup ticket counter (the hardware favours the
//     oldest wave of a SIMD; a static split leaves SIMDs half empty at the end);
//   * the fast-path instantiation of the 1024-point plan keeps every row-invariant
//     table (constants, gather addresses, twiddles) in registers, 2 waves per SIMD.
// The chip is power-limited under this kernel and the SIMDs issue-bound, so all complex
// arithmetic is written on 2-element vectors that lower to v_pk_add/mul/fma_f32 (two
// flops per lane per instruction) and quarter turns ride on instruction modifiers.
// MFMA is not used: the path is elementwise + FFT work, not a dense contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "fdoct_fft_reg.h"
#include "fdoct_kernels.h"

#ifdef FDOCT_DEV_ONE  // tuning builds: which single instantiation (see launch_typed)
#ifndef FDOCT_DEV_ONE_CPLX
#define FDOCT_DEV_ONE_CPLX false
#endif
#ifndef FDOCT_DEV_ONE_AVG
#define FDOCT_DEV_ONE_AVG 0
#endif
#endif

// Wait states after each 16-byte store of the tile write-out.  Measured on gfx950: a buffer_store_dwordx4 with an SGPR
// soffset followed directly by a VALU write of its first data register stores the NEW value in the last four lanes of each
// 16-lane row when the memory pipeline is busy (the compiler's hazard recogniser only covers the immediate-offset form of
// this store-data hazard); one wait state cures it, two are used.
#ifndef FDOCT_TRO_X
#define FDOCT_TRO_X 0
#endif
#ifndef FDOCT_TRO_NOP
#define FDOCT_TRO_NOP 2
#endif
// Cache policy of the write-out stores: 0 = write-back.  The two 64-byte halves of a 128-byte line of the B-scan come from
// neighbouring tiles, which tro_tile hands to workgroups of the same XCD: with write-back stores they meet in that L2 and
// leave it as one line.  Measured (C2, M A-scans/s): nt 277, write-back 329, sc1 357-376; with the tile pairing sc1 367-380,
// write-back 395-406.
#ifndef FDOCT_TRO_OUT_AUX
#define FDOCT_TRO_OUT_AUX 0
#endif
// 1: the write-out is shared by all (computing) waves of the workgroup; 0: the last wave only writes out (fdoct_kernels.h)
#ifndef FDOCT_TRO_DW_STEPS
#define FDOCT_TRO_DW_STEPS 1  // write-out steps a wave may take after putting a row into the ring (1: 400, 2: 392, 4: 385 M A-scans/s)
#endif

namespace fdoct {

// Stage-skipping profiling aid (tools/ablate.sh): only in builds with -DFDOCT_RUNTIME_ABLATE.
#ifdef FDOCT_RUNTIME_ABLATE
#define FDOCT_ABL(bit) ((a.ablate & (bit)) != 0)
#elif defined(FDOCT_CT_ABLATE)  // compile-time mask: the skipped stage costs nothing at all (one build per mask)
#define FDOCT_ABL(bit) (((FDOCT_CT_ABLATE) & (bit)) != 0)
#else
#define FDOCT_ABL(bit) false
#endif

// Orders this wave's LDS traffic: a wave's DS operations execute in program
// order, so a compiler-level fence is all a same-wave write->read hand-off needs.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Where a row's cycles go (measurement builds, -DFDOCT_FUSED_PROBE; tools/mkvariant.sh): the cycle counter is read at the phase
// boundaries of the row loop and the differences summed per phase over a wave's rows.  (Each read waits for the wave's
// outstanding LDS / scalar-memory operations -- s_memtime returns through the same counter -- so a phase is charged with the
// latency of what it issued; the build is for attribution, its rate is a few per cent below the shipped kernel's.)
#ifdef FDOCT_FUSED_PROBE
struct FusedProbe {
  unsigned long long acc[FUSED_PROBE_PHASES] = {}, t = 0;
  __device__ __forceinline__ void start() { t = __builtin_readcyclecounter(); }
  __device__ __forceinline__ void mark(int i) {
    // (scheduling barriers either side: without them the compiler moves a phase's arithmetic across the read -- the first
    // build of this probe showed 0.1 % for the radix-16 step)
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long n = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    acc[i] += n - t;
    t = n;
  }
};
#define FDOCT_FPR(p, i) (p).mark(i)
#else
struct FusedProbe {};
#define FDOCT_FPR(p, i) do {} while (0)
#endif

// Exchange-buffer layout: element e lives at slot e + (e >> LP) (one pad slot per
// 2^LP elements, LP = log2 of the first radix).  Unlike an XOR swizzle this is
// additive, so every LDS access below is <one per-lane base VGPR> + <immediate>;
// writes are conflict free for every compiled plan and read-backs are conflict
// free when the first radix is 32 (tests/kernel_model.py checks the banking).
template <int LP>
__device__ __forceinline__ constexpr int padded(int e) {
  return e + (e >> LP);
}

// One Stockham pass over the T-lane group, in three pieces so that the kernel can issue the
// (row-invariant) twiddle reads of a pass behind the previous pass's exchange writes and have a
// single wait cover both:  z[m] = element (l + T*m).
//   pass_twiddles : LDS -> registers, (R-1) twiddles per butterfly
//   pass_compute  : twiddle multiply, radix-R butterflies, exchange writes (or registers if LAST)
//   pass_readback : natural-order read-back of the exchange buffer
template <int NC, int T, int R, int NS>
__device__ __forceinline__ void pass_twiddles(v2f* twr, int l, const v2f* tw) {
  constexpr int P = NC / T;
  constexpr int NB = P / R;
  static_for<0, NB>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const v2f* twk = tw + ((l + T * t) & (NS - 1));
    static_for<1, R>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      twr[t * (R - 1) + (r - 1)] = twk[(r - 1) * NS];
    });
  });
}

template <int NC, int T, int R, int NS, bool LAST, int LP, bool INV>
__device__ __forceinline__ void pass_compute(v2f* z, int l, v2f* xch, const v2f* twr) {
  constexpr int P = NC / T;
  constexpr int NB = P / R;  // butterflies per lane
  static_assert(P % R == 0, "radix must divide the per-lane point count");
  static_assert(NS == 1 || NS >= (1 << LP), "later passes must keep pad-aligned strides");
  static_for<0, NB>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const int j = l + T * t;
    const int k = j & (NS - 1);
    v2f v[R];
    static_for<0, R>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      v[r] = z[t + r * NB];
    });
    if constexpr (NS > 1) {
      static_for<1, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        v[r] = cmul(v[r], twr[t * (R - 1) + (r - 1)]);
      });
    }
    fft_reg<R, INV>(v);
    if constexpr (LAST) {
      static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        z[t + r * NB] = v[r];
      });
    } else {
      const int e0 = (j / NS) * (NS * R) + k;
      v2f* dst = xch + (e0 + (e0 >> LP));
      static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        dst[padded<LP>(r * NS)] = v[r];  // (e0 + r*NS) >> LP == (e0 >> LP) + ((r*NS) >> LP) here
      });
    }
  });
}

template <int NC, int T, int LP>
__device__ __forceinline__ void pass_readback(v2f* z, int l, const v2f* xch) {
  constexpr int P = NC / T;
  if constexpr (T >= (1 << LP)) {
    const v2f* src = xch + (l + (l >> LP));
    static_for<0, P>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      z[m] = src[padded<LP>(T * m)];
    });
  } else {
    const v2f* src = xch + l;  // l < T < 2^LP: the pad term depends on m only
    static_for<0, P>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      z[m] = src[T * m + ((T * m) >> LP)];
    });
  }
}

// 1024-point inverse DFT on a full wave with ONE LDS exchange (plan "16 | row swap | 4 | LDS | 16").
// Index split n = 64*m + 16*a + b (m: register, a: 16-lane row, b: lane in row), output
// k = k1 + 16*k2 + 64*k3:
//   1. radix-16 over m in registers                      -> A[k1][a][b], k1 in the register index
//   2. 4x4 transpose between the lane row a and the low two bits of k1, done with
//      v_permlane32_swap / v_permlane16_swap on register quads (no LDS)
//   3. twiddle W_64^(a*k1), radix-4 over a               -> B[k1][k2][b]
//   4. LDS exchange b <-> (k1,k2): element (b, l') at slot 65*b + l', l' = k1 + 16*k2
//      (stride 65: conflict-free writes; reads are contiguous)
//   5. twiddle W_1024^(b*l'), radix-16 over b            -> X[l' + 64*k3] in register k3 (natural)
// tw2[(3*c + i-1)*4 + j] = W_64^(i*(4c+j)), tw3[(b-1)*64 + l'] = W_1024^(b*l')  (host tables).
// tests/kernel_model.py::fft1024_rowswap_model is the index-for-index numpy model.
__device__ __forceinline__ void swap_rows32(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_rows16(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

// RESTW: the caller keeps the 12 + 15 row-invariant twiddles in registers (rt2 / rt3, loaded with
// fft1024_rowswap_twiddles) instead of re-reading them from LDS for every row.
__device__ __forceinline__ void fft1024_rowswap_twiddles(int lane, const v2f* tw2, const v2f* tw3, v2f* rt2, v2f* rt3) {
  const int j = lane >> 4;
  static_for<0, 12>([&](auto ec) {
    constexpr int e = decltype(ec)::value;
    rt2[e] = tw2[e * 4 + j];
  });
  static_for<1, 16>([&](auto bc) {
    constexpr int bb = decltype(bc)::value;
    rt3[bb - 1] = tw3[(bb - 1) * 64 + lane];
  });
}

template <int RES2, bool RES3>  // RES2: how many of the 12 step-3 twiddles the caller keeps in registers (the first RES2)
__device__ __forceinline__ void fft1024_rowswap(v2f* z, int lane, v2f* xch, const v2f* tw2, const v2f* tw3, const v2f* rt2,
                                                const v2f* rt3, FusedProbe& pr) {
  const int j = lane >> 4, b = lane & 15;
  // 1. radix-16 over the register index
  fft_reg<16, true>(z);
  FDOCT_FPR(pr, 3);
  // Steps 2-4 one register quad at a time (the transposition stays inside a quad).  The quad's three step-3 twiddles come
  // from the caller's registers (the first RES2 of the twelve) or from LDS, read one quad ahead: six registers in flight
  // instead of twenty-four.
  auto quad_twiddles = [&](auto cc, v2f* t) {
    constexpr int c = decltype(cc)::value;
    static_for<0, 3>([&](auto ic) {
      constexpr int e = 3 * c + decltype(ic)::value;
      if constexpr (e < RES2)
        t[e - 3 * c] = rt2[e];
      else
        t[e - 3 * c] = tw2[e * 4 + j];
    });
  };
  v2f* dst = xch + (65 * b + j);
  v2f tnext[3];
  quad_twiddles(IC<0>{}, tnext);
  static_for<0, 4>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const v2f t2[3] = {tnext[0], tnext[1], tnext[2]};
    if constexpr (c < 3) quad_twiddles(IC<c + 1>{}, tnext);
    // 2. transpose (row a) <-> (k1 & 3) inside the register quad
    float x0 = z[4 * c].x, y0 = z[4 * c].y, x1 = z[4 * c + 1].x, y1 = z[4 * c + 1].y;
    float x2 = z[4 * c + 2].x, y2 = z[4 * c + 2].y, x3 = z[4 * c + 3].x, y3 = z[4 * c + 3].y;
    swap_rows32(x0, x2);
    swap_rows32(y0, y2);
    swap_rows32(x1, x3);
    swap_rows32(y1, y3);
    swap_rows16(x0, x1);
    swap_rows16(y0, y1);
    swap_rows16(x2, x3);
    swap_rows16(y2, y3);
    // 3. register 4c+i of lane (j,b) now holds A[k1 = 4c+j][a = i][b]
    v2f v[4] = {mk(x0, y0), cmul(mk(x1, y1), t2[0]), cmul(mk(x2, y2), t2[1]), cmul(mk(x3, y3), t2[2])};
    fft_reg<4, true>(v);
    // 4. B[k1 = 4c+j][k2][b] -> slot 65*b + (4c + j) + 16*k2
    static_for<0, 4>([&](auto kc) {
      constexpr int k2 = decltype(kc)::value;
      dst[4 * c + 16 * k2] = v[k2];
    });
  });
  // step-5 twiddles queue behind the exchange writes so one wait covers both
  v2f t3[15];
  static_for<1, 16>([&](auto bc) {
    constexpr int bb = decltype(bc)::value;
    if constexpr (RES3)
      t3[bb - 1] = rt3[bb - 1];
    else
      t3[bb - 1] = tw3[(bb - 1) * 64 + lane];
  });
  wave_lds_sync();
  FDOCT_FPR(pr, 5);   // (the 1024-point plan does the row swap quad by quad inside steps 3 / 4: phase 4 stays empty)
  const v2f* src = xch + lane;
  static_for<0, 16>([&](auto bc) {
    constexpr int bb = decltype(bc)::value;
    z[bb] = src[65 * bb];
  });
  wave_lds_sync();
  FDOCT_FPR(pr, 6);
  // 5. twiddle and radix-16 over b
  static_for<1, 16>([&](auto bc) {
    constexpr int bb = decltype(bc)::value;
    z[bb] = cmul(z[bb], t3[bb - 1]);
  });
  fft_reg<16, true>(z);
  FDOCT_FPR(pr, 7);
}

// 2048-point version of the same plan ("32 | row swap | 4 | LDS | 16 x2"): n = 64*m + 16*a + b with 32 registers m,
// output k = k1 + 32*k2 + 128*k3.  Steps 1-4 as above with radix 32 and eight register quads (twiddle W_128^(a*k1),
// exchange slot 129*b + l', l' = k1 + 32*k2 < 128); in step 5 every lane takes TWO columns l' = lane + 64*s, s = 0, 1
// (twiddle W_2048^(b*l'), radix-16 over b) and leaves X[lane + 64*(s + 2*k3)] in register s + 2*k3 -- the natural slot
// order.  tw2[(3*c + i-1)*4 + j] = W_128^(i*(4c+j)) (c < 8), tw3[l'] = W_2048^(l'), l' < 128.
// tests/kernel_model.py::fft2048_rowswap_model is the index-for-index numpy model.
__device__ __forceinline__ void fft2048_rowswap(v2f* z, int lane, v2f* xch, const v2f* tw2, const v2f* tw3, FusedProbe& pr) {
  const int j = lane >> 4, b = lane & 15;
  fft_reg<32, true>(z);  // 1.
  FDOCT_FPR(pr, 3);
  static_for<0, 8>([&](auto cc) {  // 2.
    constexpr int c = decltype(cc)::value;
    float x0 = z[4 * c].x, y0 = z[4 * c].y, x1 = z[4 * c + 1].x, y1 = z[4 * c + 1].y;
    float x2 = z[4 * c + 2].x, y2 = z[4 * c + 2].y, x3 = z[4 * c + 3].x, y3 = z[4 * c + 3].y;
    swap_rows32(x0, x2);
    swap_rows32(y0, y2);
    swap_rows32(x1, x3);
    swap_rows32(y1, y3);
    swap_rows16(x0, x1);
    swap_rows16(y0, y1);
    swap_rows16(x2, x3);
    swap_rows16(y2, y3);
    z[4 * c] = mk(x0, y0);
    z[4 * c + 1] = mk(x1, y1);
    z[4 * c + 2] = mk(x2, y2);
    z[4 * c + 3] = mk(x3, y3);
  });
  FDOCT_FPR(pr, 4);
  // 3. + 4.: register 4c+i of lane (j,b) holds A[k1 = 4c+j][a = i][b]
  v2f* dst = xch + (129 * b + j);
  static_for<0, 8>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const v2f* t2 = tw2 + (3 * c) * 4 + j;
    v2f v[4] = {z[4 * c], cmul(z[4 * c + 1], t2[0]), cmul(z[4 * c + 2], t2[4]), cmul(z[4 * c + 3], t2[8])};
    fft_reg<4, true>(v);
    static_for<0, 4>([&](auto kc) {
      constexpr int k2 = decltype(kc)::value;
      dst[4 * c + 32 * k2] = v[k2];
    });
  });
  wave_lds_sync();
  FDOCT_FPR(pr, 5);
  // 5. two columns per lane; the second one is read while the first is being transformed
  const v2f* src = xch + lane;
  v2f u[16], w[16];
  static_for<0, 16>([&](auto bc) {
    constexpr int bb = decltype(bc)::value;
    u[bb] = src[129 * bb];
  });
  static_for<0, 16>([&](auto bc) {
    constexpr int bb = decltype(bc)::value;
    w[bb] = src[129 * bb + 64];
  });
  wave_lds_sync();
  FDOCT_FPR(pr, 6);
  // W_2048^(bb*l'), bb = 1..15, as powers of the one table entry W_2048^(l') (squarings / products, depth <= 6
  // multiplies): 15 KB less LDS per workgroup than a full table, which is worth a wave per CU here
  auto column = [&](v2f* col, v2f w1) {
    v2f pw[16];
    pw[1] = w1;
    static_for<2, 16>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      pw[r] = (r & 1) ? cmul(pw[r - 1], pw[1]) : cmul(pw[r / 2], pw[r / 2]);
    });
    static_for<1, 16>([&](auto bc) {
      constexpr int bb = decltype(bc)::value;
      col[bb] = cmul(col[bb], pw[bb]);
    });
    fft_reg<16, true>(col);
  };
  column(u, tw3[lane]);
  column(w, tw3[lane + 64]);
  static_for<0, 16>([&](auto kc) {
    constexpr int k3 = decltype(kc)::value;
    z[2 * k3] = u[k3];
    z[2 * k3 + 1] = w[k3];
  });
  FDOCT_FPR(pr, 7);
}

// ------------------------------------------------------------ input types --
// One 8-sample chunk s0..s7 of a row as loaded (prefetched) from HBM.  unpack() gives the four pairs
// (s0,s2) (s4,s6) (s1,s3) (s5,s7): evens and odds apart, so that the slope step's "previous sample" of an odd
// pair IS the even pair, and the de-interleaved staging stores are whole registers quads (no shuffles).
// chunk_pair_offset(q) = index of the first sample of pair q; the second one is two samples later.
__device__ __forceinline__ constexpr int chunk_pair_offset(int q) { return (q & 1) * 4 + (q >> 1); }
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
template <typename IN_T>
struct RawChunk;
template <>
struct RawChunk<uint16_t> {
  uint4 v;
  __device__ __forceinline__ void load(const void* row, int i0) {
    // streaming load: camera samples are read once (nt keeps them from displacing the twiddle / plane lines in L2; +1.2 %,
    // DESIGN.md 5 -- the sc0/sc1 scope bits change nothing)
    const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(static_cast<const uint16_t*>(row) + i0));
    v = make_uint4(t.x, t.y, t.z, t.w);
  }
  __device__ __forceinline__ void zero() { v = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void pin() { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
  __device__ __forceinline__ void pin_ordered() { asm volatile("" : "+v"(v.x) : : "memory"); }
  __device__ __forceinline__ void unpack(v2f* x) const {
    x[0] = mk((float)(v.x & 0xffffu), (float)(v.y & 0xffffu));
    x[1] = mk((float)(v.z & 0xffffu), (float)(v.w & 0xffffu));
    x[2] = mk((float)(v.x >> 16), (float)(v.y >> 16));
    x[3] = mk((float)(v.z >> 16), (float)(v.w >> 16));
  }
};
template <>
struct RawChunk<uint8_t> {
  uint2 v;
  __device__ __forceinline__ void load(const void* row, int i0) {
    const u2v t = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(static_cast<const uint8_t*>(row) + i0));
    v = make_uint2(t.x, t.y);
  }
  __device__ __forceinline__ void zero() { v = make_uint2(0, 0); }
  __device__ __forceinline__ void pin() { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
  __device__ __forceinline__ void pin_ordered() { asm volatile("" : "+v"(v.x) : : "memory"); }
  __device__ __forceinline__ void unpack(v2f* x) const {
    x[0] = mk((float)(v.x & 0xffu), (float)((v.x >> 16) & 0xffu));
    x[1] = mk((float)(v.y & 0xffu), (float)((v.y >> 16) & 0xffu));
    x[2] = mk((float)((v.x >> 8) & 0xffu), (float)(v.x >> 24));
    x[3] = mk((float)((v.y >> 8) & 0xffu), (float)(v.y >> 24));
  }
};
template <>
struct RawChunk<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const void* row, int i0) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v* p = reinterpret_cast<const f4v*>(static_cast<const float*>(row) + i0);
    const f4v ta = __builtin_nontemporal_load(p), tb = __builtin_nontemporal_load(p + 1);
    a = make_float4(ta.x, ta.y, ta.z, ta.w);
    b = make_float4(tb.x, tb.y, tb.z, tb.w);
  }
  __device__ __forceinline__ void zero() { a = b = make_float4(0, 0, 0, 0); }
  __device__ __forceinline__ void pin() {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
  }
  __device__ __forceinline__ void pin_ordered() { asm volatile("" : "+v"(a.x) : : "memory"); }
  __device__ __forceinline__ void unpack(v2f* x) const {
    x[0] = mk(a.x, a.z);
    x[1] = mk(b.x, b.z);
    x[2] = mk(a.y, a.w);
    x[3] = mk(b.y, b.w);
  }
};

template <int T>
__device__ __forceinline__ float group_min(float v) {
#pragma unroll
  for (int m = T / 2; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
  return v;
}
template <int T>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int m = T / 2; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// One DPP step of a wave-wide f64 sum: v + (v moved by `CTRL`).  Only lane 63's total is used, so lanes that
// the row mask leaves unwritten may hold anything; row shifts read 0 past the row edge (bound_ctrl).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int tlo = __builtin_amdgcn_mov_dpp(lo, CTRL, ROW_MASK, 0xf, true);
  const int thi = __builtin_amdgcn_mov_dpp(hi, CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(thi, tlo);
}

// Sum over the T lanes of a row group, returned to every lane of the group.  T == 64: DPP
// row shifts / broadcasts (no LDS, no address registers) and a scalar broadcast of lane 63.
template <int T>
__device__ __forceinline__ double group_sum(double v) {
  if constexpr (T == 64) {
    v = dpp_add_f64<0x111, 0xf>(v);  // row_shr:1  -> lane i: v[i-1..i]
    v = dpp_add_f64<0x112, 0xf>(v);  // row_shr:2  -> v[i-3..i]
    v = dpp_add_f64<0x114, 0xf>(v);  // row_shr:4  -> v[i-7..i]
    v = dpp_add_f64<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each row: the row's total
    v = dpp_add_f64<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3 add the previous row's total
    v = dpp_add_f64<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3 add lane 31
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
  } else {
#pragma unroll
    for (int m = T / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
  }
}

// One DPP step of a wave-wide f32 sum (lanes the row mask leaves out add 0).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f32(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}

// Sum over the T lanes of a row group in f32, returned to every lane of the group (T == 64: the DPP chain of
// group_sum, one v_add_f32_dpp per step).
template <int T>
__device__ __forceinline__ float group_sum_f32(float v) {
  if constexpr (T == 64) {
    v = dpp_add_f32<0x111, 0xf>(v);
    v = dpp_add_f32<0x112, 0xf>(v);
    v = dpp_add_f32<0x114, 0xf>(v);
    v = dpp_add_f32<0x118, 0xf>(v);
    v = dpp_add_f32<0x142, 0xa>(v);
    v = dpp_add_f32<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  } else {
#pragma unroll
    for (int m = T / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
  }
}

// Row mean (main:1138) of a fast-path row from the per-lane f32 sums p (every lane holds the same number of samples),
// as an unevaluated two-float sum mh + ml, without f64 arithmetic.  The lane sums are all close to SPL * mean, and what
// distinguishes them (the fringes) is small next to that, so a plain f32 reduction would round at the size of the
// total.  Instead: a first, sloppy reduction gives the average lane sum `base`; the second reduction runs on
// q = p - base (values and partial sums of fringe size), whose rounding errors are those of any f32 arithmetic on
// the fringe signal itself.  mean = base / SPL + sum(q) / W; inv_spl and inv_w are exact reciprocals (powers of two
// on the compiled plans).
template <int T>
__device__ __forceinline__ void group_mean_f32(float p, float inv_t, float inv_spl, float inv_w, float& mh, float& ml) {
  const float base = group_sum_f32<T>(p) * inv_t;
  const float qs = group_sum_f32<T>(p - base);
  const float m1 = base * inv_spl, m2 = qs * inv_w;
  mh = m1 + m2;  // two-sum: mh + ml == m1 + m2 exactly
  const float bb = mh - m1;
  ml = (m1 - (mh - bb)) + (m2 - bb);
}

// fma with a half-float second operand (the low / high half of hp), converted inside the instruction: a * f16(hp) + c.
__device__ __forceinline__ float fma_mix_lo(float a, uint32_t hp, float c) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(hp), "v"(c));
  return r;
}
__device__ __forceinline__ float fma_mix_hi(float a, uint32_t hp, float c) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(hp), "v"(c));
  return r;
}

// Reads the 8 constants of chunk c of one constant plane pair in the RawChunk pair order.
// Plane layout (see the staging loop in the kernel): chunk c, parity h, lane ln -> c*8T + h*4T + 4*ln.
template <int T>
__device__ __forceinline__ void load_consts(const float* plane_lane, int c, v2f* out) {
  const float4 q0 = *reinterpret_cast<const float4*>(plane_lane + 8 * T * c);
  const float4 q1 = *reinterpret_cast<const float4*>(plane_lane + 8 * T * c + 4 * T);
  out[0] = mk(q0.x, q0.y);
  out[1] = mk(q0.z, q0.w);
  out[2] = mk(q1.x, q1.y);
  out[3] = mk(q1.z, q1.w);
}

// ------------------------------------------------------------ fused kernel --
// LOG2NC: log2 of the complex FFT length NC (= N/2 real path, N complex path)
// T: lanes per row (64/T rows per wave); R1*R2*R3 = NC (R3 = 1: two passes); KIND: 0 = Stockham
// passes through LDS, 1 = fft1024_rowswap; WCH: 8-sample chunks per lane (W <= WC = 8*T*WCH);
// CPLX: dispersion phase path.
// LEAN: the benchmark / common acquisition configuration, compiled without any
//   predication: W == WC, 1-row background, no pi/dark frame, no normalisation (any
//   number of averaged frames).  !LEAN handles everything else.
// STAGE: 0 = the whole chain in one launch (default).  1 = "resample stage" only: samples in, the
//   k-linear row (data_ylin packed as the FFT input, NC float2 per row) out to a.ylin.  2 = "FFT stage"
//   only: a.ylin in, magnitudes/dB out.  Stages 1+2 reproduce stage 0 bit for bit at 3x the HBM traffic;
//   they exist so that each stage can be timed against the HBM roofline on its own (north star) and as
//   the seam for stages that need the intermediate in memory.
// The window table arrives pre-multiplied by 1/2 on the real path (host side): the untangle
// needs X = (A + w*O)/2 and a power-of-two scale of the window commutes exactly with every step.
// IB2D (fast path of the row-swap plan only): the background is a full H x W frame, as the reference's 'b' key
//   stores it.  The resident 1/background registers then double as a one-row-ahead prefetch buffer of the frame row
//   the next A-scan needs (a.ib2d: each 8-sample group stored evens first, then odds -- the RawChunk pair order).
// NORM (same kernels): 1 = whole-frame min-max normalisation (main:1128-1129; always on in BscanFFTsim.cpp:845) from the
//   per-frame (min,max) of the pre-pass, the scale of the next A-scan's frame being fetched at the prefetch point;
//   2 = row-wise min-max normalisation (normalizerows, main:88-97, 1126) with a wave-wide min/max of the row.
// TRO (fast path of the row-swap 1024-point plan, one row per wave): the outputs are written in the reference's own
//   layout, bscan[depth][row] (main:1220), by the chain itself, through a ring of finished rows in LDS.  A workgroup owns
//   TR = FUSED_TR_ROWS consecutive A-scans of one B-scan at a time (a "tile").  Its waves claim the tile's rows from the
//   ticket counter; a finished row goes into slot (ticket mod fused_tro_ring_slots(D): 20, or 40 up to 512 depth bins) of the ring (row-major, 4 B per lane:
//   conflict free) and is counted in an LDS counter of its tile.  A complete tile is written out in steps of 64 depth bins:
//   a step reads 16-byte pieces of 4 rows x 4 bins per lane -- a 4 x 4 block whose transposition is a renaming of
//   registers -- and stores, per depth bin, 16 bytes per lane = TR * 4 contiguous bytes of the B-scan.  When both bscan and
//   bscandb are asked for, the ring holds bscan and the step takes the logarithm (the same instruction on the same value as
//   the epilogue would).  The steps are claimed one at a time (compare-and-swap on an LDS counter) by whichever wave passes
//   a hand-over point: after putting a row into the ring, while waiting for a ring slot, and -- its rows done -- until the
//   workgroup's last tile is out (FDOCT_TRO_DW = 0 keeps the first form instead: the last wave of the workgroup computes
//   nothing and writes every tile out; 4-8 % slower).  No workgroup barrier, and nothing of it crosses HBM or L2: per A-scan
//   only the camera samples are read and the images written.  The ring is a quarter larger than a tile, so the waves run on
//   into the next tile while one is written out; a wave waits only when the slot it needs still holds a row of a tile that
//   has not been written out, and takes write-out steps itself while it waits.  Nobody waits in a cycle: a row is counted
//   as soon as it is in the ring, a wave waits for write-outs of EARLIER tiles only, a step is claimed only when its tile
//   is complete, and every waiting wave works on the steps it waits for.
//   (Round 3 first built this with the tiles in global memory, 128 KB per workgroup and buffer: they did not stay in the
//   4 MB of L2 an XCD's 32 workgroups share, and the chain ran at the rate of the two-pass path; profiles/r03_tro_probe*.)
template <int LOG2NC, int T, int R1, int R2, int R3, int KIND, int WCH, typename IN_T, bool CPLX, bool LEAN, int STAGE, bool AVG,
          bool IB2D = false, int NORM = 0, bool TRO = false, bool PRECT = false>
__global__ __launch_bounds__(fused_max_block(1 << LOG2NC, T, LEAN, KIND)) void fused_kernel(const FusedArgs a) {
  constexpr int NC = 1 << LOG2NC;
  constexpr int P = NC / T;
  constexpr int RPW = 64 / T;  // rows per wave
  constexpr int WC = 8 * T * WCH;
  constexpr int NPR = 4 * WCH;  // sample pairs per lane
  constexpr int LP = (R1 == 32) ? 5 : (R1 == 16) ? 4 : (R1 == 8) ? 3 : 2;
  static_assert(R1 * R2 * R3 == NC, "radix plan");
  static_assert(KIND == 0 || (KIND == 1 && T == 64 && R1 == 16 && R2 == 4 && R3 == 16) ||
                    (KIND == 2 && T == 64 && R1 == 32 && R2 == 4 && R3 == 16),
                "row-swap plans are 16 x 4 x 16 / 32 x 4 x 16 on a full wave");
  constexpr int NPASS = (R3 > 1) ? 3 : 2;
  // 1/background as two floats (fdoct_capi.cpp::reciprocal_words): always on the any-option kernel; the fast path has both
  // instantiations (PRECT: fdoct_set_precise_division)
  constexpr bool PREC = !LEAN || PRECT;
  // the second word's form on the fast-path kernels with at most 32 samples per lane: a plane of half floats (fdoct_kernels.h)
  constexpr bool IL16 = PRECT && fused_il_half(LEAN, WCH);
  static_assert(!(PRECT && !LEAN), "the any-option kernel always multiplies by both words");
  static_assert(!(PRECT && IB2D && !IL16), "a full-frame background on the fast path: the second word is prefetched as half floats");

  __shared__ unsigned int row_ticket;  // next unclaimed row slot of this workgroup
  __shared__ unsigned int tr_arrived[4];  // TRO: rows of tile (q mod 4) in the ring
  // TRO: tiles q = i (mod 4) that have completed so far.  Tiles need not complete in order -- with a 4-row last tile of a
  // B-scan, or the 40-slot ring, no row of tile q + 1 waits for anything of tile q -- but tiles four apart do (a row of tile
  // q + 4 needs tile q + 1 written out, which the in-order counters below allow only after tile q), so these only grow and
  // "tile q is complete" is tr_done[q & 3] > q / 4.  tro_publish() turns them into the in-order counter everything else reads.
  __shared__ unsigned int tr_done[4];
#if FDOCT_TRO_DW != 1
  __shared__ unsigned int tr_released;    // TRO, write-out by one wave per tile: tiles 0 .. tr_released - 1 are written out
#endif
#if FDOCT_TRO_DW == 1
  __shared__ unsigned int tr_ready, tr_wo_next, tr_wo_done;  // TRO, write-out by all waves: complete tiles; next step to claim; steps done (cumulative)
#endif
  // TRO with FOUR rows per wave (the 512-point plan, round 6): no ring.  A GROUP of four waves (GW, fdoct_kernels.h) owns a tile of 16 rows; a finished
  // row is deposited in the wave's OWN row buffer (free between the untangle and the next row's staging), the group meets, writes
  // the tile out together, meets again, and goes on -- all eight waves compute (the ring cost two of them their LDS).
  __shared__ unsigned int grp_tile[4][2];   // the tile a group's leader has claimed, by sequence parity
  __shared__ unsigned int grp_pub[4];       // tiles published to a group so far
  __shared__ unsigned int grp_arrived[4];   // waves of a group whose rows are deposited (cumulative: 4 per tile)
  __shared__ unsigned int grp_done[4];      // waves of a group that have finished their share of a write-out (cumulative)
  static_assert(!TRO || (LEAN && STAGE == 0 && !CPLX && fused_tro_compiled(KIND, T, WCH) && (FUSED_TR_ROWS % (64 / T)) == 0),
                "fused transposed store: fast path; a wave's rows lie in one tile");
  extern __shared__ __align__(16) unsigned char smem[];
  const int cw = a.lds_planes ? WC : 0;          // resident-constant kernels: the host leaves the planes out
  float* c_ib = reinterpret_cast<float*>(smem);  // [WC] 1/background
  float* c_win = c_ib + cw;                      // [WC] window
  float* c_g = c_win + cw;                       // [WC] fractionalk by sample index
  float* c_il = c_g + cw;                        // [WC] low word of 1/background (a.prec == 1: every kernel reads it from here, the resident-constant ones too)
  const int cwl = (a.prec == 1 && STAGE != 2) ? (IL16 ? WC / 2 : WC) : 0;  // floats (the FFT-stage kernel reads no samples); IL16: 2 WC bytes
  v2f* c_tw = reinterpret_cast<v2f*>(c_il + cwl);  // twiddle tables, a.tw_count entries
  // (transposed-store kernels leave out of LDS what they read once into registers, so that the ring of finished rows can be
  // larger: fdoct_kernels.h, fused_tw3_in_lds / fused_gi_in_lds)
  // (PF2: the transposed store's samples prefetched two rows ahead -- fdoct_kernels.h, fused_tro_pf2, and the row loop below)
  constexpr bool PF2 = fused_tro_pf2(KIND, LEAN, STAGE, CPLX, AVG, TRO, IB2D, NORM, (int)sizeof(IN_T));
  constexpr bool TW3_LDS = fused_tw3_in_lds(KIND, LEAN, STAGE, TRO, IB2D && IL16), GI_LDS = fused_gi_in_lds(KIND, LEAN, STAGE, CPLX, AVG, TRO, PF2);
  const int tw_lds = TW3_LDS ? a.tw_count : (R2 - 1) * R1;   // entries staged: all, or the step-3 table only
  v2f* c_ph = c_tw + tw_lds;                     // [NC] phase (CPLX only)
  uint32_t* c_gi = reinterpret_cast<uint32_t*>(c_ph + (CPLX ? NC : 0));  // [NC] packed gather offsets
  unsigned char* scratch0 = reinterpret_cast<unsigned char*>(c_gi + (GI_LDS ? NC : 0));

#ifdef FDOCT_CLOCKPROBE
  const unsigned long long probe_c0 = __builtin_readcyclecounter(), probe_r0 = wall_clock64();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = lane & (T - 1);
  const int sub = lane / T;

  // ---- stage the per-column constants once per workgroup
  // layout: sample i = 8*(lane + T*c) + e  ->  slot c*8T + (e&1)*4T + 4*lane + (e>>1), i.e. each
  // chunk is split into an even and an odd plane of 4 floats per lane (the RawChunk pair order), so a
  // wave's b128 reads are contiguous
  for (int i = tid; i < cw; i += blockDim.x) {
    const bool in = i < a.W;
    const int e = i & 7, ln = (i >> 3) & (T - 1), c = i / (8 * T);
    const int slot = c * 8 * T + (e & 1) * 4 * T + 4 * ln + (e >> 1);
    c_ib[slot] = (in && a.ib) ? a.ib[i] : 0.f;
    c_win[slot] = in ? a.win[i] : 0.f;
    c_g[slot] = in ? a.g[i] : 0.f;
  }
  if constexpr (IL16) {  // half-float pairs, already in the order the lanes read them (fdoct_capi.cpp)
    for (int i = tid; i < cwl; i += blockDim.x) reinterpret_cast<uint32_t*>(c_il)[i] = a.il16 ? a.il16[i] : 0u;
  } else {
    for (int i = tid; i < cwl; i += blockDim.x) {  // (the same slot rule)
      const int e = i & 7, ln = (i >> 3) & (T - 1), c = i / (8 * T);
      c_il[c * 8 * T + (e & 1) * 4 * T + 4 * ln + (e >> 1)] = (i < a.W && a.il) ? a.il[i] : 0.f;
    }
  }
  {
    const v2f* gtw = reinterpret_cast<const v2f*>(a.tw);
    for (int i = tid; i < tw_lds; i += blockDim.x) c_tw[i] = gtw[i];
    if constexpr (CPLX) {
      const v2f* gph = reinterpret_cast<const v2f*>(a.phase);
      for (int i = tid; i < NC; i += blockDim.x) c_ph[i] = gph[i];
    }
  }
  constexpr bool TRO_INPLACE = TRO && RPW == 4;
  constexpr unsigned GW = (unsigned)fused_tro_group_waves();   // waves of a group (TRO_INPLACE): a tile is 4 GW rows
  if (tid == 0) row_ticket = TRO_INPLACE ? (blockDim.x >> 6) / GW : (blockDim.x >> 6) - ((TRO && !FDOCT_TRO_DW) ? 1u : 0u);  // slots 0 .. nwaves-1 are the (computing) waves' first rows (groups' first tiles)
  if (TRO && tid < 4) tr_arrived[tid] = tr_done[tid] = 0u;
  if (TRO_INPLACE && tid < 4) grp_pub[tid] = grp_arrived[tid] = grp_done[tid] = 0u;
#if FDOCT_TRO_DW == 1
  if (TRO && tid == 0) tr_ready = tr_wo_next = tr_wo_done = 0u;
#else
  if (TRO && tid == 0) tr_released = 0u;
#endif
  // gather table: entry n = ln + T*m is stored at [(m/4)][ln][m%4] so a lane's P entries are P/4
  // b128 reads with a 16-byte lane stride (re-read every row: cheaper than P resident VGPRs)
  if constexpr (GI_LDS) {
    for (int i = tid; i < NC; i += blockDim.x) {
      const int ln = i & (T - 1), m = i / T;
      c_gi[(m >> 2) * 4 * T + 4 * ln + (m & 3)] = a.gidx[i];
    }
  }
  __syncthreads();

  // ---- TRO: tiles.  Tile q of workgroup b is tile q * grid + b of the batch (front to back); tiles never straddle B-scans
  // (the last tile of a B-scan may be short).  Wave-uniform, scalar unit.
  constexpr unsigned TR = TRO_INPLACE ? 4u * GW : (unsigned)FUSED_TR_ROWS;
  auto tro_tile = [&](unsigned tq, unsigned& g, unsigned& r0, unsigned& nrows) -> bool {  // false: past the end of the batch
#ifndef FDOCT_TRO_NO_XCDPAIR
    // workgroups b, b + 8, b + 16 .. run on the same XCD (round-robin dispatch) at about the same time: they get a run of
    // NEIGHBOURING tiles, so that the two 64-byte halves of every 128-byte line of the B-scan meet in one L2 (a
    // performance matter only: any bijection of the workgroups is correct)
    const unsigned nx = gridDim.x >> 3;
    const unsigned bperm = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * nx + (blockIdx.x >> 3);
    const unsigned gt = tq * gridDim.x + bperm;
#else
    const unsigned gt = tq * gridDim.x + blockIdx.x;
#endif
    if (gt >= a.tr_total_tiles) return false;
    g = __umulhi(gt, a.tr_tpf_magic);  // gt / tiles-per-frame: floor(2^32 / tpf) under-estimates by at most one
    unsigned tf = gt - g * a.tr_tpf;
    if (tf >= a.tr_tpf) {
      g++;
      tf -= a.tr_tpf;
    }
    r0 = tf * TR;
    const unsigned left = (unsigned)a.H - r0;
    nrows = left < TR ? left : TR;
    return true;
  };
  // ring slots: one of a few compile-time values (kTroRingChoices), so that "mod RS" stays a multiplication; the launch picks
  // the largest that fits the LDS (wave-uniform: a scalar branch)
  const unsigned RS = (unsigned)__builtin_amdgcn_readfirstlane((int)a.tr_ring);
  auto ring_mod = [&](unsigned x) -> unsigned {
    switch (RS) {
      case 21: return x % 21u;
      case 22: return x % 22u;
      case 23: return x % 23u;
      case 24: return x % 24u;
      case 26: return x % 26u;
      case 28: return x % 28u;
      case 32: return x % 32u;
      case 40: return x % 40u;
      case 44: return x % 44u;
      case 48: return x % 48u;
      default: return x % 20u;
    }
  };
  constexpr int TRO_WRITERS = FDOCT_TRO_DW ? 0 : 1;  // waves of the workgroup that only write out
  // the ring lies behind the computing waves' row buffers; a slot is D + 4 floats (the pad moves consecutive rows 4 banks apart)
  float* const tro_ring = reinterpret_cast<float*>(scratch0 + (size_t)((blockDim.x >> 6) - TRO_WRITERS) * RPW * a.scratch_bytes);
  const int tro_slot = a.D + 4;
  // One write-out step: bins s0 .. s0 + SB - 1 of tile tq, all its rows.  Lane (dg, rq) takes rows 4 rq .. 4 rq + 3 and bins
  // 4 dg .. 4 dg + 3: four ds_read_b128 (one per row), four 16-byte stores (one per bin: the lanes of a row-quad group cover
  // TR * 4 contiguous bytes of the B-scan).  When both images are asked for the ring holds bscan and the logarithm is taken
  // here (the same instruction on the same value as the epilogue's).
  constexpr int TRO_RQ = TR / 4, TRO_DGN = 64 / TRO_RQ, TRO_SB = 4 * TRO_DGN;
  // PF2: store INSTRUCTIONS this wave has issued since its last wait for samples (wave-uniform; an instruction counts whatever
  // its lanes' masks): the wait names how many younger operations may still be outstanding
  [[maybe_unused]] unsigned pf2_stores = 0u;
  auto tro_step = [&](unsigned tq, unsigned g, unsigned r0, unsigned nrows, int s0) {
    if constexpr (PF2) pf2_stores += (a.out_mag && a.out_db) ? 8u : 4u;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int rq = lane % TRO_RQ, dg = lane / TRO_RQ;
    const int Dn = a.D, Hn = a.H;
#if FDOCT_TRO_X == 2   // measurement builds (tools/tro_cost_probe.sh): 2 = no write-out work at all, 1 = the LDS reads without the stores
    return;
#endif
    if (4 * rq >= (int)nrows) return;
    const bool both = a.out_mag && a.out_db;
    const bool mask = both && a.dcmask && Dn > 4;  // dB bins 0, 1 <- bin 4 (main:1237-1238); alone, dB arrives masked
    float* const out0 = a.out_mag ? a.out_mag : a.out_db;
#if FDOCT_TRO_X == 3   // measurement build: every tile's stores land in the same 64 KB (16 "rows" per depth bin): same instruction
    const int Hs = 16;  // stream, L2-resident targets -- do the waves wait for the stores' COMPLETION (EXPERIMENTS.md section 5)?  Results are wrong.
#else
    const int Hs = Hn;
#endif
    const int vout = (4 * dg * Hs + 4 * rq) * 4;
    const float* rowp[4];
    if constexpr (TRO_INPLACE) {
      // rows 4 rq .. 4 rq + 3 of the tile lie in the row buffers of wave rq of the group (tq names the GROUP here): row i of the
      // tile is row buffer 16 group + i, buffers scratch_bytes apart (1060 floats: four banks on, like the ring's slots)
#pragma unroll
      for (int i = 0; i < 4; i++)
        rowp[i] = reinterpret_cast<const float*>(scratch0 + (size_t)(TR * tq + 4u * (unsigned)rq + (unsigned)i) * a.scratch_bytes) + 4 * dg + s0;
    } else {
      const unsigned sl0 = ring_mod(TR * tq + 4u * (unsigned)rq);  // this lane's four rows: ring slots (TR tq + 4 rq + i) mod RS
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned sl = sl0 + i >= RS ? sl0 + i - RS : sl0 + i;
        rowp[i] = tro_ring + sl * tro_slot + 4 * dg + s0;
      }
    }
#if FDOCT_TRO_X == 3
    const size_t goff = 0;
#else
    const size_t goff = ((size_t)g * Dn) * Hn + r0;
#endif
    __amdgpu_buffer_rsrc_t rout0 = __builtin_amdgcn_make_buffer_rsrc(out0 + goff, 0, 0x7ffffff0, 0x00020000);
    f4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = *reinterpret_cast<const f4*>(rowp[i]);
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
      const f4 w = {v[0][bb], v[1][bb], v[2][bb], v[3][bb]};
#if FDOCT_TRO_X == 1
      asm volatile("" ::"v"(w));
      if (Dn < 0)
#endif
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, w), rout0, vout, ((s0 + bb) * Hs) * 4, FDOCT_TRO_OUT_AUX);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop %0" ::"n"(FDOCT_TRO_NOP - 1));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (both) {
      __amdgpu_buffer_rsrc_t rout1 = __builtin_amdgcn_make_buffer_rsrc(a.out_db + goff, 0, 0x7ffffff0, 0x00020000);
      f4 v4 = {0.f, 0.f, 0.f, 0.f};  // bin 4 of the four rows (DC mask)
      if (mask && s0 == 0 && dg == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) v4[i] = rowp[i][4];
      }
#pragma unroll
      for (int bb = 0; bb < 4; bb++) {
        f4 w = {v[0][bb], v[1][bb], v[2][bb], v[3][bb]};
        if (mask && bb < 2 && s0 == 0 && dg == 0) w = v4;
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = a.db_scale * fast_log2(w[k]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, w), rout1, vout, ((s0 + bb) * Hs) * 4, FDOCT_TRO_OUT_AUX);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop %0" ::"n"(FDOCT_TRO_NOP - 1));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
#if FDOCT_TRO_DW != 0
  // Tile tq of this workgroup has just completed (FDOCT_TRO_DW = 1: all its rows are in the ring; 2: it is written out).
  // Tiles need not complete in order (see tr_done), but everything that waits or claims counts tiles IN ORDER: the tile is
  // counted in tr_done, then the in-order counter *inorder is advanced over every tile that is complete by now -- by this
  // wave or by whichever wave's compare-and-swap wins; a wave that completes tile q + 1 before tile q leaves the counter
  // alone and the wave that completes tile q later takes it past both.  (No lost hand-over: a wave reads the counters AFTER
  // its own tile is counted, and LDS operations execute one at a time, a wave's own in program order.)  Returns the counter
  // as last read.  Three LDS round trips per tile, i.e. per FUSED_TR_ROWS rows.
  auto tro_publish = [&](unsigned tq, unsigned int* inorder) -> unsigned {
    if (lane == 0) {
      // (next user of the row counter: tile tq + 4, whose rows wait for tile tq + 1 to be written out -- after this)
      __hip_atomic_store(&tr_arrived[tq & 3u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&tr_done[tq & 3u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned r;
    for (;;) {
      const unsigned r_l = __hip_atomic_load(inorder, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned d_l = __hip_atomic_load(&tr_done[lane & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      r = (unsigned)__builtin_amdgcn_readfirstlane((int)r_l);
      const unsigned dc = (unsigned)__builtin_amdgcn_readlane((int)d_l, (int)(r & 3u));
      if (dc <= (r >> 2)) break;  // tile r is not complete yet: whoever completes it goes on from here
      if (lane == 0) {
        unsigned expect = r;
        (void)__hip_atomic_compare_exchange_strong(inorder, &expect, r + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    return r;
  };
#endif
#if FDOCT_TRO_DW == 2
  // Write-out by the wave whose row completes the tile: all SPT steps in one go, the LDS reads of a step issued ahead of the
  // stores of the step before it.  tr_released counts the tiles written out, in order (tro_publish).
  auto tro_tile_out = [&](unsigned tq, unsigned g, unsigned r0, unsigned nrows) {
    for (int s0 = 0; s0 < a.D; s0 += TRO_SB) tro_step(tq, g, r0, nrows, s0);
    asm volatile("" ::: "memory");  // every LDS read has returned (its data fed a store that has been issued)
    (void)tro_publish(tq, &tr_released);
  };
  unsigned tro_rel_seen = 0u;  // tiles written out, as last read (the counter only grows: a valid lower bound)
#elif FDOCT_TRO_DW == 1
  // Distributed write-out: no wave is set aside.  The wave whose row completes a tile publishes it (tr_ready counts the
  // complete tiles IN ORDER, tro_publish: tile q's steps exist once tiles 0 .. q are all complete); its SPT = D / SB steps are
  // then claimed one at a time (compare-and-swap on tr_wo_next, so a step is never claimed before it exists) by whichever
  // wave passes a hand-over point: after putting a row into the ring, while waiting for a ring slot, and -- all rows
  // done -- until the workgroup's last tile is out.  The wave that finishes a tile's last step releases its slots.
  // (LDS round trips are what this costs -- a returning LDS operation takes 100-200 cycles next to the chain's own LDS
  // traffic -- so they are batched: tr_wo_next and tr_ready are read together, the steps-done counter is cumulative and bumped
  // without a return value (tile q is out when it reaches (q + 1) SPT), and only the claim itself is a round trip of its own.)
  auto tro_try_step = [&](unsigned s, unsigned ready) -> bool {  // s = tr_wo_next as just read, ready = complete tiles
    const unsigned spt = (unsigned)a.D / (unsigned)TRO_SB;
    if (s >= ready * spt) return false;
    unsigned got = 0u;
    if (lane == 0) {
      unsigned expect = s;
      got = __hip_atomic_compare_exchange_strong(&tr_wo_next, &expect, s + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
    }
    if (!__builtin_amdgcn_readfirstlane((int)got)) return true;  // another wave took it: something is going on
    const unsigned tq = s / spt, k = s - tq * spt;
    unsigned g, r0, nrows;
    (void)tro_tile(tq, g, r0, nrows);
    asm volatile("" ::: "memory");
    tro_step(tq, g, r0, nrows, (int)k * TRO_SB);
    asm volatile("" ::: "memory");  // the step's LDS reads have returned (they fed stores that have been issued)
    if (lane == 0) __hip_atomic_fetch_add(&tr_wo_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return true;
  };
  auto tro_writeout = [&](int max_steps) -> bool {  // true: a step was there (taken by this wave or another)
    bool did = false;
    for (int it = 0; it < max_steps; it++) {
      const unsigned s_l = __hip_atomic_load(&tr_wo_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned r_l = __hip_atomic_load(&tr_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (!tro_try_step((unsigned)__builtin_amdgcn_readfirstlane((int)s_l), (unsigned)__builtin_amdgcn_readfirstlane((int)r_l))) break;
      did = true;
    }
    return did;
  };
  unsigned tro_done_seen = 0u;  // steps written out, as last read (the counter only grows: a valid lower bound)
#else  // FDOCT_TRO_DW == 0
  if constexpr (TRO) {
    // The LAST wave of the workgroup is the write-out wave: it computes nothing.
    if (wave == (int)(blockDim.x >> 6) - 1) {
      for (unsigned tq = 0;; tq++) {
        unsigned g, r0, nrows;
        if (!tro_tile(tq, g, r0, nrows)) break;
        // every row of the tile in the ring?  (The rows arrive whatever this wave does; the bound is the exit condition a
        // spinning wave must have all the same -- about half a second -- and is reported through a.tr_fault.)
        for (unsigned spin = 0;; spin++) {
          const unsigned have = (unsigned)__builtin_amdgcn_readfirstlane(
              (int)__hip_atomic_load(&tr_arrived[tq & 3u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          if (have >= nrows) break;
          if (spin >= FDOCT_TRO_SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_store(a.tr_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        for (int s0 = 0; s0 < a.D; s0 += TRO_SB) tro_step(tq, g, r0, nrows, s0);
        // every LDS read above has returned (its data fed a store that has been issued): the slots may be overwritten
        asm volatile("" ::: "memory");
        if (lane == 0) {
          __hip_atomic_store(&tr_arrived[tq & 3u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(&tr_released, tq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      return;
    }
  }
#endif

  unsigned char* scr = scratch0 + (size_t)(wave * RPW + sub) * a.scratch_bytes;
  float* stg = reinterpret_cast<float*>(scr);
  v2f* xch = reinterpret_cast<v2f*>(scr);

  // ---- per-lane constants kept in registers for every row
  v2f utw = mk(1.f, 0.f);
  if constexpr (!CPLX) utw = reinterpret_cast<const v2f*>(a.utw)[l];  // exp(+2*pi*i*l/N)

  const v2f* tw_p2 = c_tw;                                       // pass 2 table: (R2-1) x R1
  // pass 3 table: (R3-1) x (R1*R2) (row-swap plans: the step-5 table); kernels that hold its entries in registers and need
  // the LDS for something else (TW3_LDS false) read them once from the global table
  const v2f* tw_p3 = TW3_LDS ? c_tw + (R2 - 1) * R1 : reinterpret_cast<const v2f*>(a.tw) + (R2 - 1) * R1;

  // Fast-path row-swap plan (the benchmark configuration): everything that does not depend on the row --
  // the per-column constants, the gather addresses and the FFT twiddles -- stays in registers (2 waves per
  // SIMD, 256 VGPRs), which removes half of the LDS traffic per row.  Every other instantiation re-reads
  // them from LDS each row.
  constexpr bool RES = fused_resident_consts(KIND, LEAN, AVG, WCH, STAGE);
#ifdef FDOCT_X_NO_GRES
  constexpr bool GRES = false;
#else
  constexpr bool GRES = LEAN && STAGE != 2 && KIND == 1 && !CPLX && !AVG && !PF2;  // (with averaging the accumulators need those registers; PF2: the second sample set does)
#endif
  uint32_t gaddr[GRES ? 2 * P : 1];
  if constexpr (GRES) {  // 2 LDS byte addresses per FFT point
    const uint4* gl4 = reinterpret_cast<const uint4*>(c_gi) + l;
    const uint32_t sbase = (uint32_t)(uintptr_t)scr;  // LDS byte address of this wave's staging buffer
#pragma unroll
    for (int q = 0; q < P / 4; q++) {
      uint4 g4;
      if constexpr (GI_LDS)
        g4 = gl4[q * T];
      else   // (straight from the global table: entry n = l + T m)
        g4 = make_uint4(a.gidx[l + T * (4 * q)], a.gidx[l + T * (4 * q + 1)], a.gidx[l + T * (4 * q + 2)], a.gidx[l + T * (4 * q + 3)]);
      const uint32_t g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        gaddr[2 * (4 * q + j)] = sbase + (g[j] & 0xffffu);
        gaddr[2 * (4 * q + j) + 1] = sbase + (g[j] >> 16);
      }
    }
  }

#ifdef FDOCT_X_NO_RESTW
  constexpr bool RESTW = false;
#else
  constexpr bool RESTW = LEAN && KIND == 1 && STAGE != 1;
#endif
  constexpr bool ILDMA_ = IB2D && IL16 && !TRO && FDOCT_IL16_DMA;  // (= ILDMA, defined with the prefetch buffers below)
#ifdef FDOCT_X_RES_T3_ONLY  // tuning: only the 15 step-5 twiddles stay in registers
  constexpr int RES2 = 0;
  constexpr bool RES3 = LEAN && KIND == 1 && STAGE != 1;
#else
#ifdef FDOCT_TRO_RES2  // tuning: keep the step-3 twiddles resident in the transposed-store variant too (spills 7 registers)
  constexpr int RES2 = RESTW ? 12 : 0;
  constexpr bool RES3 = RESTW;
#else
  // (the transposed-store variant, and every variant that multiplies by both words of the reciprocal background, is a
  // few registers over the budget with everything resident: their 12 step-3 twiddles come from LDS every row)
  constexpr int RES2 = !RESTW ? 0 : (TRO ? 0 : (PREC ? (IL16 ? ((IB2D && !ILDMA_) ? FDOCT_PREC16_T2_IB2D : (ILDMA_ ? FDOCT_PREC16_T2_DMA : FDOCT_PREC16_T2)) : FDOCT_PREC_T2) : 12));
  // (the transposed store with a full-frame background and both words holds 48 prefetch registers: its step-5 twiddles come
  // from LDS too, or the row loop spills)
  constexpr bool RES3 = RESTW && !(TRO && IB2D && IL16 && !FDOCT_TRO_IB2D_RES3);
#endif
#endif
  v2f r_t2[RES2 ? RES2 : 1], r_t3[RES3 ? 15 : 1];
  if constexpr (RES2 > 0 || RES3) {
    v2f t2tmp[12], t3tmp[15];
    fft1024_rowswap_twiddles(lane, tw_p2, tw_p3, t2tmp, t3tmp);
    if constexpr (RES2 > 0) {
#pragma unroll
      for (int i = 0; i < RES2; i++) r_t2[i] = t2tmp[i];
    }
    if constexpr (RES3) {
#pragma unroll
      for (int i = 0; i < 15; i++) r_t3[i] = t3tmp[i];
    }
  }
  constexpr bool RESC = RES;
  static_assert(TW3_LDS || RES3, "the step-5 table stays out of LDS only where its entries are resident");
  static_assert(GI_LDS || GRES, "the gather table stays out of LDS only where its addresses are resident");
  static_assert(!(IB2D || NORM) || (fused_resident_consts(KIND, LEAN, AVG, WCH, STAGE) && STAGE == 0),
                "fast-path options: resident-constant kernels only");
  // NORM: scale/shift of input frame (o / H) * A + ai (host guarantees rows < 2^31)
  float nsc = 1.f, nsh = 0.f, nmn = 0.f;
  auto frame_scale = [&](long long o, int ai) {
    if constexpr (NORM == 1) {
      const unsigned fr = (o < a.total_out_rows) ? (unsigned)o / (unsigned)a.H : 0u;
      const float2 mmx = a.minmax[(size_t)fr * (AVG ? a.A : 1) + ai];
      nsc = (mmx.y - mmx.x > 2.220446049250313e-16f) ? 1.f / (mmx.y - mmx.x) : 0.f;
      nsh = -mmx.x * nsc;
      nmn = mmx.x;
    }
  };
  v2f r_ib[RESC ? NPR : 1], r_win[RESC ? NPR : 1], r_g[RESC ? NPR : 1];
  // IB2D + both words: the half-float pattern of the same background row is prefetched with it -- into a per-wave LDS slot by
  // the global -> LDS loads of gfx950 (global_load_lds_dwordx4: no registers held through the transform; ILDMA), or, where the
  // LDS belongs to the ring of the transposed store, into 16 registers
  constexpr bool ILDMA = IB2D && IL16 && !TRO && FDOCT_IL16_DMA;
  uint4 r_il16[(IB2D && IL16 && !ILDMA) ? WCH : 1];
  // (the slots lie behind the waves' row buffers: 2 WC bytes each)
  unsigned char* const il_dma = scratch0 + (size_t)(blockDim.x >> 6) * RPW * a.scratch_bytes + (size_t)wave * (2 * WC);
  // The averaging fast-path kernels with more than 32 samples per lane (C4): the half-float pattern of the second word stays
  // RESIDENT -- 4 registers per chunk, loaded once per wave (the kernel has them to spare since the low words' source became a
  // compile-time property) -- instead of sixteen 16-byte loads of the float low words per input A-scan from the global plane.
  constexpr bool IL16R = LEAN && PRECT && fused_il_global(LEAN, AVG, WCH, T) && FDOCT_IL16_RESIDENT && STAGE != 2;
  uint4 r_il16r[IL16R ? WCH : 1];
  if constexpr (IL16R) {
    const uint4* h4 = reinterpret_cast<const uint4*>(a.il16) + l;
#pragma unroll
    for (int c = 0; c < WCH; c++) r_il16r[c] = h4[T * c];
  }
  // IB2D: (re)load r_ib with the reciprocal-background row of output row o (any o: rows repeat every H)
  auto issue_ib2d = [&](long long o) {
    if constexpr (IB2D) {
      const unsigned rr = (o < a.total_out_rows) ? (unsigned)o % (unsigned)a.H : 0u;  // host guarantees rows < 2^31
      const float4* p4 = reinterpret_cast<const float4*>(a.ib2d + (size_t)rr * WC + 8 * l);
#pragma unroll
      for (int c = 0; c < WCH; c++) {
        const float4 q0 = p4[2 * T * c], q1 = p4[2 * T * c + 1];
        r_ib[4 * c + 0] = mk(q0.x, q0.y);
        r_ib[4 * c + 1] = mk(q0.z, q0.w);
        r_ib[4 * c + 2] = mk(q1.x, q1.y);
        r_ib[4 * c + 3] = mk(q1.z, q1.w);
      }
      if constexpr (ILDMA) {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const uint4* h4 = reinterpret_cast<const uint4*>(a.il16_2d) + (size_t)rr * (WC / 8) + l;
#pragma unroll
        for (int c = 0; c < WCH; c++)   // lane l's 16 bytes of chunk c land at slot + (c T + l) 16: the order the row top reads them in
          __builtin_amdgcn_global_load_lds((gptr_t)(h4 + T * c), (lptr_t)(il_dma + c * T * 16), 16, 0, 0);
      } else if constexpr (IL16) {
        const uint4* h4 = reinterpret_cast<const uint4*>(a.il16_2d) + (size_t)rr * (WC / 8) + l;
#pragma unroll
        for (int c = 0; c < WCH; c++) r_il16[c] = h4[T * c];
      }
    }
  };
  if constexpr (RESC) {
    // straight from the global tables (W == WC here), 32 bytes per lane and chunk, into the pair order
    auto load_plane = [&](const float* tab, int c, v2f* out) {
      const float4* p4 = reinterpret_cast<const float4*>(tab + 8 * (l + T * c));
      const float4 q0 = p4[0], q1 = p4[1];
      out[0] = mk(q0.x, q0.z);
      out[1] = mk(q1.x, q1.z);
      out[2] = mk(q0.y, q0.w);
      out[3] = mk(q1.y, q1.w);
    };
#pragma unroll
    for (int c = 0; c < WCH; c++) {
      if constexpr (!IB2D) load_plane(a.ib, c, r_ib + 4 * c);
      load_plane(a.win, c, r_win + 4 * c);
      load_plane(a.g, c, r_g + 4 * c);
    }
  }

  const long long total = a.total_out_rows;
  // Row slots: slot s of this workgroup is rows (s*gridDim.x + blockIdx.x)*RPW .. +RPW-1, so the chip sweeps the
  // batch front to back.  Waves take slots from a workgroup-wide ticket counter instead of a fixed stride: the
  // hardware favours the oldest wave of a SIMD, which with a static split finishes its share long before its
  // younger sibling and leaves the SIMD with one wave for the last quarter of the launch.
  // The ticket is claimed a row's work before it is consumed: at the row top for the prefetch in mid-row, or -- in
  // the FFT-stage kernel, which prefetches at the row top -- one whole row ahead (EARLY).
  constexpr bool EARLY = STAGE == 2;
  auto slot_row = [&](unsigned s) { return ((long long)s * gridDim.x + blockIdx.x) * RPW; };
  auto claim = [&]() -> unsigned {
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(&row_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return t;
  };
  // (An asm ds_add_rtn_u32 whose result stays in flight until it is needed would save the LDS round trip the compiler
  // waits out here at every row top -- about 0.5 % -- but nothing stops the register allocator from copying the
  // not-yet-written result register in the instantiations that are short of registers; tried and withdrawn.)
  auto ticket_value = [&](unsigned t) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)t); };
  // one row per wave (T == 64): everything row-related is wave-uniform; saying so keeps the 64-bit row arithmetic on the
  // scalar unit (the compiler cannot see it through the loop-carried ticket)
  auto uni64 = [&](long long v) -> long long {
    if constexpr (RPW == 1) {
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
      const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
      return (long long)(((unsigned long long)hi << 32) | lo);
    } else {
      return v;
    }
  };
  // TRO: ticket t of this workgroup is row t % TR of its tile t / TR (the missing rows of a short tile use up tickets that
  // name no row).
  struct TroRow {
    unsigned t, tq, rt, g, r0, nrows;  // ticket, workgroup tile number, row in tile, B-scan, first row of the tile in the B-scan, rows of the tile
    bool dummy;                        // (four rows per wave) a short tile has no rows for this wave: it recomputes rows of the tile and stores nothing
  };
  // (RPW > 1, round 6: a claim names RPW consecutive rows -- RPW divides the tile and, host-checked, the frame height, so the
  // rows of a claim are all there or all missing; t, rt count ROWS: the first row of the claim)
  auto tro_map = [&](unsigned tc, TroRow& tr) -> int {  // tc: the claimed ticket.  0: rows; 1: no such rows in this (short) tile; 2: past the end of the batch
    const unsigned t = tc * (unsigned)RPW;
    tr.t = t;
    tr.tq = t / TR;
    tr.rt = t % TR;
    if (!tro_tile(tr.tq, tr.g, tr.r0, tr.nrows)) return 2;
    return tr.rt < tr.nrows ? 0 : 1;
  };
  auto tro_take = [&](unsigned t, TroRow& tr) -> long long {  // t: a claimed ticket (uniform); claims on while tickets name no row
    for (;;) {
      const int k = tro_map(t, tr);
      if (k == 0) return (long long)tr.g * a.H + (tr.r0 + tr.rt);
      if (k == 2) return a.total_out_rows;
      t = ticket_value(claim());
    }
  };
  // ---- four rows per wave: tiles are owned by GROUPS of four waves (see grp_* above).  Wave m of a group takes rows 4 m .. 4 m + 3
  // of the group's tile; in a short last tile of a B-scan (4, 8 or 12 rows) the waves without rows recompute rows of the tile and
  // deposit nothing, so that every wave of a group walks through the same two meetings per tile.
  const unsigned grp = (unsigned)wave / GW, mem = (unsigned)wave % GW;
  unsigned grp_seq = 0u;   // tiles this group has finished
  auto grp_rows = [&](unsigned tq, TroRow& tr) -> long long {
    tr.t = 0u;
    tr.tq = tq;
    if (!tro_tile(tq, tr.g, tr.r0, tr.nrows)) return a.total_out_rows;
    const unsigned rt = 4u * mem;
    tr.dummy = rt >= tr.nrows;
    tr.rt = tr.dummy ? rt % tr.nrows : rt;
    return (long long)tr.g * a.H + (tr.r0 + tr.rt);
  };
  // the tile after the current one: the leader has claimed it at the row top and publishes it, the others wait for it (it was
  // claimed most of a row ago; the bound is the exit condition a spinning wave must have all the same)
  auto grp_next_tile = [&](unsigned leader_ticket) -> unsigned {
    const unsigned seq = grp_seq + 1u;
    unsigned tq;
    if (mem == 0u) {
      tq = ticket_value(leader_ticket);
      if (lane == 0) {
        __hip_atomic_store(&grp_tile[grp][seq & 1u], tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&grp_pub[grp], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (LDS operations of a wave execute in order)
      }
    } else {
      for (unsigned spin = 0;; spin++) {
        const unsigned pub = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&grp_pub[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (pub >= seq) break;
        if (spin >= FDOCT_TRO_SPIN_LIMIT) {
          if (lane == 0) __hip_atomic_store(a.tr_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      tq = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&grp_tile[grp][seq & 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    return tq;
  };
  auto grp_meet = [&](unsigned int* counter, unsigned target) {   // every wave of the group adds one; all wait for the group's total
    if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (unsigned spin = 0;; spin++) {
      const unsigned have = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (have >= target) break;
      if (spin >= FDOCT_TRO_SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store(a.tr_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  };
  TroRow tro_cur{}, tro_next{};
  long long o_wave;  // wave-uniform
  if constexpr (TRO_INPLACE)
    o_wave = grp_rows(grp, tro_cur);   // the groups' first tiles are their indices (row_ticket starts behind them)
  else if constexpr (TRO)
    o_wave = tro_take((unsigned)wave, tro_cur);
  else
    o_wave = slot_row((unsigned)wave);
  auto claim_row = [&]() -> unsigned {   // (four rows per wave: only a group's leader claims -- tiles, not rows)
    if constexpr (TRO_INPLACE) return mem == 0u ? claim() : 0u;
    return claim();
  };
  unsigned ticket = EARLY ? claim_row() : 0u;

  const int W = LEAN ? WC : a.W;
  // staging layout (even/odd sample planes when the gather stride is about 2): fixed by the plan on the fast path,
  // where W == WC and N == NC or 2 NC (the same rule as the host's select_plan)
  constexpr bool SPLIT_LEAN = (2 * WC >= 3 * NC) && (WC <= 3 * NC);
  const bool split = LEAN ? SPLIT_LEAN : (a.split != 0);
  const int A = AVG ? a.A : 1;  // AVG == false: compiled for one frame per output (no frame arithmetic at all)
  const unsigned char* frames = static_cast<const unsigned char*>(a.frames);
  const int i0l = 8 * l;  // this lane's sample offset inside a chunk
  const int c0l = 4 * l;  // this lane's slot inside a constant plane (see the staging loop above)

  RawChunk<IN_T> raw[WCH];
  // PF2 (fdoct_kernels.h, fused_tro_pf2): TWO sets of sample registers.  Row r's samples live in set r & 1; they are asked for
  // in the middle of row r - 2 -- into the set row r - 2 unpacked at its top -- by loads the COMPILER DOES NOT TRACK (inline asm),
  // because the wait they need is one it cannot express: in front of row r - 1's write-out, for the samples of row r only, with
  // the stores of row r - 2's write-out and the loads of row r + 1 -- all younger -- still outstanding: s_waitcnt vmcnt(N) with
  // N = the number of those younger instructions, counted as they are issued (pf2_stores + WCH; a smaller N is always safe).
  // A wave's vector-memory operations return in order: with N exact, the wait covers the stores issued TWO rows ago and older,
  // which have had 10 us to be acknowledged (EXPERIMENTS.md section 5), instead of one row's 5 us.
  typedef unsigned pf2_u4 __attribute__((ext_vector_type(4)));
  [[maybe_unused]] pf2_u4 pf2_s0[PF2 ? WCH : 1], pf2_s1[PF2 ? WCH : 1];
  [[maybe_unused]] bool pf2_ph = false;   // the set the CURRENT row's samples were unpacked from (wave-uniform)
  // One asm statement per set, executed on EVERY row with the branch INSIDE it: a conditional around an asm that writes the set
  // makes the register allocator merge two versions of the set behind it -- with copies of registers whose loads are in flight
  // (seen in the first build: v_mov of the whole set right behind the loads).  `skip` != 0: the statement does nothing.
  auto pf2_load = [&](pf2_u4* set, const void* row, unsigned skip) {
    static_assert(!PF2 || (T == 64 && WCH == 4 && sizeof(IN_T) == 2), "PF2: 16 bytes per lane and chunk, four chunks 1 KB apart");
    const unsigned voff = 16u * (unsigned)l;
    constexpr int C1 = WCH > 1 ? 1 : 0, C2 = WCH > 2 ? 2 : 0, C3 = WCH > 3 ? 3 : 0;
    asm volatile(
        "s_cmp_lg_u32 %[skip], 0\n\t"
        "s_cbranch_scc1 .Lpf2_skip_%=\n\t"
        "global_load_dwordx4 %[a], %[off], %[base] nt\n\t"
        "global_load_dwordx4 %[b], %[off], %[base] offset:1024 nt\n\t"
        "global_load_dwordx4 %[c], %[off], %[base] offset:2048 nt\n\t"
        "global_load_dwordx4 %[d], %[off], %[base] offset:3072 nt\n"
        ".Lpf2_skip_%=:"
        : [a] "+v"(set[0]), [b] "+v"(set[C1]), [c] "+v"(set[C2]), [d] "+v"(set[C3])
        : [off] "v"(voff), [base] "s"(row), [skip] "s"(skip)
        : "scc");
  };
  // wait until at most n of this wave's vector-memory operations are outstanding (n wave-uniform; the ladder rounds it DOWN)
  auto pf2_wait = [&](unsigned n) {
    if (n >= 36u) asm volatile("s_waitcnt vmcnt(36)");
    else if (n >= 28u) asm volatile("s_waitcnt vmcnt(28)");
    else if (n >= 20u) asm volatile("s_waitcnt vmcnt(20)");
    else if (n >= 16u) asm volatile("s_waitcnt vmcnt(16)");
    else if (n >= 12u) asm volatile("s_waitcnt vmcnt(12)");
    else if (n >= 8u) asm volatile("s_waitcnt vmcnt(8)");
    else if (n >= 4u) asm volatile("s_waitcnt vmcnt(4)");
    else asm volatile("s_waitcnt vmcnt(0)");
  };
  auto pf2_pin = [&](pf2_u4* set) {   // (the registers of a set that has landed: nothing that reads them moves above this)
#pragma unroll
    for (int c = 0; c < (PF2 ? WCH : 1); c++) asm volatile("" : "+v"(set[c]));
  };
  auto issue_loads = [&](long long o, int avg_i) {
    const bool valid = o < total;
    long long in_row = valid ? o : 0;
    if constexpr (AVG) {
      if (A > 1 && valid) {  // averaging: output row o = (group g, row r) reads frames g*A .. g*A+A-1
        const long long g = o / a.H;
        in_row = (g * A + avg_i) * (long long)a.H + (o - g * a.H);
      }
    }
    const void* row = frames + uni64(in_row) * a.pitch_bytes;
    if constexpr (PF2) {   // into the set the current row has unpacked (its loop-top copy is done)
      const unsigned into1 = (unsigned)__builtin_amdgcn_readfirstlane(pf2_ph ? 1 : 0);
      pf2_load(pf2_s0, row, into1);
      pf2_load(pf2_s1, row, into1 ^ 1u);
      return;
    }
#pragma unroll
    for (int c = 0; c < WCH; c++) {
      const int i0 = i0l + 8 * T * c;
      if ((LEAN || i0 < W) && !FDOCT_ABL(256))
        raw[c].load(row, i0);
      else
        raw[c].zero();
    }
  };

  // The prefetched row is "pinned" (an empty asm that names its registers) right after the
  // magnitude step and BEFORE the row's stores: the s_waitcnt the compiler needs there is
  // vmcnt(0) with the loads as the youngest VMEM operations, and the stores that follow get a
  // whole row of work to drain.  A wait at the loop top would also wait for those stores.
  // STAGE 2 streams the packed k-linear rows instead of camera samples
  v2f znext[STAGE == 2 ? P : 1];
  auto issue_zloads = [&](long long o, int avg_i) {
    if constexpr (STAGE == 2) {
      const bool valid = o < total;
      long long in_row = valid ? o : 0;
      if constexpr (AVG) {
        if (A > 1 && valid) {  // averaging: the resample stage left one k-linear row per INPUT A-scan (frame g*A + avg_i)
          const long long g = o / a.H;
          in_row = (g * A + avg_i) * (long long)a.H + (o - g * a.H);
        }
      }
      const v2f* zr = reinterpret_cast<const v2f*>(a.ylin) + uni64(in_row) * NC + l;
#pragma unroll
      for (int m = 0; m < P; m++) znext[m] = valid ? __builtin_nontemporal_load(zr + T * m) : mk(0.f, 0.f);
    }
  };
  // Two-word division on the fast-path kernels with at most 32 samples per lane (PREC): the 32 low words of a lane are
  // row-invariant, but there is no register to keep them in through the transform; they are read from the workgroup's LDS
  // plane at every row top (ilx).  (Re-loading them from a global plane behind the previous pass's magnitudes instead -- older
  // than the row's stores, so that the wait never includes a store -- was measured and lost: 489-491 against 509 M A-scans/s
  // on C2, tools/ab.sh.  The LDS-bound averaging kernels with more samples per lane do read them from global memory: prec == 3.)
  constexpr bool ILX = PREC && LEAN && WCH <= 4;
  static_assert(!IL16 || ILX, "half-float second word: the kernels that read the row's low words at its top");
  v2f ilx[(ILX && !IL16) ? NPR : 1];
  uint4 ilh[IL16 ? WCH : 1];  // IL16: chunk c's four half-float pairs (pair q = dword q), 16 registers instead of 32
  [[maybe_unused]] long long pf2_o1 = total, pf2_o2 = total;   // PF2: the rows after the current one (tro_next, tro_n2)
  [[maybe_unused]] TroRow tro_n2{};
  if (o_wave < total) {
    // (the first row like every later one: the full-frame background's rows -- and the global -> LDS load of its half-float
    // pattern -- are issued BEFORE the samples, so that a wait for the samples covers them: loads return in order)
    issue_ib2d(o_wave + sub);
    if constexpr (STAGE == 2)
      issue_zloads(o_wave + sub, 0);
    else
      issue_loads(o_wave + sub, 0);
    frame_scale(o_wave + sub, 0);
    if constexpr (PF2) {   // the wave's second row, into the other set; both have landed before the loop
      pf2_o1 = tro_take(ticket_value(claim()), tro_next);
      pf2_ph = true;
      issue_loads(pf2_o1 + sub, 0);
      pf2_ph = false;
      asm volatile("s_waitcnt vmcnt(0)");
      pf2_pin(pf2_s0);
      pf2_pin(pf2_s1);
    }
  }


  FusedProbe pr;
#ifdef FDOCT_FUSED_PROBE
  pr.start();
#endif
  while (o_wave < total) {
    o_wave = uni64(o_wave);
    const long long o = o_wave + sub;
    const bool valid = o < total;
    long long o_next = total;
    if constexpr (EARLY) {
      o_next = slot_row(ticket_value(ticket));  // claimed one row ago
      ticket = claim_row();
    } else {
      ticket = claim_row();  // back long before the prefetch below needs it
    }
    long long gi = 0;  // output group (frame when A == 1) and row inside the frame
    int r = 0;
    if constexpr (!LEAN) {
      if (a.need_rc && valid) {
        gi = o / a.H;
        r = (int)(o - gi * a.H);
      }
    }

    float acc[P];
#pragma unroll
    for (int m = 0; m < P; m++) acc[m] = 0.f;

    for (int ai = 0; ai < A; ai++) {
      v2f z[P];
      if constexpr (STAGE == 2) {
#pragma unroll
        for (int m = 0; m < P; m++) z[m] = znext[m];
        if (ai + 1 < A)
          issue_zloads(o, ai + 1);
        else
          issue_zloads(o_next + sub, 0);
      } else {
      // ---------------- A2: dark, normalise, pi frame, background
      // (fast path, low words of the reciprocal background in LDS: their reads are issued here, ahead of everything the row
      // does with them -- the samples are still packed, so this is where registers are to spare)
      if constexpr (IL16) {
        if constexpr (ILDMA) {
          // The slot was filled by global_load_lds a row ago.  Explicit wait, not the compiler's own tracking of
          // LDS writes by that instruction: the samples in `raw` were issued AFTER the DMA and loads return in order, so an
          // empty asm that consumes a sample register makes the compiler wait for it here, and its memory clobber keeps the
          // LDS reads below it.  (For every row but the first that wait already happened at the pin() in front of the
          // previous row's stores: no cost.)
          raw[0].pin_ordered();
          const uint4* h4 = reinterpret_cast<const uint4*>(il_dma) + l;
#pragma unroll
          for (int c = 0; c < WCH; c++) ilh[c] = h4[T * c];
        } else if constexpr (IB2D) {
#pragma unroll
          for (int c = 0; c < WCH; c++) ilh[c] = r_il16[c];
        } else {
          const uint4* h4 = reinterpret_cast<const uint4*>(c_il) + l;
#pragma unroll
          for (int c = 0; c < WCH; c++) ilh[c] = h4[T * c];
        }
      } else if constexpr (ILX) {
#pragma unroll
        for (int c = 0; c < WCH; c++) load_consts<T>(c_il + c0l, c, ilx + 4 * c);
      }
      v2f v[NPR];  // sample pairs: v[4c+q] = samples 8*(l+T*c) + chunk_pair_offset(q), +2
      if constexpr (PF2) {   // this row's samples: landed before the previous row's write-out (the wait in front of it)
        if constexpr (sizeof(IN_T) == 2) {
#pragma unroll
          for (int c = 0; c < WCH; c++) {
            const pf2_u4 t = pf2_ph ? pf2_s1[c] : pf2_s0[c];
            raw[c].v = make_uint4(t.x, t.y, t.z, t.w);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < WCH; c++) raw[c].unpack(v + 4 * c);
      // Fast-path normalisations.  With the two-word division (PREC) the normalised sample p = (v - min) * scale is carried as two
      // floats as well -- rounding it is a rounding at the size of the DC level, random from sample to sample -- and formed
      // inside the division below (NPREC: v stays the camera sample here, nmn / nsc are the row's or the frame's).
      constexpr bool NPREC = NORM != 0 && PREC;
      if constexpr (NORM == 1 && !NPREC) {  // main:1128-1129, same expression as the general kernel below
#pragma unroll
        for (int i = 0; i < NPR; i++) v[i] = pk_fma(v[i], mk(nsc, nsc), mk(nsh, nsh));
      }
      if constexpr (NORM == 2) {  // main:88-97, 1126: the general kernel's row-wise normalisation, W == WC here
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NPR; i++) {
          mn = fminf(mn, fminf(v[i].x, v[i].y));
          mx = fmaxf(mx, fmaxf(v[i].x, v[i].y));
        }
        mn = group_min<T>(mn);
        mx = group_max<T>(mx);
        const float sc = (mx - mn > 2.220446049250313e-16f) ? 1.f / (mx - mn) : 0.f;
        if constexpr (NPREC) {
          nmn = mn;
          nsc = sc;
        } else {
          const float sh = -mn * sc;
#pragma unroll
          for (int i = 0; i < NPR; i++) v[i] = pk_fma(v[i], mk(sc, sc), mk(sh, sh));
        }
      }

      bool gnorm = false;  // any-option kernel: a normalisation is on (nmn, nsc hold its minimum and scale)
      if constexpr (!LEAN) {
        const long long in_frame = gi * A + ai;
        if (a.yd) {
          const float* ydr = a.yd + (a.yd_2d ? (size_t)r * W : 0);
#pragma unroll
          for (int c = 0; c < WCH; c++) {
            const int i0 = i0l + 8 * T * c;
            if (i0 < W) {
#pragma unroll
              for (int p = 0; p < 4; p++) v[4 * c + p] -= mk(ydr[i0 + chunk_pair_offset(p)], ydr[i0 + chunk_pair_offset(p) + 2]);
            }
          }
        }
        if (a.rowwisenormalize) {  // main:88-97,1126
          float mn = INFINITY, mx = -INFINITY;
#pragma unroll
          for (int c = 0; c < WCH; c++) {
            if (i0l + 8 * T * c < W) {
#pragma unroll
              for (int p = 0; p < 4; p++) {
                mn = fminf(mn, fminf(v[4 * c + p].x, v[4 * c + p].y));
                mx = fmaxf(mx, fmaxf(v[4 * c + p].x, v[4 * c + p].y));
              }
            }
          }
          mn = group_min<T>(mn);
          mx = group_max<T>(mx);
          // (the normalised sample (v - min) * scale is formed inside the division below, as two floats: see there)
          gnorm = true;
          nmn = mn;
          nsc = (mx - mn > 2.220446049250313e-16f) ? 1.f / (mx - mn) : 0.f;
        } else if (a.minmax) {  // main:1128-1129 whole-frame min-max, from the pre-pass
          const float2 mmx = a.minmax[in_frame];
          gnorm = true;
          nmn = mmx.x;
          nsc = (mmx.y - mmx.x > 2.220446049250313e-16f) ? 1.f / (mmx.y - mmx.x) : 0.f;
        }
      }
      // main:1132 ... / data_yb as a multiply by the host-side reciprocal; main:1138 row mean; main:1142 window; and the
      // slope step of main:1153-1173.  With x = v / yb, t = x - mean and y = t * w, the reference's
      //   s_i = y_i + g_i * (y_i - y_(i-1))        (g_i = fractionalk by SAMPLE index, see A5 below)
      // is  s_i = a_i * t_i + b_i * t_(i-1),       a_i = (1 + g_i) w_i,  b_i = -g_i w_(i-1)
      // so the host folds window and slope weights into the two per-sample planes a (a.win) and b (a.g): two packed
      // instructions per sample pair instead of four.  Sample 0 has slopes[0] = slopes[1] (main:1161):
      // a_0 = (1 - g_0) w_0, b_0 = +g_0 w_1, and t_1 stands in for t_(-1).
      float mh = 0.f, ml = 0.f;
      v2f av[NPR], bv[NPR];
      if (!FDOCT_ABL(1)) {
        v2f ibv[NPR];
        bool from_lds = true;
        if constexpr (!LEAN) {
          if (a.ib2d) {
            from_lds = false;
#pragma unroll
            for (int c = 0; c < WCH; c++) {
              const int i0 = i0l + 8 * T * c;
              if (i0 < W) {
                const float4* p4 = reinterpret_cast<const float4*>(a.ib2d + (size_t)r * WC + i0);  // rows are padded to WC
                const float4 q0 = p4[0], q1 = p4[1];
                ibv[4 * c + 0] = mk(q0.x, q0.y);  // a.ib2d holds each 8-sample group evens first, then odds
                ibv[4 * c + 1] = mk(q0.z, q0.w);
                ibv[4 * c + 2] = mk(q1.x, q1.y);
                ibv[4 * c + 3] = mk(q1.z, q1.w);
              } else {
#pragma unroll
                for (int p = 0; p < 4; p++) ibv[4 * c + p] = mk(0.f, 0.f);
              }
            }
          }
        }
        // fast path with at most 32 samples per lane: 1/background stays in registers through the sum and the
        // subtraction (resident, or read once from the LDS plane), so x is never formed -- the row sum is accumulated
        // by fma (four chains), t = fma(v, 1/yb, -mh) - ml rounds once, and the mean needs no f64 (group_mean_f32)
        constexpr bool FMAX = LEAN && WCH <= 4;
        if constexpr (FMAX) {
          if constexpr (RESC) {
#pragma unroll
            for (int i = 0; i < NPR; i++) ibv[i] = r_ib[i];
          } else {
#pragma unroll
            for (int c = 0; c < WCH; c++) load_consts<T>(c_ib + c0l, c, ibv + 4 * c);
#pragma unroll
            for (int c = 0; c < WCH; c++) load_consts<T>(c_win + c0l, c, av + 4 * c);
          }
#ifdef FDOCT_X_OLD_MEAN  // tuning: round 2's form (lane sums of the DC-sized products, two-float mean)
          v2f s4[4] = {mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f)};
#pragma unroll
          for (int c = 0; c < WCH; c++) {
#pragma unroll
            for (int p = 0; p < 4; p++) s4[p] = pk_fma(v[4 * c + p], ibv[4 * c + p], s4[p]);
          }
          const v2f part = (s4[0] + s4[1]) + (s4[2] + s4[3]);
          group_mean_f32<T>(part.x + part.y, 1.f / (float)T, 1.f / (float)(8 * WCH), 1.f / (float)WC, mh, ml);
#else
          // Row mean without any DC-sized sum: c0, the average of one x = v / yb per lane, is a wave-uniform estimate
          // of the mean; d = fma(v, 1/yb, -c0) is the exact product minus c0 rounded at the size of the DEVIATION from it
          // (fringes, residual envelope), so are the sums of d, and x - mean = d - mean(d).  Same operation count as summing
          // the products (whose lane sums of ~ 8 WCH x mean rounded at the size of the DC level and left 1e-8 of it in the mean:
          // 4e-6 of the DC level in depth bins 0 and 1, above the tolerance once the fringes are weaker than ~2 % of the DC level).
          // 1/yb is the two-float sum ib + il (fdoct_capi.cpp::reciprocal_words).  The f32 reciprocal alone is off by up to
          // 6e-8 of the quotient: a fixed per-column pattern of the size of the DC level, <= 4e-6 of it per depth bin -- above the
          // tolerance for fringes weaker than 1 % of the DC level.  PREC (the two-word instantiations, fdoct_set_precise_division):
          // a second fma adds v * il, rounded at the size of the deviation like the first; the low words come from the
          // workgroup's LDS plane (there is no register left to keep them resident), read at the row top (ilx).
          // (the 64 samples 8 l of the first chunk.  The middle chunk was tried in round 6 -- it is what the any-option kernel below
          // takes its single-chunk estimate from -- and left C2's flat 1e-3 dB pass rate at 0.9999 instead of 1: kept as it was.)
          const float c0 = group_sum_f32<T>((NPREC ? (v[0].x - nmn) * nsc : v[0].x) * ibv[0].x) * (1.f / (float)T);
          // IL16: what the first word leaves out, v * il = (v * ib) * rho = (c0 + d) * rho with rho = il / ib (|rho| <= 2^-24), is
          // c0 * rho up to d * rho -- below the rounding of d.  rho * 2^38 comes as half floats (ten bits of a correction that is
          // 8 x the tolerance at fringes of 1e-3 of the DC level), v_fma_mix_f32 converts them inside the fma.
          const float c0s = c0 * 3.637978807091713e-12f;  // 2^-38 (kPrec16Shift)
          auto second_word = [&](v2f d, int i) -> v2f {
            if constexpr (IL16) {
              const uint4 hq = ilh[i >> 2];
              const uint32_t hp = (i & 3) == 0 ? hq.x : (i & 3) == 1 ? hq.y : (i & 3) == 2 ? hq.z : hq.w;
              return mk(fma_mix_lo(c0s, hp, d.x), fma_mix_hi(c0s, hp, d.y));
            } else {
              return d;
            }
          };
          if constexpr (NPREC) {
            // p = (v - min) * scale as two floats (v - min is exact on the camera's integer samples; p_lo = the product's exact
            // residual), times ib + il: nothing of normalise-and-divide rounds at the size of the DC level
#pragma unroll
            for (int i = 0; i < NPR; i++) {
              const v2f vm = v[i] - mk(nmn, nmn);
              const v2f ph = vm * mk(nsc, nsc);
              const v2f pl = pk_fma(vm, mk(nsc, nsc), -ph);
              if constexpr (IL16)
                v[i] = pk_fma(pl, ibv[i], second_word(pk_fma(ph, ibv[i], mk(-c0, -c0)), i));
              else
                v[i] = pk_fma(pl, ibv[i], pk_fma(ph, ilx[i], pk_fma(ph, ibv[i], mk(-c0, -c0))));
            }
          } else if constexpr (IL16) {
#pragma unroll
            for (int i = 0; i < NPR; i++) v[i] = second_word(pk_fma(v[i], ibv[i], mk(-c0, -c0)), i);
          } else if constexpr (PREC) {
#pragma unroll
            for (int i = 0; i < NPR; i++) v[i] = pk_fma(v[i], ilx[i], pk_fma(v[i], ibv[i], mk(-c0, -c0)));
          } else {
#pragma unroll
            for (int i = 0; i < NPR; i++) v[i] = pk_fma(v[i], ibv[i], mk(-c0, -c0));
          }
          v2f s4[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
          for (int c = 1; c < WCH; c++) {
#pragma unroll
            for (int p = 0; p < 4; p++) s4[p] += v[4 * c + p];
          }
          const v2f part = (s4[0] + s4[1]) + (s4[2] + s4[3]);
          mh = group_sum_f32<T>(part.x + part.y) * (1.f / (float)WC);  // mean of d (W == WC on this path)
#endif
          if constexpr (RESC) {
#pragma unroll
            for (int i = 0; i < NPR; i++) {
              av[i] = r_win[i];
              bv[i] = r_g[i];
            }
          } else {
#pragma unroll
            for (int c = 0; c < WCH; c++) load_consts<T>(c_g + c0l, c, bv + 4 * c);
          }
#ifdef FDOCT_X_OLD_MEAN
#pragma unroll
          for (int i = 0; i < NPR; i++) v[i] = pk_fma(v[i], ibv[i], mk(-mh, -mh)) - mk(ml, ml);
#else
#pragma unroll
          for (int i = 0; i < NPR; i++) v[i] -= mk(mh, mh);
#endif
        } else {
          // WCH <= 4: all reciprocal-background reads are issued up front (one LDS wait); wider rows read
          // them chunk by chunk to stay inside the register budget
          if (from_lds && WCH <= 4) {
#pragma unroll
            for (int c = 0; c < WCH; c++) load_consts<T>(c_ib + c0l, c, ibv + 4 * c);
          }
          double sum = 0.0;
#ifndef FDOCT_X_OLD_MEAN
          constexpr bool CMEAN = LEAN;  // the fast-path rows that are too wide for the block above (C4, C1): the same mean, see there
#else
          constexpr bool CMEAN = false;
#endif
          float c0 = 0.f;
          [[maybe_unused]] bool have_c0 = false;
          v2f s4[4] = {mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f)};
#pragma unroll
          for (int cc = 0; cc < WCH; cc++) {
            // (any-option kernel: the chunks from the MIDDLE of the row on -- c0 comes from the first chunk taken, see below)
            const int c = CMEAN ? cc : (cc + WCH / 2) % WCH;
            if (from_lds && WCH > 4) load_consts<T>(c_ib + c0l, c, ibv + 4 * c);
            // low words of 1/background (see the block above): from the LDS plane, or -- full-frame background -- from the
            // frame's own row in global memory
            // (where the low words come from is a COMPILE-TIME property of the fast-path kernels -- fused_il_global, one rule for
            // kernel and host: a run-time branch on a.prec here made the compiler zero and merge the eight registers of every
            // chunk, 130 v_mov per input A-scan of C4 -- round 5)
            v2f ilv[4] = {mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f)};
            if constexpr (IL16R) {
              // (resident half-float pattern: applied below as c0 * rho)
            } else if constexpr (LEAN && PREC && fused_il_global(LEAN, AVG, WCH, T)) {
              // (averaging fast-path kernels that are short of LDS: the same plane, in the same order, from global memory -- L1 / L2
              // hits; the wave's stores, which such a load would have to wait behind, come once per A input rows there)
              load_consts<T>(a.ilp + c0l, c, ilv);
            } else if constexpr (LEAN && PREC) {
              load_consts<T>(c_il + c0l, c, ilv);
            } else if (!LEAN && a.prec == 1) {
              load_consts<T>(c_il + c0l, c, ilv);
            } else if constexpr (!LEAN) {
              if (a.prec == 2 && i0l + 8 * T * c < W) {
                const float4* p4 = reinterpret_cast<const float4*>(a.il2d + (size_t)r * WC + i0l + 8 * T * c);
                const float4 q0 = p4[0], q1 = p4[1];
                ilv[0] = mk(q0.x, q0.y);
                ilv[1] = mk(q0.z, q0.w);
                ilv[2] = mk(q1.x, q1.y);
                ilv[3] = mk(q1.z, q1.w);
              }
            }
            if constexpr (CMEAN) {
              if (c == 0) c0 = group_sum_f32<T>(v[0].x * ibv[0].x) * (1.f / (float)T);
              if constexpr (IL16R) {   // what the first word leaves out is c0 * rho up to d * rho (see the fast path above)
                const float c0s = c0 * 3.637978807091713e-12f;  // 2^-38 (kPrec16Shift)
                const uint32_t hq[4] = {r_il16r[c].x, r_il16r[c].y, r_il16r[c].z, r_il16r[c].w};
#pragma unroll
                for (int p = 0; p < 4; p++) {
                  const v2f d = pk_fma(v[4 * c + p], ibv[4 * c + p], mk(-c0, -c0));
                  v[4 * c + p] = mk(fma_mix_lo(c0s, hq[p], d.x), fma_mix_hi(c0s, hq[p], d.y));
                }
              } else if constexpr (PREC) {
#pragma unroll
                for (int p = 0; p < 4; p++) v[4 * c + p] = pk_fma(v[4 * c + p], ilv[p], pk_fma(v[4 * c + p], ibv[4 * c + p], mk(-c0, -c0)));
              } else {
#pragma unroll
                for (int p = 0; p < 4; p++) v[4 * c + p] = pk_fma(v[4 * c + p], ibv[4 * c + p], mk(-c0, -c0));
              }
#pragma unroll
              for (int p = 0; p < 4; p++) s4[p] += v[4 * c + p];
            } else {
              // (any-option kernel: the same deviation form with the f64 sum kept; chunks past the end of a narrow row stay zero.
              // c0 = the average of v / yb over the group's samples of the first chunk taken that has any -- the middle of a full
              // row.  Any c0 is correct; a POOR one costs precision: d = v / yb - c0 is rounded at its own size, and the row's first
              // sample -- round 5's c0 -- sits in the tail of the source spectrum, where a few hundred counts make v / yb noisy
              // at 1e-3: 0.6 x the tolerance on rows whose fringes a moving average had all but cancelled, round 6's sweeps.)
              const bool in_row = LEAN || (i0l + 8 * T * c < W);
              // a normalisation: p = (v - min) * scale as two floats (rounded product and its exact residual), so that the
              // normalised sample is not rounded at the size of the DC level; then the pi frame (main:1132: data_y - data_yp)
              v2f plo[4] = {mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f)};
              if constexpr (!LEAN) {
                // frames handed over as doubles (main:987): the samples' low words, carried like the normalisation's
                if constexpr (std::is_same<IN_T, float>::value) {
                  if (a.frames_lo && in_row) {
                    const float* lr = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.frames_lo) +
                                                                     ((gi * A + ai) * (long long)a.H + r) * a.pitch_bytes) + i0l + 8 * T * c;
#pragma unroll
                    for (int p = 0; p < 4; p++) plo[p] = mk(lr[chunk_pair_offset(p)], lr[chunk_pair_offset(p) + 2]);
                  }
                }
                // Round 6: none of the subtractions on the way rounds at the size of the DC level -- the exact residual of
                // the sample less a (non-integer) dark frame, less the normalisation's minimum, less the pi frame (two_diff)
                // joins the low word.  The dark difference itself was formed above (the row-wise min / max need it); its
                // residual is taken here from the samples still held in `raw`.
                if (a.yd && in_row) {
                  const float* ydr = a.yd + (a.yd_2d ? (size_t)r * W : 0) + i0l + 8 * T * c;
                  v2f orig[4];
                  raw[c].unpack(orig);
#pragma unroll
                  for (int p = 0; p < 4; p++) {
                    v2f e;
                    (void)two_diff(orig[p], mk(ydr[chunk_pair_offset(p)], ydr[chunk_pair_offset(p) + 2]), e);
                    plo[p] += e;
                  }
                }
                if (gnorm) {
#pragma unroll
                  for (int p = 0; p < 4; p++) {
                    v2f e;
                    const v2f vm = two_diff(v[4 * c + p], mk(nmn, nmn), e);
                    const v2f vml = plo[p] + e;
                    v[4 * c + p] = vm * mk(nsc, nsc);
                    plo[p] = pk_fma(vml, mk(nsc, nsc), pk_fma(vm, mk(nsc, nsc), -v[4 * c + p]));
                  }
                }
                if (a.yp && in_row) {
                  const float* ypr = a.yp + (a.yp_2d ? (size_t)r * W : 0) + i0l + 8 * T * c;
#pragma unroll
                  for (int p = 0; p < 4; p++) {
                    v2f e;
                    v[4 * c + p] = two_diff(v[4 * c + p], mk(ypr[chunk_pair_offset(p)], ypr[chunk_pair_offset(p) + 2]), e);
                    plo[p] += e;
                  }
                }
              }
              if (!have_c0 && (LEAN || 8 * T * c < W)) {
                const float cnt = group_sum_f32<T>(in_row ? 1.f : 0.f);
                c0 = group_sum_f32<T>(in_row ? v[4 * c].x * ibv[4 * c].x : 0.f) / cnt;   // (cnt >= 1: the chunk's first lane is in the row)
                have_c0 = true;
              }
#pragma unroll
              for (int p = 0; p < 4; p++)
                v[4 * c + p] = in_row ? pk_fma(plo[p], ibv[4 * c + p], pk_fma(v[4 * c + p], ilv[p], pk_fma(v[4 * c + p], ibv[4 * c + p], mk(-c0, -c0)))) : mk(0.f, 0.f);
              const v2f part = (v[4 * c] + v[4 * c + 1]) + (v[4 * c + 2] + v[4 * c + 3]);
              sum += (double)(part.x + part.y);
            }
          }
          // the a plane: issued here so its LDS latency hides under the mean reduction (WCH <= 4)
          if constexpr (WCH <= 4) {
#pragma unroll
            for (int c = 0; c < WCH; c++) load_consts<T>(c_win + c0l, c, av + 4 * c);
            __builtin_amdgcn_sched_barrier(0);
          }
          // ---------------- A3: DC removal (mean in double; CMEAN: mean of the deviations from c0, in float)
          if constexpr (CMEAN) {
            const v2f part = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            mh = group_sum_f32<T>(part.x + part.y) * (1.f / (float)WC);
            ml = 0.f;
          } else {
            sum = group_sum<T>(sum);
            const double mean = sum / (double)W;
            mh = (float)mean;
            ml = (float)(mean - (double)mh);
          }
          // the b plane: in flight while the mean is subtracted
          if constexpr (WCH <= 4) {
#pragma unroll
            for (int c = 0; c < WCH; c++) load_consts<T>(c_g + c0l, c, bv + 4 * c);
          }
          if constexpr (CMEAN) {
#pragma unroll
            for (int i = 0; i < NPR; i++) v[i] -= mk(mh, mh);
          } else {
#pragma unroll
            for (int i = 0; i < NPR; i++) v[i] = (v[i] - mk(mh, mh)) - mk(ml, ml);
          }
        }
      }
      FDOCT_FPR(pr, 0);   // row top: the prefetched samples' arrival, unpack, A2 / A3 (division, mean)
      // ---------------- A3 (window) + A5 (first half): s_i = a_i t_i + b_i t_(i-1)
      // (the reference weights the slope by fractionalk[nearestkindex[q]], a per-SAMPLE
      //  quantity, so the slope step is done here once per sample)
      if (!FDOCT_ABL(2)) {
        float* stl = stg + (split ? (i0l >> 1) : i0l);
        float prev_last = 0.f;  // t of the sample just before this lane's chunk
#pragma unroll
        for (int c = 0; c < WCH; c++) {
          if constexpr (!RESC && WCH > 4) {
            load_consts<T>(c_win + c0l, c, av + 4 * c);
            load_consts<T>(c_g + c0l, c, bv + 4 * c);
          }
          const float last = v[4 * c + 3].y;
          float left;
          if constexpr (T == 64) {
            // wave_shr:1 -- lane i takes lane i-1's last sample, lane 0 keeps `old` = the previous
            // chunk's lane-63 sample (no LDS round trip)
            left = __int_as_float(
                __builtin_amdgcn_update_dpp(__float_as_int(prev_last), __float_as_int(last), 0x138, 0xf, 0xf, false));
            if (c + 1 < WCH) prev_last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(last), 63));
          } else {
            left = __shfl_up(last, 1, T);
            if (l == 0) left = prev_last;  // last sample of the previous chunk (lane T-1)
            if (c + 1 < WCH) prev_last = __shfl(last, T - 1, T);
          }
          // e0 = (t0,t2), e1 = (t4,t6), o0 = (t1,t3), o1 = (t5,t7); the left neighbours of an odd pair ARE the even pair
          const v2f e0 = v[4 * c], e1 = v[4 * c + 1], o0 = v[4 * c + 2], o1 = v[4 * c + 3];
          v2f pE0 = mk(left, o0.x);
          const v2f pE1 = mk(o0.y, o1.x);
          if (c == 0 && l == 0) pE0.x = o0.x;  // sample 0: slopes[0] = slopes[1] (main:1161), see the planes above
          const v2f sE0 = pk_fma(av[4 * c + 0], e0, bv[4 * c + 0] * pE0);
          const v2f sE1 = pk_fma(av[4 * c + 1], e1, bv[4 * c + 1] * pE1);
          const v2f sO0 = pk_fma(av[4 * c + 2], o0, bv[4 * c + 2] * e0);
          const v2f sO1 = pk_fma(av[4 * c + 3], o1, bv[4 * c + 3] * e1);
          if (LEAN || (i0l + 8 * T * c < W)) {
            if (split) {
              *reinterpret_cast<float4*>(stl + 4 * T * c) = make_float4(sE0.x, sE0.y, sE1.x, sE1.y);
              *reinterpret_cast<float4*>(stl + 4 * T * c + WC / 2) = make_float4(sO0.x, sO0.y, sO1.x, sO1.y);
            } else {
              *reinterpret_cast<float4*>(stl + 8 * T * c) = make_float4(sE0.x, sO0.x, sE0.y, sO0.y);
              *reinterpret_cast<float4*>(stl + 8 * T * c + 4) = make_float4(sE1.x, sO1.x, sE1.y, sO1.y);
            }
          }
        }
        if (l == 0) stg[WC] = 0.f;  // source of data_ylin[0] and data_ylin[N-1] (defined 0)
      }
      wave_lds_sync();
      FDOCT_FPR(pr, 1);   // window + slope step, staging stores

      // prefetch the next row this group will need (its registers are free from here on)
      {
        long long no = o;
        int na = ai + 1;
        if (na == A) {
          na = 0;
          if constexpr (PF2) {   // the row after next; the next one's samples have been on their way since the previous row
            pf2_o2 = tro_take(ticket_value(ticket), tro_n2);
            o_next = pf2_o1;
          } else if constexpr (TRO_INPLACE)
            o_next = grp_rows(grp_next_tile(ticket), tro_next);
          else if constexpr (TRO)
            o_next = tro_take(ticket_value(ticket), tro_next);
          else
            o_next = slot_row(ticket_value(ticket));
          no = (PF2 ? pf2_o2 : o_next) + sub;
          issue_ib2d(no);  // r_ib was consumed at the top of this pass
        }
        issue_loads(no, na);
        frame_scale(no, na);
      }

      // ---------------- A5 (second half) + A6: gather into FFT registers
      static_assert(P % 4 == 0, "gather table is read four entries at a time");
      uint32_t gsrc[P];  // packed 16-bit LDS byte offsets (relative to this row's staging buffer)
      if constexpr (!GRES) {
        const uint4* gl4 = reinterpret_cast<const uint4*>(c_gi) + l;
#pragma unroll
        for (int q = 0; q < P / 4; q++) {
          const uint4 g4 = gl4[q * T];
          gsrc[4 * q + 0] = g4.x; gsrc[4 * q + 1] = g4.y; gsrc[4 * q + 2] = g4.z; gsrc[4 * q + 3] = g4.w;
        }
      }
      if (FDOCT_ABL(2)) {
#pragma unroll
        for (int m = 0; m < P; m++) z[m] = v[m % NPR];
      } else if constexpr (CPLX) {
        const v2f* phl = c_ph + l;
#pragma unroll
        for (int m = 0; m < P; m++) {
          const float y = *reinterpret_cast<const float*>(scr + (gsrc[m] & 0xffffu));
          z[m] = phl[T * m] * mk(y, y);
        }
      } else if constexpr (GRES) {
#pragma unroll
        for (int m = 0; m < P; m++) {
          const float zx = *reinterpret_cast<const __attribute__((address_space(3))) float*>(gaddr[2 * m]);
          const float zy = *reinterpret_cast<const __attribute__((address_space(3))) float*>(gaddr[2 * m + 1]);
          z[m] = mk(zx, zy);
        }
      } else {
#pragma unroll
        for (int m = 0; m < P; m++) {
          const float zx = *reinterpret_cast<const float*>(scr + (gsrc[m] & 0xffffu));
          const float zy = *reinterpret_cast<const float*>(scr + (gsrc[m] >> 16));
          z[m] = mk(zx, zy);
        }
      }
      wave_lds_sync();
      FDOCT_FPR(pr, 2);   // prefetch issue, gather (+ phase multiply)
      }  // STAGE != 2

      if constexpr (STAGE == 1) {
        // resample stage: the packed k-linear row goes to memory, 8 bytes per lane, coalesced
        if (valid) {
          v2f* zr = reinterpret_cast<v2f*>(a.ylin) + o * NC + l;
#pragma unroll
          for (int m = 0; m < P; m++) __builtin_nontemporal_store(z[m], zr + T * m);
        }
      } else {
      // ---------------- A7: NC-point inverse DFT
      if constexpr (KIND == 1) {
        if (!FDOCT_ABL(4)) fft1024_rowswap<RES2, RES3>(z, lane, xch, tw_p2, tw_p3, r_t2, r_t3, pr);
      } else if constexpr (KIND == 2) {
        if (!FDOCT_ABL(4)) fft2048_rowswap(z, lane, xch, tw_p2, tw_p3, pr);
      } else if (!FDOCT_ABL(4)) {
        constexpr int NTW2 = (P / R2) * (R2 - 1);
        constexpr int NTW3 = (R3 > 1) ? (P / R3) * (R3 - 1) : 1;
        pass_compute<NC, T, R1, 1, false, LP, true>(z, l, xch, nullptr);
        v2f tw2[NTW2];
        pass_twiddles<NC, T, R2, R1>(tw2, l, tw_p2);  // queued behind the exchange writes
        wave_lds_sync();
        pass_readback<NC, T, LP>(z, l, xch);
        wave_lds_sync();
        if constexpr (NPASS == 3) {
          pass_compute<NC, T, R2, R1, false, LP, true>(z, l, xch, tw2);
          v2f tw3[NTW3];
          pass_twiddles<NC, T, R3, R1 * R2>(tw3, l, tw_p3);
          wave_lds_sync();
          pass_readback<NC, T, LP>(z, l, xch);
          wave_lds_sync();
          pass_compute<NC, T, R3, R1 * R2, true, LP, true>(z, l, xch, tw3);
        } else {
          pass_compute<NC, T, R2, R1, true, LP, true>(z, l, xch, tw2);
        }
      }

      // ---------------- A8: magnitude (+ untangle on the real path)
      if (FDOCT_ABL(32)) {
#pragma unroll
        for (int m = 0; m < P; m++) acc[m] += z[m].x + z[m].y;
      } else if constexpr (CPLX) {
        // slot m is bin l + T*m: with D <= NC/2 (the usual half-depth display) the upper half of the slots is
        // never stored, so its magnitudes (and the epilogue below) are skipped -- a wave-uniform branch
        auto mag_slots = [&](auto lo_c, auto hi_c) {
          static_for<decltype(lo_c)::value, decltype(hi_c)::value>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const v2f q = z[m] * z[m];
            acc[m] = AVG ? acc[m] + fast_sqrt(q.x + q.y) : fast_sqrt(q.x + q.y);
          });
        };
        mag_slots(IC<0>{}, IC<P / 2>{});
        if (a.D > NC / 2) mag_slots(IC<P / 2>{}, IC<P>{});
      } else {
        // Bins k and NC-k come out of the same two values: with Zp = Z[NC-k],
        //   A = Z[k] + conj(Zp), B = Z[k] - conj(Zp), q = w^k * B:  2X[k] = A - i*q,  2|X[NC-k]| = |A + i*q|
        // (E[NC-k] = conj E[k], O[NC-k] = conj O[k], w^(NC-k) = -conj w^k; the 1/2 is folded into the
        // window on the host).  So each lane takes its LOWER-half registers m < P/2 (bins k = l + T*m <
        // NC/2), fetches the partner Z[NC-k] from lane (T-l)%T -- register P-1-m there, or (P-m)%P when
        // l == 0 -- and produces both magnitudes: half the permutes and half the arithmetic of doing
        // every bin on its own.  acc[m] = |X[l + T*m]|, acc[P/2 + m] = |X[NC - l - T*m]|; the slot that
        // would be bin NC (lane 0, m = 0) carries bin NC/2 (self-paired, lane 0 register P/2) instead.
        const int plane = ((lane & ~(T - 1)) | ((T - l) & (T - 1))) << 2;  // byte address for bpermute
        // keep the per-bin phasors utw*const from being hoisted out of the row loop (they would cost
        // resident registers or, worse, scratch reloads): utw is opaque from here
        v2f utw_row = utw;
        asm volatile("" : "+v"(utw_row));
        constexpr int PH = P / 2;
        v2f pz[PH];
        // every lane publishes the register its reader wants; all permutes are issued before any of
        // the arithmetic (one LDS round trip)
        static_for<0, PH>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          constexpr int pm1 = P - 1 - m;      // asked for by lane T-l' != 0
          constexpr int pm0 = (P - m) % P;    // asked for by lane 0 (of itself)
          const float sx = (l == 0) ? z[pm0].x : z[pm1].x;
          const float sy = (l == 0) ? z[pm0].y : z[pm1].y;
          pz[m] = mk(__int_as_float(__builtin_amdgcn_ds_bpermute(plane, __float_as_int(sx))),
                     __int_as_float(__builtin_amdgcn_ds_bpermute(plane, __float_as_int(sy))));
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, PH>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const v2f A_ = add_conj(z[m], pz[m]);
          const v2f B_ = sub_conj(z[m], pz[m]);
          // w^k = exp(2*pi*i*(l + T*m)/N) = utw * exp(2*pi*i*m/(2P))
          const v2f q = cmul(twc<m, 2 * P, true>(utw_row), B_);
          const v2f Xa = add_mulmi(A_, q);   // A - i*q
          const v2f Xb = sub_mulmi(A_, q);   // A + i*q
          const v2f Xa2 = Xa * Xa, Xb2 = Xb * Xb;
          const float lo_mag = fast_sqrt(Xa2.x + Xa2.y);
          acc[m] = AVG ? acc[m] + lo_mag : lo_mag;
          float hi = fast_sqrt(Xb2.x + Xb2.y);
          if constexpr (m == 0) {
            // bin NC/2: Z[NC/2] is its own partner; only lane 0 keeps the result
            const v2f zc = z[PH];
            const v2f Ac = add_conj(zc, zc), Bc = sub_conj(zc, zc);
            const v2f qc = cmul(twc<PH, 2 * P, true>(utw_row), Bc);
            const v2f Xc = add_mulmi(Ac, qc);
            const v2f Xc2 = Xc * Xc;
            const float mid = fast_sqrt(Xc2.x + Xc2.y);
            hi = (l == 0) ? mid : hi;
          }
          acc[PH + m] = AVG ? acc[PH + m] + hi : hi;
        });
      }
      }  // STAGE != 1
      FDOCT_FPR(pr, 8);   // untangle + magnitude (+ accumulate)
      // the prefetched samples have had this whole pass to arrive (see the comment at the first issue_loads)
      if constexpr (STAGE == 2) {
#pragma unroll
        for (int m = 0; m < P; m++) asm volatile("" : "+v"(znext[m]));
      } else {
        if constexpr (PF2) {
          // the NEXT row's samples (the other set), asked for a row and a half ago: younger than they are the write-out stores this
          // wave has issued since the last wait here and the loads of the row after next
          pf2_wait(pf2_stores + (unsigned)WCH);
          pf2_stores = 0u;
          pf2_pin(pf2_s0);   // (both sets, unconditionally: see pf2_load)
          pf2_pin(pf2_s1);
        } else {
#pragma unroll
          for (int c = 0; c < WCH; c++) raw[c].pin();
        }
        if constexpr (IB2D) {
#pragma unroll
          for (int i = 0; i < NPR; i++) asm volatile("" : "+v"(r_ib[i]));
          if constexpr (IL16 && !ILDMA) {
#pragma unroll
            for (int c = 0; c < WCH; c++) asm volatile("" : "+v"(r_il16[c].x), "+v"(r_il16[c].y), "+v"(r_il16[c].z), "+v"(r_il16[c].w));
          }
        }
      }
    }  // averaging loop
    if constexpr (STAGE == 1) {
      o_wave = o_next;
      continue;
    }

    if constexpr (TRO && !TRO_INPLACE) {
      // the ring slot of this row (ticket mod RS) last held the row of ticket - RS: its tile must have been written out
      const unsigned need = tro_cur.t >= RS ? (tro_cur.t - RS) / TR + 1u : 0u;
#if FDOCT_TRO_DW == 2
      // (the write-out this waits for is done by the wave that completes that tile, at once and without a wait of its own;
      // the bound is the exit condition a spinning wave must have all the same and is reported through a.tr_fault)
      for (unsigned spin = 0; tro_rel_seen < need; spin++) {
        tro_rel_seen = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&tr_released, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (tro_rel_seen >= need) break;
        if (spin >= FDOCT_TRO_SPIN_LIMIT) {
          if (lane == 0) __hip_atomic_store(a.tr_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
#elif FDOCT_TRO_DW == 1
      const unsigned need_steps = need * ((unsigned)a.D / (unsigned)TRO_SB);
      // (the write-out this waits for may be this wave's own to do; the bound is the exit condition a spinning wave must
      // have all the same and is reported through a.tr_fault)
      for (unsigned spin = 0; tro_done_seen < need_steps; spin++) {
        tro_done_seen = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&tr_wo_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (tro_done_seen >= need_steps) break;
        if (spin >= FDOCT_TRO_SPIN_LIMIT) {
          if (lane == 0) __hip_atomic_store(a.tr_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        if (!tro_writeout(1)) __builtin_amdgcn_s_sleep(2);
      }
#else
      if (need) {
        unsigned seen = (unsigned)__builtin_amdgcn_readfirstlane(
            (int)__hip_atomic_load(&tr_released, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        // (the write-out this waits for depends on no wait of its own; the bound is the exit condition a spinning wave must
        // have all the same and is reported through a.tr_fault)
        for (unsigned spin = 0; seen < need; spin++) {
          if (spin >= FDOCT_TRO_SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_store(a.tr_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          seen = (unsigned)__builtin_amdgcn_readfirstlane(
              (int)__hip_atomic_load(&tr_released, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        }
      }
#endif
    }
    // ---------------- A9/A10: average, epsilon, dB, DC mask, store
    const int D = a.D;
    const bool upper_slots = !CPLX || D > NC / 2;  // complex path, half-depth output: slots >= P/2 are never stored
    float outv[P];
#pragma unroll
    for (int m = 0; m < P / 2; m++) outv[m] = AVG ? fmaf(acc[m], a.inv_A, a.eps) : (acc[m] + a.eps);
    if (upper_slots) {
#pragma unroll
      for (int m = P / 2; m < P; m++) outv[m] = AVG ? fmaf(acc[m], a.inv_A, a.eps) : (acc[m] + a.eps);
    } else {
#pragma unroll
      for (int m = P / 2; m < P; m++) outv[m] = 1.f;
    }
    // slot -> depth bin.  Complex path: slot m is bin l + T*m.  Real path (see the untangle above):
    // slots m < P/2 are bins l + T*m, slots P/2 + m are bins NC - l - T*m (lane 0, m = 0: bin NC/2).
    // Stores are <per-lane base pointer> + <immediate>: lo slots ascend from orow + l, hi slots
    // descend from orow + NC - l (a wave still writes 64 consecutive floats per instruction).
    // Non-temporal stores: the output is a stream nobody on this GPU reads back soon (+0.8 % on C2).
    // BUFFER stores: a wave's RPW rows are consecutive rows of the output, so one descriptor (four SGPRs) based at the
    // wave's first row covers them and the per-lane part is one 32-bit byte offset, where a global store carries a
    // 64-bit VGPR address (+1.6 % on C2: fewer address registers read per store).  num_records = the floats of the
    // wave's valid rows: the hardware drops the rows past the end of the batch and, with one row per wave, whatever lies
    // past the crop.
    float* const stg_row = stg;  // (FDOCT_X_LDS_STORE) the wave's own LDS buffer, free between the untangle and the next row's staging
    auto store_row = [&](float* obase, const float* val) {
      constexpr int NLO = CPLX ? P : P / 2;
#ifdef FDOCT_X_PLAIN_STORE  // tuning: ordinary (write-back) global stores
      constexpr bool BUF = false;
      auto stg = [](float* p, float v) { *p = v; };
#elif defined(FDOCT_X_GLOBAL_STORE)  // tuning: non-temporal global stores
      constexpr bool BUF = false;
      auto stg = [](float* p, float v) { __builtin_nontemporal_store(v, p); };
#else
#ifdef FDOCT_X_LDS_STORE  // tuning experiment: the row goes through the wave's own (free) LDS buffer and leaves as 16-byte stores
      constexpr bool LSX = LEAN && KIND == 1 && !TRO && RPW == 1;
#else
      constexpr bool LSX = false;
#endif
      constexpr bool BUF = !TRO && !LSX;
      auto stg = [](float* p, float v) { __builtin_nontemporal_store(v, p); };
#endif
      // o_wave is wave-uniform by construction (slot_row of a wave-uniform ticket); say so for RPW > 1 too, or the
      // descriptor would be built per lane and the stores wrapped in a waterfall loop
      const long long ow = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(o_wave >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)o_wave));
      const long long left = total - ow;  // >= 1 inside the row loop
      int nrows = left < RPW ? (int)left : RPW;
      float* wbase = obase + (size_t)ow * D;  // the wave's first row
      float* orow = wbase + ((RPW > 1) ? (size_t)sub * D : 0);
      // this row's ring slot (LDS: the stores below are ds_write_b32); the RPW rows of a wave take consecutive slots (the ring
      // is a whole number of them: fused_tro_ring_pick)
      if constexpr (TRO_INPLACE)
        orow = reinterpret_cast<float*>(scratch0 + (size_t)(wave * RPW + sub) * a.scratch_bytes);   // the row's own buffer, free since the untangle
      else if constexpr (TRO)
        orow = tro_ring + (ring_mod(tro_cur.t) + (unsigned)sub) * tro_slot;
      float* const grow = orow;  // (LSX) where the row goes in global memory
      if constexpr (LSX) orow = stg_row;
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(BUF ? wbase : nullptr, 0, BUF ? nrows * D * 4 : 0, 0x00020000);
      // bin index -> store; lo(m) = bin l + T*m, hi(m) = bin NC - l - T*m, hi0 = slot P/2 (lane 0: bin NC/2)
      float* plo = orow + l;
      float* phi = orow + (NC - l);
      // per-lane byte offsets of the ascending and of the descending run; opaque to the optimiser so that "+ constant"
      // stays an add the backend folds into the instruction's immediate offset (it does not fold the `or` it would become)
      const int vrow = (RPW > 1) ? 4 * sub * D : 0;
      int vlo = vrow + 4 * l, vhi = vrow + 4 * (NC - T * (P / 2 - 1) - l);
      if constexpr (BUF) asm volatile("" : "+v"(vlo), "+v"(vhi));
      auto st_lo = [&](int m, float v) {
        if constexpr (BUF)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vlo + 4 * T * m, 0, 2);  // aux 2 = nt
        else if constexpr (TRO || LSX)
          plo[T * m] = v;
        else
          stg(plo + T * m, v);
      };
      auto st_hi = [&](int m, float v) {
        if constexpr (BUF)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vhi + 4 * T * (P / 2 - 1 - m), 0, 2);
        else if constexpr (TRO || LSX)
          phi[-T * m] = v;
        else
          stg(phi - T * m, v);
      };
      auto st_hi0 = [&](float v) {  // slot P/2 of lane 0 is bin NC/2
        if constexpr (BUF)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vrow + 4 * ((l == 0) ? NC / 2 : NC - l), 0, 2);
        else if constexpr (TRO || LSX)
          *((l == 0) ? orow + NC / 2 : phi) = v;
        else
          stg((l == 0) ? orow + NC / 2 : phi, v);
      };
      if (D == NC || (BUF && RPW == 1 && (D % T) != 0)) {  // full depth: nothing to crop; ragged crop of a wave-per-row plan: the hardware drops bins >= D
#pragma unroll
        for (int m = 0; m < NLO; m++) st_lo(m, val[m]);
        if constexpr (!CPLX) {
          st_hi0(val[NLO]);
#pragma unroll
          for (int m = 1; m < P / 2; m++) st_hi(m, val[NLO + m]);
        }
      } else if ((D % T) == 0) {
        // cropped to whole T-bin slots (the usual half-depth display): which slots are stored is the same for every
        // lane, so the tests are scalar branches -- no per-store exec masking (and none of its SGPR pressure)
        const int nfull = D / T;
#pragma unroll
        for (int m = 0; m < NLO; m++)
          if (m < nfull) st_lo(m, val[m]);
        if constexpr (!CPLX) {
          const int h = (NC - D) / T;  // >= 1: hi slot m holds bins NC - T*m - l, all below D iff m > h
          if (((l == 0) ? NC / 2 : NC - l) < D) st_hi0(val[NLO]);
#pragma unroll
          for (int m = 1; m < P / 2; m++) {
            if (m > h) {
              st_hi(m, val[NLO + m]);
            } else if (m == h) {
              if (l != 0) st_hi(m, val[NLO + m]);  // lane 0's bin is D itself
            }
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < NLO; m++)
          if (l + T * m < D) st_lo(m, val[m]);
        if constexpr (!CPLX) {
          if (((l == 0) ? NC / 2 : NC - l) < D) st_hi0(val[NLO]);
#pragma unroll
          for (int m = 1; m < P / 2; m++)
            if (NC - l - T * m < D) st_hi(m, val[NLO + m]);
        }
      }
      if constexpr (LSX) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        wave_lds_sync();
        const int nq = D >> 8;  // 256 bins per instruction (D a multiple of 256 here: host)
        for (int k = 0; k < nq; k++) {
          const f4 v = *reinterpret_cast<const f4*>(orow + 4 * l + 256 * k);
          __builtin_nontemporal_store(v, reinterpret_cast<f4*>(grow + 4 * l + 256 * k));
        }
        asm volatile("s_nop 1");
        wave_lds_sync();
      }
    };
    const bool deposit = valid && !(TRO_INPLACE && tro_cur.dummy);   // (a wave without rows in a short tile stores nothing)
    if (deposit && a.out_mag && !FDOCT_ABL(128)) store_row(a.out_mag, outv);
    // (transposed store with both images asked for: the ring holds bscan, the write-out wave takes the logarithm)
    if (a.out_db && !(TRO && a.out_mag)) {
      float db[P];
#pragma unroll
      for (int m = 0; m < P / 2; m++) db[m] = FDOCT_ABL(64) ? outv[m] : a.db_scale * fast_log2(outv[m]);  // db_scale carries ln 2
      if (upper_slots) {
#pragma unroll
        for (int m = P / 2; m < P; m++) db[m] = FDOCT_ABL(64) ? outv[m] : a.db_scale * fast_log2(outv[m]);
      } else {
#pragma unroll
        for (int m = P / 2; m < P; m++) db[m] = 0.f;
      }
      if (a.dcmask && T > 4 && D > 4) {  // (the reference indexes row 4 unconditionally; with D <= 4 there is none: no mask)
        // depth bins 0 and 1 <- bin 4 (main:1237-1238): bins 0, 1, 4 are slot 0 of lanes 0, 1, 4
        const float d4 = __shfl(db[0], (lane & ~(T - 1)) | 4, 64);
        if (l < 2) db[0] = d4;
      }
      if (deposit && !FDOCT_ABL(128)) store_row(a.out_db, db);
      if (FDOCT_ABL(128)) {  // keep the values alive: without this the whole row would be dead code
#pragma unroll
        for (int m = 0; m < P; m++) asm volatile("" ::"v"(db[m]));
      }
    }
    o_wave = o_next;
    FDOCT_FPR(pr, 9);     // epilogue: average, epsilon, dB, stores (+ the transposed store's ring wait)
    if constexpr (TRO_INPLACE) {
      // the wave's rows lie in its own row buffers: meet the group, write the tile out together (wave m takes steps m, m + 4, ...:
      // each step reads all sixteen rows, four of them from every wave's buffers), meet again -- nobody may start staging its
      // next rows before every wave of the group has read what it needs -- and go on
      wave_lds_sync();
      const unsigned target = GW * (grp_seq + 1u);
      grp_meet(&grp_arrived[grp], target);
      asm volatile("" ::: "memory");
      const unsigned spt = (unsigned)a.D / (unsigned)TRO_SB;
      for (unsigned k = mem; k < spt; k += GW) tro_step(grp, tro_cur.g, tro_cur.r0, tro_cur.nrows, (int)(k * (unsigned)TRO_SB));
      asm volatile("" ::: "memory");   // the steps' LDS reads have returned (they fed stores that have been issued)
      wave_lds_sync();
      grp_meet(&grp_done[grp], target);
      grp_seq++;
      tro_cur = tro_next;
    } else if constexpr (TRO) {
      // the row is in the ring (a wave's LDS operations execute in order): count it for the write-out wave
      wave_lds_sync();
#if FDOCT_TRO_DW == 2
      unsigned cnt = 0u;
      if (lane == 0) cnt = __hip_atomic_fetch_add(&tr_arrived[tro_cur.tq & 3u], (unsigned)RPW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)cnt) + (unsigned)RPW == tro_cur.nrows) tro_tile_out(tro_cur.tq, tro_cur.g, tro_cur.r0, tro_cur.nrows);
#elif FDOCT_TRO_DW == 1
      // count the row, and read the write-out state in the same LDS round trip
      unsigned cnt = 0u;
      if (lane == 0) cnt = __hip_atomic_fetch_add(&tr_arrived[tro_cur.tq & 3u], (unsigned)RPW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned s_l = __hip_atomic_load(&tr_wo_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned r_l = __hip_atomic_load(&tr_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      unsigned ready = (unsigned)__builtin_amdgcn_readfirstlane((int)r_l);
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)cnt) + (unsigned)RPW == tro_cur.nrows)  // the tile is complete: publish it
        ready = tro_publish(tro_cur.tq, &tr_ready);
      if (tro_try_step((unsigned)__builtin_amdgcn_readfirstlane((int)s_l), ready) && FDOCT_TRO_DW_STEPS > 1) (void)tro_writeout(FDOCT_TRO_DW_STEPS - 1);
#else
      if (lane == 0) __hip_atomic_fetch_add(&tr_arrived[tro_cur.tq & 3u], (unsigned)RPW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
      tro_cur = tro_next;
      if constexpr (PF2) {
        tro_next = tro_n2;
        pf2_o1 = pf2_o2;
        pf2_ph = !pf2_ph;
      }
    }
  }
#if FDOCT_TRO_DW == 1
  if constexpr (TRO && !TRO_INPLACE) {
    // all rows of this wave are done: help until the workgroup's last tile is out (it completes when its last row is in the
    // ring, which needs no help from here; the bound is the exit condition a spinning wave must have all the same)
    const unsigned nx_ = gridDim.x >> 3;
    const unsigned bp_ = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * nx_ + (blockIdx.x >> 3);
    const unsigned my_tiles = bp_ < a.tr_total_tiles ? (a.tr_total_tiles - bp_ + gridDim.x - 1u) / gridDim.x : 0u;
    for (unsigned spin = 0;; spin++) {
      const unsigned outn = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&tr_wo_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (outn >= my_tiles * ((unsigned)a.D / (unsigned)TRO_SB)) break;
      if (tro_writeout(4)) {
        spin = 0;
        continue;
      }
      if (spin >= FDOCT_TRO_SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store(a.tr_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
#endif
#ifdef FDOCT_FUSED_PROBE
  if (a.phase_probe && lane == 0 && blockIdx.x < 4 && wave < 16) {
    for (int i = 0; i < FUSED_PROBE_PHASES; i++) a.phase_probe[(blockIdx.x * 16 + wave) * FUSED_PROBE_PHASES + i] = pr.acc[i];
  }
#endif
#ifdef FDOCT_CLOCKPROBE
  if (a.probe && blockIdx.x == 0 && lane == 0 && wave < 16) {
    a.probe[2 * wave] = __builtin_readcyclecounter() - probe_c0;
    a.probe[2 * wave + 1] = wall_clock64() - probe_r0;
  }
  if (a.probe && tid == 0 && blockIdx.x < 512) {  // per-workgroup start / end on the chip-wide 100 MHz clock
    a.probe[32 + 2 * blockIdx.x] = probe_r0;
    a.probe[32 + 2 * blockIdx.x + 1] = wall_clock64();
  }
#endif
}

#ifndef FDOCT_ONLY_PLAN  // (per-plan translation units hold the fused kernels only)
// ---------------------------------------------------------- small kernels --
// Whole-frame min/max (main:1128-1129) of the raw samples, one float2 per frame.
template <typename IN_T>
__global__ void minmax_kernel(const void* frames, long long pitch_bytes, int W, int H, const float* yd,
                              int yd_2d, float2* out) {
  const int f = blockIdx.x;
  const unsigned char* base = static_cast<const unsigned char*>(frames) + (long long)f * H * pitch_bytes;
  float mn = INFINITY, mx = -INFINITY;
  for (int r = 0; r < H; r++) {
    const IN_T* row = reinterpret_cast<const IN_T*>(base + (long long)r * pitch_bytes);
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
      float x = (float)row[i];
      if (yd) x -= yd[(yd_2d ? (size_t)r * W : 0) + i];
      mn = fminf(mn, x);
      mx = fmaxf(mx, x);
    }
  }
  __shared__ float smn[16], smx[16];
  mn = group_min<64>(mn);
  mx = group_max<64>(mx);
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
      mn = fminf(mn, smn[w]);
      mx = fmaxf(mx, smx[w]);
    }
    out[f] = make_float2(mn, mx);
  }
}

// The same for plain integer camera frames (no dark frame, 16-byte-aligned rows): a streaming reduction with
// 16-byte loads and packed 16-bit min/max, MINMAX_PARTS workgroups per frame, then one thread per frame folds the
// partial results.  HBM-bound: one read of the batch.
constexpr int MINMAX_PARTS = 16;
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

template <typename IN_T>
__global__ __launch_bounds__(256) void minmax_fast_kernel(const void* frames, long long pitch_bytes, int row_vecs, int H,
                                                           float2* partial) {
  const int f = blockIdx.y;
  const unsigned char* base = static_cast<const unsigned char*>(frames) + (long long)f * H * pitch_bytes;
  us2 mn = (us2){0xffff, 0xffff}, mx = (us2){0, 0};
  auto fold = [&](unsigned w) {
    if constexpr (sizeof(IN_T) == 2) {
      const us2 x = __builtin_bit_cast(us2, w);
      mn = __builtin_elementwise_min(mn, x);
      mx = __builtin_elementwise_max(mx, x);
    } else {  // four bytes: even and odd ones as two 16-bit pairs
      const us2 x0 = __builtin_bit_cast(us2, w & 0x00ff00ffu), x1 = __builtin_bit_cast(us2, (w >> 8) & 0x00ff00ffu);
      mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(x0, x1));
      mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(x0, x1));
    }
  };
  for (int r = blockIdx.x; r < H; r += gridDim.x) {
    const uint4* row = reinterpret_cast<const uint4*>(base + (long long)r * pitch_bytes);
    for (int v = threadIdx.x; v < row_vecs; v += blockDim.x) {
      const uint4 q = row[v];
      fold(q.x);
      fold(q.y);
      fold(q.z);
      fold(q.w);
    }
  }
  unsigned lo = mn.x < mn.y ? mn.x : mn.y, hi = mx.x > mx.y ? mx.x : mx.y;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned l2 = __shfl_xor(lo, m, 64), h2 = __shfl_xor(hi, m, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  __shared__ unsigned slo[4], shi[4];
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      lo = slo[w] < lo ? slo[w] : lo;
      hi = shi[w] > hi ? shi[w] : hi;
    }
    partial[(size_t)f * gridDim.x + blockIdx.x] = make_float2((float)lo, (float)hi);
  }
}

__global__ void minmax_finish_kernel(const float2* partial, int parts, int nframes, float2* out) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nframes) return;
  float2 r = partial[(size_t)f * parts];
  for (int i = 1; i < parts; i++) {
    const float2 p = partial[(size_t)f * parts + i];
    r.x = fminf(r.x, p.x);
    r.y = fmaxf(r.y, p.y);
  }
  out[f] = r;
}

// (rows x cols) -> (cols x rows) per group, through a 32x33 LDS tile.
__global__ void transpose_kernel(const float* in, float* out, int rows, int cols) {
  __shared__ float tile[32][33];
  const size_t goff = (size_t)blockIdx.z * rows * cols;
  int x = blockIdx.x * 32 + threadIdx.x;
  int y0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    int y = y0 + j;
    if (x < cols && y < rows) tile[j][threadIdx.x] = in[goff + (size_t)y * cols + x];
  }
  __syncthreads();
  int ox = blockIdx.y * 32 + threadIdx.x;  // row index of input
  int oy0 = blockIdx.x * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    int oy = oy0 + j;  // col index of input
    if (ox < rows && oy < cols) out[goff + (size_t)oy * rows + ox] = tile[threadIdx.x][j];
  }
}

// The same for rows and cols that are multiples of 4 (every B-scan of the bench and of the shipped configurations): a
// 64 x 64 tile, 16 bytes per lane on both sides -- a wave reads 4 rows x 256 bytes and writes 4 output rows x 256 bytes,
// non-temporal (each byte is touched once).  Tile rows are padded to 65 floats: the column reads of the second phase
// (rows 4 r4 + i of column c) land two lanes per bank.
__global__ void __launch_bounds__(256) transpose64_kernel(const float* in, float* out, int rows, int cols) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  __shared__ float tile[64][65];
  const size_t goff = (size_t)blockIdx.z * rows * cols;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;  // tile origin: column, row of the input
  const int q = threadIdx.x & 15, p = threadIdx.x >> 4;  // 16 lanes x 16 bytes across, 16 lines down per pass
#pragma unroll
  for (int pass = 0; pass < 4; pass++) {
    const int y = y0 + p + 16 * pass, x = x0 + 4 * q;
    if (y < rows && x < cols) {
      const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(in + goff + (size_t)y * cols + x));
      float* t = &tile[p + 16 * pass][4 * q];
      t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 4; pass++) {
    const int c = p + 16 * pass;           // input column = output row
    const int oy = x0 + c, ox = y0 + 4 * q;  // output row, first of four output columns (= input rows)
    if (oy < cols && ox < rows) {
      const f4v v = {tile[4 * q][c], tile[4 * q + 1][c], tile[4 * q + 2][c], tile[4 * q + 3][c]};
      __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(out + goff + (size_t)oy * rows + ox));
    }
  }
}

// data_y as the reference holds it, CV_64F (main:987, 1125): split once into two f32 planes, x = hi + lo to 2^-48 of x.  The
// kernels that take f32 frames carry lo into the division by the background next to hi (FusedArgs::frames_lo), so that a
// frame of non-integer samples is not rounded at the size of the DC level (camera frames are integers: lo is 0 there).
__global__ void f64_split_kernel(const double* in, long long pitch_elems, float* hi, float* lo, int W, long long rows) {
  const long long n = rows * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / W;
    const int c = (int)(i - r * W);
    const double x = in[r * pitch_elems + c];
    const float h = (float)x;
    hi[i] = h;
    const double l = x - (double)h;
    lo[i] = (l == l && h - h == 0.f) ? (float)l : 0.f;   // (inf / nan: no low word)
  }
}

#endif  // !FDOCT_ONLY_PLAN

// ---------------------------------------------------------------- dispatch --
template <typename X>
struct TypeTag {
  using type = X;
};

template <int LOG2NC, int T, int R1, int R2, int R3, int KIND, int WCH, typename IN_T, bool CPLX, bool LEAN, int STAGE = 0, bool AVG = true,
          bool IB2D = false, int NORM = 0, bool TRO = false, bool PRECT = false>
static hipError_t launch_one(const FusedArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  auto k = fused_kernel<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, LEAN, STAGE, AVG, IB2D, NORM, TRO, PRECT>;
  static LdsGrant grant;  // one per instantiation
  if (hipError_t e = grant.ensure(k, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(k, grid, block, lds, st, a);
  return hipGetLastError();
}

// PRECT: the fast-path instantiations that multiply by both words of the reciprocal background (a.prec != 0 on a lean
// launch); they live in translation units of their own (launch_plan<ID, true>).  The any-option kernel (always both words)
// and the one-word fast path are launch_plan<ID, false>.
template <int LOG2NC, int T, int R1, int R2, int R3, int KIND, int WCH, bool CPLX, bool PRECT>
static hipError_t launch_typed(const FusedArgs& a, int dtype, bool lean, dim3 grid, dim3 block, size_t lds,
                               hipStream_t st) {
  if (PRECT != (lean && a.prec != 0)) return hipErrorInvalidValue;  // (launch_fused picks the translation unit)
#ifdef FDOCT_DEV_ONE  // tuning builds (tools/mkvariant.sh): ONE instantiation -- plain u16 fast path, fused, no averaging
  if constexpr (FDOCT_DEV_ONE_CPLX != CPLX) return hipErrorNotSupported;
  else {
    if (!lean || dtype != FDOCT_K_U16 || a.stage != 0 || a.ib2d || a.minmax || a.rowwisenormalize) return hipErrorNotSupported;
    if ((a.A == 1) != (FDOCT_DEV_ONE_AVG == 0)) return hipErrorNotSupported;
    if constexpr (fused_tro_compiled(KIND, T, WCH) && !CPLX) {
      if (a.tro) return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, uint16_t, CPLX, true, 0, FDOCT_DEV_ONE_AVG != 0, false, 0, true, PRECT>(a, grid, block, lds, st);
    } else if (a.tro) {
      return hipErrorNotSupported;
    }
    return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, uint16_t, CPLX, true, 0, FDOCT_DEV_ONE_AVG != 0, false, 0, false, PRECT>(a, grid, block, lds, st);
  }
#else
  if (a.stage != 0) {  // staged mode: built for the fast-path configuration only
    if (!lean || dtype != FDOCT_K_U16) return hipErrorNotSupported;  // (capi checks this before launching)
    // the resample stage always runs over INPUT A-scans (capi hands it A = 1); the FFT stage averages (and reads no samples:
    // one instantiation serves both modes)
    if (a.stage == 1)
      return a.A == 1 ? launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, uint16_t, CPLX, true, 1, false, false, 0, false, PRECT>(a, grid, block, lds, st)
                      : hipErrorNotSupported;
    return a.A == 1 ? launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, uint16_t, CPLX, true, 2, false, false, 0, false, PRECT>(a, grid, block, lds, st)
                    : launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, uint16_t, CPLX, true, 2, true, false, 0, false, PRECT>(a, grid, block, lds, st);
  }
  if (a.tro) {  // transposed output written by the chain itself (capi checks the conditions before asking for it)
    if constexpr (fused_tro_compiled(KIND, T, WCH) && !CPLX) {
      if (!lean || a.rowwisenormalize || !a.tr_fault) return hipErrorNotSupported;
      // the plain acquisition configuration with any averaging; one frame per B-scan also with a full-frame background and /
      // or the whole-frame normalisation (what the 'b' key stores, what BscanFFTsim.cpp always does: sim:803-813, 845)
      auto tro = [&](auto in_c) {
        using IN_T = typename decltype(in_c)::type;
        if (a.ib2d || a.minmax) {
          if (a.A != 1) return hipErrorNotSupported;
          if constexpr (KIND != 1) {   // (the 512-point Stockham plan: plain and averaging only; the host keeps the rest on the two-pass route)
            return hipErrorNotSupported;
          } else {
            if constexpr (!PRECT || fused_il_half(true, WCH)) {  // (both words with a full-frame background: the half-float form only)
              if (a.ib2d && a.minmax) return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, false, true, 1, true, PRECT>(a, grid, block, lds, st);
              if (a.ib2d) return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, false, true, 0, true, PRECT>(a, grid, block, lds, st);
            } else if (a.ib2d) {
              return hipErrorNotSupported;
            }
            return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, false, false, 1, true, PRECT>(a, grid, block, lds, st);
          }
        }
        return a.A == 1 ? launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, false, false, 0, true, PRECT>(a, grid, block, lds, st)
                        : launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, true, false, 0, true, PRECT>(a, grid, block, lds, st);
      };
      if (dtype == FDOCT_K_U16) return tro(TypeTag<uint16_t>{});
      if (dtype == FDOCT_K_U8) return tro(TypeTag<uint8_t>{});
    }
    return hipErrorNotSupported;
  }
  if constexpr ((KIND == 1 || KIND == 2) && WCH <= 4) {
    // fast path with a full-frame background (capi hands the evens/odds-ordered copy) and / or a normalisation
    if (lean && (a.ib2d || a.minmax || a.rowwisenormalize)) {
      auto opts = [&](auto in_c, auto avg_c) {
        using IN_T = typename decltype(in_c)::type;
        constexpr bool AVG = decltype(avg_c)::value;
        if constexpr (!fused_resident_consts(KIND, true, AVG, WCH, 0)) {
          return hipErrorNotSupported;  // (capi keeps such configurations on the general kernel)
        } else {
        const int norm = a.rowwisenormalize ? 2 : (a.minmax ? 1 : 0);
        if (a.ib2d) {
          if constexpr (!PRECT || fused_il_half(true, WCH)) {
            if (norm == 2) return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, AVG, true, 2, false, PRECT>(a, grid, block, lds, st);
            if (norm == 1) return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, AVG, true, 1, false, PRECT>(a, grid, block, lds, st);
            return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, AVG, true, 0, false, PRECT>(a, grid, block, lds, st);
          } else {
            return hipErrorNotSupported;
          }
        }
        if (norm == 2) return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, AVG, false, 2, false, PRECT>(a, grid, block, lds, st);
        return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, AVG, false, 1, false, PRECT>(a, grid, block, lds, st);
        }
      };
      if (dtype == FDOCT_K_U16)
        return a.A == 1 ? opts(TypeTag<uint16_t>{}, std::false_type{}) : opts(TypeTag<uint16_t>{}, std::true_type{});
      if (dtype == FDOCT_K_U8)
        return a.A == 1 ? opts(TypeTag<uint8_t>{}, std::false_type{}) : opts(TypeTag<uint8_t>{}, std::true_type{});
      return hipErrorInvalidValue;
    }
  }
  auto plain = [&](auto in_c) {
    using IN_T = typename decltype(in_c)::type;
    if (lean)
      return a.A == 1 ? launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, false, false, 0, false, PRECT>(a, grid, block, lds, st)
                      : launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, true, 0, true, false, 0, false, PRECT>(a, grid, block, lds, st);
    if constexpr (!PRECT)
      return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, IN_T, CPLX, false>(a, grid, block, lds, st);
    else
      return hipErrorInvalidValue;
  };
  switch (dtype) {
    case FDOCT_K_U16: return plain(TypeTag<uint16_t>{});
    case FDOCT_K_U8: return plain(TypeTag<uint8_t>{});  // 8-bit cameras (the shipped ini's default) get the fast path too
    case FDOCT_K_F32:
      if constexpr (!PRECT)
        return launch_one<LOG2NC, T, R1, R2, R3, KIND, WCH, float, CPLX, false>(a, grid, block, lds, st);
      else
        return hipErrorInvalidValue;
    default:
      return hipErrorInvalidValue;
  }
#endif  // FDOCT_DEV_ONE
}

// The table of compiled plans: {id, log2 nc, T, R1, R2, R3, kind, WCH}.  nc = complex FFT length;
// kind 0 = Stockham passes through LDS, kind 1 = fft1024_rowswap, kind 2 = fft2048_rowswap.
#ifdef FDOCT_DEV_ONE  // fastest compile while tuning: one instantiation of one plan (-DFDOCT_DEV_ONE=<plan id>)
#define FDOCT_DEV_SINGLE
#define FDOCT_PLAN_1(X) X(1, 9, 16, 32, 16, 1, 0, 8)
#define FDOCT_PLAN_5(X) X(5, 10, 64, 16, 4, 16, 1, 4)
#define FDOCT_PLAN_7(X) X(7, 11, 64, 32, 4, 16, 2, 4)
#define FDOCT_PLAN_8(X) X(8, 11, 64, 32, 4, 16, 2, 8)
#define FDOCT_CAT_(a, b) a##b
#define FDOCT_CAT(a, b) FDOCT_CAT_(a, b)
#define FDOCT_PLANS(X) FDOCT_CAT(FDOCT_PLAN_, FDOCT_DEV_ONE)(X)
#elif defined(FDOCT_DEV_SINGLE)  // fast compile while tuning: only the benchmark plan
#define FDOCT_PLANS(X) X(5, 10, 64, 16, 4, 16, 1, 4)
#else
#define FDOCT_PLANS(X)               \
  X(0, 8, 16, 16, 16, 1, 0, 4)       \
  X(1, 9, 16, 32, 16, 1, 0, 8)       \
  X(2, 10, 64, 16, 16, 4, 0, 4)      \
  X(3, 10, 32, 32, 32, 1, 0, 8)      \
  X(4, 11, 64, 32, 8, 8, 0, 8)       \
  X(5, 10, 64, 16, 4, 16, 1, 4)      \
  X(6, 11, 64, 32, 8, 8, 0, 4)       \
  X(7, 11, 64, 32, 4, 16, 2, 4)      \
  X(8, 11, 64, 32, 4, 16, 2, 8)
#endif

template <int ID>
struct PlanOf;
#define FDOCT_PLANOF(ID, L2_, T_, R1_, R2_, R3_, K_, WCH_)                                      \
  template <>                                                                                   \
  struct PlanOf<ID> {                                                                           \
    static constexpr int L2 = L2_, T = T_, R1 = R1_, R2 = R2_, R3 = R3_, K = K_, WCH = WCH_;    \
  };
FDOCT_PLANS(FDOCT_PLANOF)
#undef FDOCT_PLANOF

// Every instantiation of one plan.  The build compiles this file once per plan (-DFDOCT_ONLY_PLAN=<id>: that
// translation unit instantiates launch_plan<id> and nothing else) plus once for everything else, in parallel.
template <int ID, bool PRECT>
hipError_t launch_plan(const FusedArgs& a, int dtype, bool cplx, bool lean, dim3 g, dim3 b, size_t lds, hipStream_t st) {
  using P = PlanOf<ID>;
  return cplx ? launch_typed<P::L2, P::T, P::R1, P::R2, P::R3, P::K, P::WCH, true, PRECT>(a, dtype, lean, g, b, lds, st)
              : launch_typed<P::L2, P::T, P::R1, P::R2, P::R3, P::K, P::WCH, false, PRECT>(a, dtype, lean, g, b, lds, st);
}

#ifdef FDOCT_ONLY_PLAN
#ifndef FDOCT_ONLY_PRECT
#define FDOCT_ONLY_PRECT 0
#endif
template hipError_t launch_plan<FDOCT_ONLY_PLAN, FDOCT_ONLY_PRECT != 0>(const FusedArgs&, int, bool, bool, dim3, dim3, size_t, hipStream_t);
#else
#ifndef FDOCT_DEV_SINGLE
#define FDOCT_EXTERN(ID, L2, T_, R1_, R2_, R3_, K_, WCH_)                                                                    \
  extern template hipError_t launch_plan<ID, false>(const FusedArgs&, int, bool, bool, dim3, dim3, size_t, hipStream_t); \
  extern template hipError_t launch_plan<ID, true>(const FusedArgs&, int, bool, bool, dim3, dim3, size_t, hipStream_t);
FDOCT_PLANS(FDOCT_EXTERN)
#undef FDOCT_EXTERN
#endif

int fused_plan_count() {
  int n = 0;
#define FDOCT_COUNT(ID, L2, T_, R1_, R2_, R3_, K_, WCH_) n++;
  FDOCT_PLANS(FDOCT_COUNT)
#undef FDOCT_COUNT
  return n;
}

bool fused_plan_get(int id, FusedPlan* p) {
#define FDOCT_GET(ID, L2, T_, R1_, R2_, R3_, K_, WCH_)        \
  if (id == ID) {                                             \
    *p = FusedPlan{ID, 1 << L2, T_, R1_, R2_, R3_, WCH_, K_}; \
    return true;                                              \
  }
  FDOCT_PLANS(FDOCT_GET)
#undef FDOCT_GET
  return false;
}

hipError_t launch_fused(const FusedPlan& p, const FusedArgs& a, int dtype, bool cplx, bool lean, int grid, int block,
                        size_t lds, hipStream_t st) {
  dim3 g(grid), b(block);
#define FDOCT_CASE(ID, L2, T_, R1_, R2_, R3_, K_, WCH_) \
  if (p.id == ID) return (lean && a.prec) ? launch_plan<ID, true>(a, dtype, cplx, lean, g, b, lds, st) : launch_plan<ID, false>(a, dtype, cplx, lean, g, b, lds, st);
  FDOCT_PLANS(FDOCT_CASE)
#undef FDOCT_CASE
  return hipErrorInvalidValue;
}

int minmax_partial_count(int nframes) { return nframes * MINMAX_PARTS; }

hipError_t launch_minmax(const void* frames, int dtype, long long pitch_bytes, int W, int H, int nframes,
                         const float* yd, int yd_2d, float2* out, float2* partial, hipStream_t st) {
  const int es = dtype == FDOCT_K_U16 ? 2 : 1;
  if (!yd && partial && (dtype == FDOCT_K_U16 || dtype == FDOCT_K_U8) && (W * es) % 16 == 0 && pitch_bytes % 16 == 0 &&
      (reinterpret_cast<uintptr_t>(frames) % 16) == 0) {
    // frames ride in gridDim.y (<= 65535 per launch): longer batches go in slices
    for (int f0 = 0; f0 < nframes; f0 += 65535) {
      const int nf = nframes - f0 < 65535 ? nframes - f0 : 65535;
      const dim3 g(MINMAX_PARTS, nf), b(256);
      const void* fr = static_cast<const unsigned char*>(frames) + (long long)f0 * H * pitch_bytes;
      float2* part = partial + (size_t)f0 * MINMAX_PARTS;
      if (dtype == FDOCT_K_U16)
        hipLaunchKernelGGL(minmax_fast_kernel<uint16_t>, g, b, 0, st, fr, pitch_bytes, W * es / 16, H, part);
      else
        hipLaunchKernelGGL(minmax_fast_kernel<uint8_t>, g, b, 0, st, fr, pitch_bytes, W * es / 16, H, part);
    }
    hipLaunchKernelGGL(minmax_finish_kernel, dim3((nframes + 255) / 256), dim3(256), 0, st, partial, MINMAX_PARTS, nframes, out);
    return hipGetLastError();
  }
  dim3 g(nframes), b(1024);
  switch (dtype) {
    case FDOCT_K_U16: hipLaunchKernelGGL(minmax_kernel<uint16_t>, g, b, 0, st, frames, pitch_bytes, W, H, yd, yd_2d, out); break;
    case FDOCT_K_U8: hipLaunchKernelGGL(minmax_kernel<uint8_t>, g, b, 0, st, frames, pitch_bytes, W, H, yd, yd_2d, out); break;
    case FDOCT_K_F32: hipLaunchKernelGGL(minmax_kernel<float>, g, b, 0, st, frames, pitch_bytes, W, H, yd, yd_2d, out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_transpose(const float* in, float* out, int rows, int cols, int groups, hipStream_t st) {
  // groups ride in gridDim.z and row tiles in gridDim.y (<= 65535 each per launch): more go in slices
  const int ytiles = (rows + 31) / 32;
  if (ytiles > 65535) return hipErrorInvalidValue;  // > 2 M rows per B-scan
  const bool wide = rows % 4 == 0 && cols % 4 == 0 && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
  for (int g0 = 0; wide && g0 < groups; g0 += 65535) {
    const int ng = groups - g0 < 65535 ? groups - g0 : 65535;
    const size_t off = (size_t)g0 * rows * cols;
    hipLaunchKernelGGL(transpose64_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, ng), dim3(256), 0, st, in + off, out + off, rows, cols);
  }
  if (wide) return hipGetLastError();
  for (int g0 = 0; g0 < groups; g0 += 65535) {
    const int ng = groups - g0 < 65535 ? groups - g0 : 65535;
    const size_t off = (size_t)g0 * rows * cols;
    dim3 b(32, 8), g((cols + 31) / 32, ytiles, ng);
    hipLaunchKernelGGL(transpose_kernel, g, b, 0, st, in + off, out + off, rows, cols);
  }
  return hipGetLastError();
}

hipError_t launch_f64_split(const double* in, long long pitch_elems, float* hi, float* lo, int W, long long rows, hipStream_t st) {
  hipLaunchKernelGGL(f64_split_kernel, dim3(2048), dim3(256), 0, st, in, pitch_elems, hi, lo, W, rows);
  return hipGetLastError();
}

#endif  // !FDOCT_ONLY_PLAN

}  // namespace fdoct
