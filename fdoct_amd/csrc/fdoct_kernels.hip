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
//   * the IDFT is a Stockham autosort FFT: each lane holds P = NC/T complex
//     points, radix-R butterflies run in registers, passes exchange data through
//     an XOR-swizzled (bank-conflict-free) LDS buffer (A7);
//   * real input uses the N/2-point complex FFT + untangle; the partner bin
//     lives in lane (T - l) and is fetched with ds_bpermute (no LDS memory);
//   * magnitude, crop, averaging, epsilon, dB and the DC mask are the epilogue
//     (A8-A10); only D floats per A-scan are written.
// MFMA is not used: the path is HBM/LDS/VALU bound elementwise + FFT work.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "fdoct_kernels.h"
#include "fft_consts.h"

namespace fdoct {

// Stage-skipping profiling aid (tools/ablate.sh): only in builds with -DFDOCT_RUNTIME_ABLATE.
#ifdef FDOCT_RUNTIME_ABLATE
#define FDOCT_ABL(bit) ((a.ablate & (bit)) != 0)
#else
#define FDOCT_ABL(bit) false
#endif

// ---------------------------------------------------------------- helpers --
template <int I>
using IC = std::integral_constant<int, I>;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(IC<B>{});
    static_for<B + 1, E>(f);
  }
}

__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return {a.x - b.x, a.y - b.y}; }
// Hardware v_sqrt_f32 / v_log_f32 (1 ulp) without the library's denormal-range fix-ups:
// magnitudes are sums of >= 512 products and the log argument is >= epsilon = 1e-6.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return {fmaf(-a.y, b.y, a.x * b.x), fmaf(a.y, b.x, a.x * b.y)};
}

// multiply by exp(+-2*pi*i*J/R) with J, R compile-time (R divides 64)
template <int J, int R, bool INV>
__device__ __forceinline__ float2 twc(float2 v) {
  constexpr int j = ((J % R) + R) % R;
  if constexpr (j == 0) {
    return v;
  } else if constexpr (2 * j == R) {
    return {-v.x, -v.y};
  } else if constexpr (4 * j == R) {
    return INV ? float2{-v.y, v.x} : float2{v.y, -v.x};
  } else if constexpr (4 * j == 3 * R) {
    return INV ? float2{v.y, -v.x} : float2{-v.y, v.x};
  } else {
    constexpr int idx = j * (64 / R);
    constexpr float c = COS64[idx];
    constexpr float s = INV ? SIN64[idx] : -SIN64[idx];
    return {fmaf(-v.y, s, v.x * c), fmaf(v.y, c, v.x * s)};
  }
}

// In-register R-point DFT, natural order in and out.  R in {1,2,4,8,16,32}.
template <int R, bool INV>
__device__ __forceinline__ void fft_reg(float2* v) {
  if constexpr (R == 1) {
  } else if constexpr (R == 2) {
    float2 a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
  } else if constexpr (R == 4) {
    float2 t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], d = v[1] - v[3];
    float2 t3 = INV ? float2{-d.y, d.x} : float2{d.y, -d.x};
    v[0] = t0 + t2;
    v[1] = t1 + t3;
    v[2] = t0 - t2;
    v[3] = t1 - t3;
  } else {
    constexpr int Rb = R / 4;
    static_for<0, Rb>([&](auto n2c) {
      constexpr int n2 = decltype(n2c)::value;
      float2 t[4] = {v[n2], v[Rb + n2], v[2 * Rb + n2], v[3 * Rb + n2]};
      fft_reg<4, INV>(t);
      static_for<0, 4>([&](auto k1c) {
        constexpr int k1 = decltype(k1c)::value;
        v[k1 * Rb + n2] = twc<k1 * n2, R, INV>(t[k1]);
      });
    });
    static_for<0, 4>([&](auto k1c) {
      constexpr int k1 = decltype(k1c)::value;
      fft_reg<Rb, INV>(v + k1 * Rb);
    });
    float2 o[R];
    static_for<0, R>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      o[(i / Rb) + 4 * (i % Rb)] = v[i];
    });
    static_for<0, R>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      v[i] = o[i];
    });
  }
}

// Orders this wave's LDS traffic: a wave's DS operations execute in program
// order, so a compiler-level fence is all a same-wave write->read hand-off needs.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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
__device__ __forceinline__ void pass_twiddles(float2* twr, int l, const float2* tw) {
  constexpr int P = NC / T;
  constexpr int NB = P / R;
  static_for<0, NB>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const float2* twk = tw + ((l + T * t) & (NS - 1));
    static_for<1, R>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      twr[t * (R - 1) + (r - 1)] = twk[(r - 1) * NS];
    });
  });
}

template <int NC, int T, int R, int NS, bool LAST, int LP, bool INV>
__device__ __forceinline__ void pass_compute(float2* z, int l, float2* xch, const float2* twr) {
  constexpr int P = NC / T;
  constexpr int NB = P / R;  // butterflies per lane
  static_assert(P % R == 0, "radix must divide the per-lane point count");
  static_assert(NS == 1 || NS >= (1 << LP), "later passes must keep pad-aligned strides");
  static_for<0, NB>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const int j = l + T * t;
    const int k = j & (NS - 1);
    float2 v[R];
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
      float2* dst = xch + (e0 + (e0 >> LP));
      static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        dst[padded<LP>(r * NS)] = v[r];  // (e0 + r*NS) >> LP == (e0 >> LP) + ((r*NS) >> LP) here
      });
    }
  });
}

template <int NC, int T, int LP>
__device__ __forceinline__ void pass_readback(float2* z, int l, const float2* xch) {
  constexpr int P = NC / T;
  if constexpr (T >= (1 << LP)) {
    const float2* src = xch + (l + (l >> LP));
    static_for<0, P>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      z[m] = src[padded<LP>(T * m)];
    });
  } else {
    const float2* src = xch + l;  // l < T < 2^LP: the pad term depends on m only
    static_for<0, P>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      z[m] = src[T * m + ((T * m) >> LP)];
    });
  }
}

// ------------------------------------------------------------ input types --
template <typename IN_T>
struct RawChunk;
template <>
struct RawChunk<uint16_t> {
  uint4 v;
  __device__ __forceinline__ void load(const void* row, int i0) {
    v = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(row) + i0);
  }
  __device__ __forceinline__ void zero() { v = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void pin() { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
  __device__ __forceinline__ void unpack(float* x) const {
    x[0] = (float)(v.x & 0xffffu); x[1] = (float)(v.x >> 16);
    x[2] = (float)(v.y & 0xffffu); x[3] = (float)(v.y >> 16);
    x[4] = (float)(v.z & 0xffffu); x[5] = (float)(v.z >> 16);
    x[6] = (float)(v.w & 0xffffu); x[7] = (float)(v.w >> 16);
  }
};
template <>
struct RawChunk<uint8_t> {
  uint2 v;
  __device__ __forceinline__ void load(const void* row, int i0) {
    v = *reinterpret_cast<const uint2*>(static_cast<const uint8_t*>(row) + i0);
  }
  __device__ __forceinline__ void zero() { v = make_uint2(0, 0); }
  __device__ __forceinline__ void pin() { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
  __device__ __forceinline__ void unpack(float* x) const {
    x[0] = (float)(v.x & 0xffu); x[1] = (float)((v.x >> 8) & 0xffu);
    x[2] = (float)((v.x >> 16) & 0xffu); x[3] = (float)(v.x >> 24);
    x[4] = (float)(v.y & 0xffu); x[5] = (float)((v.y >> 8) & 0xffu);
    x[6] = (float)((v.y >> 16) & 0xffu); x[7] = (float)(v.y >> 24);
  }
};
template <>
struct RawChunk<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const void* row, int i0) {
    const float4* p = reinterpret_cast<const float4*>(static_cast<const float*>(row) + i0);
    a = p[0];
    b = p[1];
  }
  __device__ __forceinline__ void zero() { a = b = make_float4(0, 0, 0, 0); }
  __device__ __forceinline__ void pin() {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
  }
  __device__ __forceinline__ void unpack(float* x) const {
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
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
// One DPP step of a wave-wide f64 sum: v + (v moved by `CTRL`), lanes masked out by the row/bank
// masks (or reading past the row edge) contribute 0.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int tlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, false);
  const int thi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, false);
  return v + __hiloint2double(thi, tlo);
}

// Sum over the T lanes of a row group, returned to every lane of the group.  T == 64: DPP
// row shifts / broadcasts (no LDS, no address registers) and a scalar broadcast of lane 63.
template <int T>
__device__ __forceinline__ double group_sum(double v) {
  if constexpr (T == 64) {
    v = dpp_add_f64<0x111, 0xf, 0xf>(v);  // row_shr:1
    v = dpp_add_f64<0x112, 0xf, 0xf>(v);  // row_shr:2
    v = dpp_add_f64<0x114, 0xf, 0xe>(v);  // row_shr:4, banks 1-3
    v = dpp_add_f64<0x118, 0xf, 0xc>(v);  // row_shr:8, banks 2-3
    v = dpp_add_f64<0x142, 0xa, 0xf>(v);  // row_bcast:15 -> rows 1,3
    v = dpp_add_f64<0x143, 0xc, 0xf>(v);  // row_bcast:31 -> rows 2,3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
  } else {
#pragma unroll
    for (int m = T / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
  }
}

// ------------------------------------------------------------ fused kernel --
// LOG2NC: log2 of the complex FFT length NC (= N/2 real path, N complex path)
// T: lanes per row (64/T rows per wave); R1*R2*R3 = NC (R3 = 1: two passes);
// WCH: 8-sample chunks per lane (W <= WC = 8*T*WCH); CPLX: dispersion phase path.
// LEAN: the benchmark / common acquisition configuration, compiled without any
//   predication: W == WC, averages == 1, 1-row background, no pi/dark frame, no
//   normalisation, D % T == 0.  !LEAN handles everything else.
template <int LOG2NC, int T, int R1, int R2, int R3, int WCH, typename IN_T, bool CPLX, bool LEAN>
__global__ __launch_bounds__(FDOCT_MAX_BLOCK) void fused_kernel(const FusedArgs a) {
  constexpr int NC = 1 << LOG2NC;
  constexpr int P = NC / T;
  constexpr int RPW = 64 / T;  // rows per wave
  constexpr int WC = 8 * T * WCH;
  constexpr int LP = (R1 == 32) ? 5 : (R1 == 16) ? 4 : (R1 == 8) ? 3 : 2;
  static_assert(R1 * R2 * R3 == NC, "radix plan");
  constexpr int NPASS = (R3 > 1) ? 3 : 2;

  extern __shared__ __align__(16) unsigned char smem[];
  float* c_ib = reinterpret_cast<float*>(smem);  // [WC] 1/background
  float* c_win = c_ib + WC;                      // [WC] window
  float* c_g = c_win + WC;                       // [WC] fractionalk by sample index
  float2* c_tw = reinterpret_cast<float2*>(c_g + WC);  // twiddle tables, a.tw_count entries
  float2* c_ph = c_tw + a.tw_count;                    // [NC] phase (CPLX only)
  uint32_t* c_gi = reinterpret_cast<uint32_t*>(c_ph + (CPLX ? NC : 0));  // [NC] packed gather offsets
  unsigned char* scratch0 = reinterpret_cast<unsigned char*>(c_gi + NC);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int l = lane & (T - 1);
  const int sub = lane / T;

  // ---- stage the per-column constants once per workgroup
  // layout: sample i = 8*(lane + T*c) + e  ->  slot c*8T + (e>>2)*4T + 4*lane + (e&3), i.e. each
  // chunk is split into two planes of 4 floats per lane, so a wave's b128 reads are contiguous
  for (int i = tid; i < WC; i += blockDim.x) {
    const bool in = i < a.W;
    const int e = i & 7, ln = (i >> 3) & (T - 1), c = i / (8 * T);
    const int slot = c * 8 * T + (e >> 2) * 4 * T + 4 * ln + (e & 3);
    c_ib[slot] = (in && a.ib) ? a.ib[i] : 0.f;
    c_win[slot] = in ? a.win[i] : 0.f;
    c_g[slot] = in ? a.g[i] : 0.f;
  }
  for (int i = tid; i < a.tw_count; i += blockDim.x) c_tw[i] = a.tw[i];
  if constexpr (CPLX)
    for (int i = tid; i < NC; i += blockDim.x) c_ph[i] = a.phase[i];
  // gather table: entry n = ln + T*m is stored at [(m/4)][ln][m%4] so a lane's P entries are P/4
  // b128 reads with a 16-byte lane stride (re-read every row: cheaper than P resident VGPRs)
  for (int i = tid; i < NC; i += blockDim.x) {
    const int ln = i & (T - 1), m = i / T;
    c_gi[(m >> 2) * 4 * T + 4 * ln + (m & 3)] = a.gidx[i];
  }
  __syncthreads();

  unsigned char* scr = scratch0 + (size_t)(wave * RPW + sub) * a.scratch_bytes;
  float* stg = reinterpret_cast<float*>(scr);
  float2* xch = reinterpret_cast<float2*>(scr);

  // ---- per-lane constants kept in registers for every row
  float2 utw = make_float2(1.f, 0.f);
  if constexpr (!CPLX) utw = a.utw[l];  // exp(+2*pi*i*l/N)

  const float2* tw_p2 = c_tw;                    // pass 2 table: (R2-1) x R1
  const float2* tw_p3 = c_tw + (R2 - 1) * R1;    // pass 3 table: (R3-1) x (R1*R2)

  const long long total = a.total_out_rows;
  const long long wstride = (long long)gridDim.x * nwaves * RPW;
  long long o_wave = ((long long)blockIdx.x * nwaves + wave) * RPW;  // wave-uniform

  const int W = LEAN ? WC : a.W;
  const int A = LEAN ? 1 : a.A;
  const unsigned char* frames = static_cast<const unsigned char*>(a.frames);
  const int i0l = 8 * l;  // this lane's sample offset inside a chunk
  const int c0l = 4 * l;  // this lane's slot inside a constant plane (see the staging loop above)

  RawChunk<IN_T> raw[WCH];
  auto issue_loads = [&](long long o, int avg_i) {
    const bool valid = o < total;
    long long in_row = valid ? o : 0;
    if constexpr (!LEAN) {
      if (A > 1 && valid) {
        const long long g = o / a.H;
        in_row = (g * A + avg_i) * (long long)a.H + (o - g * a.H);
      }
    }
    const void* row = frames + in_row * a.pitch_bytes;
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
  if (o_wave < total) issue_loads(o_wave + sub, 0);
  for (; o_wave < total; o_wave += wstride) {
    const long long o = o_wave + sub;
    const bool valid = o < total;
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
      // ---------------- A2: dark, normalise, pi frame, background (v was unpacked at the end of the
      // previous pass, see below)
      float v[8 * WCH];
#pragma unroll
      for (int c = 0; c < WCH; c++) raw[c].unpack(v + 8 * c);

      if constexpr (!LEAN) {
        const long long in_frame = gi * A + ai;
        if (a.yd) {
          const float* ydr = a.yd + (a.yd_2d ? (size_t)r * W : 0);
#pragma unroll
          for (int c = 0; c < WCH; c++) {
            const int i0 = i0l + 8 * T * c;
            if (i0 < W) {
#pragma unroll
              for (int e = 0; e < 8; e++) v[8 * c + e] -= ydr[i0 + e];
            }
          }
        }
        if (a.rowwisenormalize) {  // main:88-97,1126
          float mn = INFINITY, mx = -INFINITY;
#pragma unroll
          for (int c = 0; c < WCH; c++) {
            if (i0l + 8 * T * c < W) {
#pragma unroll
              for (int e = 0; e < 8; e++) {
                mn = fminf(mn, v[8 * c + e]);
                mx = fmaxf(mx, v[8 * c + e]);
              }
            }
          }
          mn = group_min<T>(mn);
          mx = group_max<T>(mx);
          const float sc = (mx - mn > 2.220446049250313e-16f) ? 1.f / (mx - mn) : 0.f;
          const float sh = -mn * sc;
#pragma unroll
          for (int i = 0; i < 8 * WCH; i++) v[i] = fmaf(v[i], sc, sh);
        }
        if (a.minmax) {  // main:1128-1129 whole-frame min-max, from the pre-pass
          const float2 mmx = a.minmax[in_frame];
          const float sc = (mmx.y - mmx.x > 2.220446049250313e-16f) ? 1.f / (mmx.y - mmx.x) : 0.f;
          const float sh = -mmx.x * sc;
#pragma unroll
          for (int i = 0; i < 8 * WCH; i++) v[i] = fmaf(v[i], sc, sh);
        }
        if (a.yp) {  // main:1132 (data_y - data_yp)
          const float* ypr = a.yp + (a.yp_2d ? (size_t)r * W : 0);
#pragma unroll
          for (int c = 0; c < WCH; c++) {
            const int i0 = i0l + 8 * T * c;
            if (i0 < W) {
#pragma unroll
              for (int e = 0; e < 8; e++) v[8 * c + e] -= ypr[i0 + e];
            }
          }
        }
      }
      // main:1132 ... / data_yb as a multiply by the host-side reciprocal
      double sum = 0.0;
      if (!FDOCT_ABL(1)) {
        const float* ibl = c_ib + c0l;
        float ibv[8 * WCH];
        bool from_lds = true;
        if constexpr (!LEAN) {
          if (a.ib2d) {
            from_lds = false;
#pragma unroll
            for (int c = 0; c < WCH; c++) {
              const int i0 = i0l + 8 * T * c;
              if (i0 < W) {
                const float4* p4 = reinterpret_cast<const float4*>(a.ib2d + (size_t)r * W + i0);
                const float4 q0 = p4[0], q1 = p4[1];
                ibv[8 * c + 0] = q0.x; ibv[8 * c + 1] = q0.y; ibv[8 * c + 2] = q0.z; ibv[8 * c + 3] = q0.w;
                ibv[8 * c + 4] = q1.x; ibv[8 * c + 5] = q1.y; ibv[8 * c + 6] = q1.z; ibv[8 * c + 7] = q1.w;
              } else {
#pragma unroll
                for (int e = 0; e < 8; e++) ibv[8 * c + e] = 0.f;
              }
            }
          }
        }
        if (from_lds) {
#pragma unroll
          for (int c = 0; c < WCH; c++) {
            const float4 q0 = *reinterpret_cast<const float4*>(ibl + 8 * T * c);
            const float4 q1 = *reinterpret_cast<const float4*>(ibl + 8 * T * c + 4 * T);
            ibv[8 * c + 0] = q0.x; ibv[8 * c + 1] = q0.y; ibv[8 * c + 2] = q0.z; ibv[8 * c + 3] = q0.w;
            ibv[8 * c + 4] = q1.x; ibv[8 * c + 5] = q1.y; ibv[8 * c + 6] = q1.z; ibv[8 * c + 7] = q1.w;
          }
        }
#pragma unroll
        for (int c = 0; c < WCH; c++) {
          float part = 0.f;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            v[8 * c + e] *= ibv[8 * c + e];
            part += v[8 * c + e];
          }
          sum += (double)part;
        }
      }
      // window and slope weights: issued here so their LDS latency hides under the mean reduction
      float wv[8 * WCH], gv[8 * WCH];
      {
        const float* wl = c_win + c0l;
        const float* gl = c_g + c0l;
#pragma unroll
        for (int c = 0; c < WCH; c++) {
          const float4 w0 = *reinterpret_cast<const float4*>(wl + 8 * T * c);
          const float4 w1 = *reinterpret_cast<const float4*>(wl + 8 * T * c + 4 * T);
          const float4 g0 = *reinterpret_cast<const float4*>(gl + 8 * T * c);
          const float4 g1 = *reinterpret_cast<const float4*>(gl + 8 * T * c + 4 * T);
          wv[8 * c + 0] = w0.x; wv[8 * c + 1] = w0.y; wv[8 * c + 2] = w0.z; wv[8 * c + 3] = w0.w;
          wv[8 * c + 4] = w1.x; wv[8 * c + 5] = w1.y; wv[8 * c + 6] = w1.z; wv[8 * c + 7] = w1.w;
          gv[8 * c + 0] = g0.x; gv[8 * c + 1] = g0.y; gv[8 * c + 2] = g0.z; gv[8 * c + 3] = g0.w;
          gv[8 * c + 4] = g1.x; gv[8 * c + 5] = g1.y; gv[8 * c + 6] = g1.z; gv[8 * c + 7] = g1.w;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- A3: DC removal (mean in double), window
      if (!FDOCT_ABL(1)) sum = group_sum<T>(sum);
      const double mean = sum / (double)W;
      const float mh = (float)mean;
      const float ml = (float)(mean - (double)mh);
      if (!FDOCT_ABL(1)) {
#pragma unroll
        for (int i = 0; i < 8 * WCH; i++) v[i] = ((v[i] - mh) - ml) * wv[i];
      }
      // ---------------- A5 (first half): s_i = y_i + g_i * (y_i - y_{i-1})
      // (the reference weights by fractionalk[nearestkindex[q]], a per-SAMPLE
      //  quantity, so the slope step is done here once per sample)
      if (!FDOCT_ABL(2)) {
        float* stl = stg + (a.split ? (i0l >> 1) : i0l);
        float prev_last = 0.f;  // y of the sample just before this lane's chunk
#pragma unroll
        for (int c = 0; c < WCH; c++) {
          float left;
          if constexpr (T == 64) {
            // wave_shr:1 -- lane i takes lane i-1's last sample, lane 0 keeps `old` = the previous
            // chunk's lane-63 sample (no LDS round trip)
            left = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(prev_last), __float_as_int(v[8 * c + 7]),
                                                              0x138, 0xf, 0xf, false));
            if (c + 1 < WCH) prev_last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[8 * c + 7]), 63));
          } else {
            left = __shfl_up(v[8 * c + 7], 1, T);
            if (l == 0) left = prev_last;  // last sample of the previous chunk (lane T-1)
            if (c + 1 < WCH) prev_last = __shfl(v[8 * c + 7], T - 1, T);
          }
          float s[8];
          float first_slope = v[8 * c] - left;
          if (c == 0 && l == 0) first_slope = v[1] - v[0];  // slopes[0] = slopes[1] (main:1161)
          s[0] = fmaf(gv[8 * c], first_slope, v[8 * c]);
#pragma unroll
          for (int e = 1; e < 8; e++) s[e] = fmaf(gv[8 * c + e], v[8 * c + e] - v[8 * c + e - 1], v[8 * c + e]);
          if (LEAN || (i0l + 8 * T * c < W)) {
            if (a.split) {
              *reinterpret_cast<float4*>(stl + 4 * T * c) = make_float4(s[0], s[2], s[4], s[6]);
              *reinterpret_cast<float4*>(stl + 4 * T * c + WC / 2) = make_float4(s[1], s[3], s[5], s[7]);
            } else {
              *reinterpret_cast<float4*>(stl + 8 * T * c) = make_float4(s[0], s[1], s[2], s[3]);
              *reinterpret_cast<float4*>(stl + 8 * T * c + 4) = make_float4(s[4], s[5], s[6], s[7]);
            }
          }
        }
        if (l == 0) stg[WC] = 0.f;  // source of data_ylin[0] and data_ylin[N-1] (defined 0)
      }
      wave_lds_sync();

      // prefetch the next row this group will need (its registers are free from here on)
      {
        long long no = o;
        int na = ai + 1;
        if (LEAN || na == A) {
          na = 0;
          no = o + wstride;
        }
        issue_loads(no, na);
      }

      // ---------------- A5 (second half) + A6: gather into FFT registers
      static_assert(P % 4 == 0, "gather table is read four entries at a time");
      uint32_t gsrc[P];  // packed 16-bit LDS byte offsets (relative to this row's staging buffer)
      {
        const uint4* gl4 = reinterpret_cast<const uint4*>(c_gi) + l;
#pragma unroll
        for (int q = 0; q < P / 4; q++) {
          const uint4 g4 = gl4[q * T];
          gsrc[4 * q + 0] = g4.x; gsrc[4 * q + 1] = g4.y; gsrc[4 * q + 2] = g4.z; gsrc[4 * q + 3] = g4.w;
        }
      }
      float2 z[P];
      if (FDOCT_ABL(2)) {
#pragma unroll
        for (int m = 0; m < P; m++) z[m] = make_float2(v[(2 * m) % (8 * WCH)], v[(2 * m + 1) % (8 * WCH)]);
      } else if constexpr (CPLX) {
        const float2* phl = c_ph + l;
#pragma unroll
        for (int m = 0; m < P; m++) {
          const float y = *reinterpret_cast<const float*>(scr + (gsrc[m] & 0xffffu));
          const float2 ph = phl[T * m];
          z[m] = make_float2(y * ph.x, y * ph.y);
        }
      } else {
#pragma unroll
        for (int m = 0; m < P; m++) {
          z[m].x = *reinterpret_cast<const float*>(scr + (gsrc[m] & 0xffffu));
          z[m].y = *reinterpret_cast<const float*>(scr + (gsrc[m] >> 16));
        }
      }
      wave_lds_sync();

      // ---------------- A7: NC-point inverse DFT
      if (!FDOCT_ABL(4)) {
        constexpr int NTW2 = (P / R2) * (R2 - 1);
        constexpr int NTW3 = (R3 > 1) ? (P / R3) * (R3 - 1) : 1;
        pass_compute<NC, T, R1, 1, false, LP, true>(z, l, xch, nullptr);
        float2 tw2[NTW2];
        pass_twiddles<NC, T, R2, R1>(tw2, l, tw_p2);  // queued behind the exchange writes
        wave_lds_sync();
        pass_readback<NC, T, LP>(z, l, xch);
        wave_lds_sync();
        if constexpr (NPASS == 3) {
          pass_compute<NC, T, R2, R1, false, LP, true>(z, l, xch, tw2);
          float2 tw3[NTW3];
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
#pragma unroll
        for (int m = 0; m < P; m++) acc[m] += fast_sqrt(fmaf(z[m].x, z[m].x, z[m].y * z[m].y));
      } else {
        // partner of e = l + T*m is (NC - e) mod NC: lane (T-l)%T, reg P-1-m (l>0) or (P-m)%P (l==0)
        const int plane = ((lane & ~(T - 1)) | ((T - l) & (T - 1))) << 2;  // byte address for bpermute
        float px[P], py[P];
        // every lane publishes the register its reader wants: lane l' != 0 is read by lane
        // T-l' (!= 0) asking for reg P-1-m; lane 0 is read by lane 0 asking for (P-m)%P.
        // All 2P permutes are issued before any of the arithmetic (one LDS round trip, not P).
        static_for<0, P>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          constexpr int pm1 = P - 1 - m;
          constexpr int pm0 = (P - m) % P;
          const float sx = (l == 0) ? z[pm0].x : z[pm1].x;
          const float sy = (l == 0) ? z[pm0].y : z[pm1].y;
          px[m] = __int_as_float(__builtin_amdgcn_ds_bpermute(plane, __float_as_int(sx)));
          py[m] = __int_as_float(__builtin_amdgcn_ds_bpermute(plane, __float_as_int(sy)));
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, P>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          // A = Z + conj(Zp), B = Z - conj(Zp), O = B/(2i), X = A/2 + w*O
          const float ax = z[m].x + px[m], ay = z[m].y - py[m];
          const float bx = z[m].x - px[m], by = z[m].y + py[m];
          // w = exp(2*pi*i*(l + T*m)/N) = utw * exp(2*pi*i*m/(2P))
          const float2 wm = twc<m, 2 * P, true>(utw);
          const float2 wo = cmul(wm, make_float2(by, -bx));
          const float xr = ax + wo.x, xi = ay + wo.y;  // = 2*X
          acc[m] += 0.5f * fast_sqrt(fmaf(xr, xr, xi * xi));
        });
      }
      // the prefetched samples have had this whole pass to arrive (see the comment at the first issue_loads)
#pragma unroll
      for (int c = 0; c < WCH; c++) raw[c].pin();
    }  // averaging loop

    // ---------------- A9/A10: average, epsilon, dB, DC mask, store
    float outv[P];
#pragma unroll
    for (int m = 0; m < P; m++) outv[m] = LEAN ? (acc[m] + a.eps) : fmaf(acc[m], a.inv_A, a.eps);
    const int D = a.D;
    const int mfull = D / T;  // registers m < mfull are stored by every lane
    auto store_row = [&](float* orow, const float* val) {
#pragma unroll
      for (int m = 0; m < P; m++) {
        if (m < mfull)
          orow[T * m] = val[m];
        else if (!LEAN && l + T * m < D)
          orow[T * m] = val[m];
      }
    };
    if (valid && a.out_mag && !FDOCT_ABL(128)) store_row(a.out_mag + (size_t)o * D + l, outv);
    if (a.out_db) {
      float db[P];
#pragma unroll
      for (int m = 0; m < P; m++) db[m] = FDOCT_ABL(64) ? outv[m] : a.db_scale * fast_log2(outv[m]);  // db_scale carries ln 2
      if (a.dcmask && T > 4) {
        const float d4 = __shfl(db[0], (lane & ~(T - 1)) | 4, 64);
        if (l < 2) db[0] = d4;
      }
      if (valid && !FDOCT_ABL(128)) store_row(a.out_db + (size_t)o * D + l, db);
    }
  }
}

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

__global__ void f64_to_f32_kernel(const double* in, long long pitch_elems, float* out, int W, long long rows) {
  const long long n = rows * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / W;
    const int c = (int)(i - r * W);
    out[i] = (float)in[r * pitch_elems + c];
  }
}

// ---------------------------------------------------------------- dispatch --
template <int LOG2NC, int T, int R1, int R2, int R3, int WCH, typename IN_T, bool CPLX, bool LEAN>
static hipError_t launch_one(const FusedArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  auto k = fused_kernel<LOG2NC, T, R1, R2, R3, WCH, IN_T, CPLX, LEAN>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, grid, block, lds, st, a);
  return hipGetLastError();
}

template <int LOG2NC, int T, int R1, int R2, int R3, int WCH, bool CPLX>
static hipError_t launch_typed(const FusedArgs& a, int dtype, bool lean, dim3 grid, dim3 block, size_t lds,
                               hipStream_t st) {
  switch (dtype) {
    case FDOCT_K_U16:
      return lean ? launch_one<LOG2NC, T, R1, R2, R3, WCH, uint16_t, CPLX, true>(a, grid, block, lds, st)
                  : launch_one<LOG2NC, T, R1, R2, R3, WCH, uint16_t, CPLX, false>(a, grid, block, lds, st);
    case FDOCT_K_U8:  // 8-bit cameras: general kernel only (keeps the build small)
      return launch_one<LOG2NC, T, R1, R2, R3, WCH, uint8_t, CPLX, false>(a, grid, block, lds, st);
    case FDOCT_K_F32:
      return launch_one<LOG2NC, T, R1, R2, R3, WCH, float, CPLX, false>(a, grid, block, lds, st);
    default:
      return hipErrorInvalidValue;
  }
}

// The table of compiled plans: {id, nc, T, R1, R2, R3, WCH}.  nc = complex FFT length.
#define FDOCT_PLANS(X)            \
  X(0, 8, 16, 16, 16, 1, 4)       \
  X(1, 9, 16, 32, 16, 1, 8)       \
  X(2, 10, 64, 16, 16, 4, 4)      \
  X(3, 10, 32, 32, 32, 1, 8)      \
  X(4, 11, 64, 32, 8, 8, 8)

int fused_plan_count() {
  int n = 0;
#define FDOCT_COUNT(ID, L2, T_, R1_, R2_, R3_, WCH_) n++;
  FDOCT_PLANS(FDOCT_COUNT)
#undef FDOCT_COUNT
  return n;
}

bool fused_plan_get(int id, FusedPlan* p) {
#define FDOCT_GET(ID, L2, T_, R1_, R2_, R3_, WCH_) \
  if (id == ID) {                                  \
    *p = FusedPlan{ID, 1 << L2, T_, R1_, R2_, R3_, WCH_}; \
    return true;                                   \
  }
  FDOCT_PLANS(FDOCT_GET)
#undef FDOCT_GET
  return false;
}

hipError_t launch_fused(const FusedPlan& p, const FusedArgs& a, int dtype, bool cplx, bool lean, int grid, int block,
                        size_t lds, hipStream_t st) {
  dim3 g(grid), b(block);
#define FDOCT_CASE(ID, L2, T_, R1_, R2_, R3_, WCH_)                                                    \
  if (p.id == ID) {                                                                                    \
    return cplx ? launch_typed<L2, T_, R1_, R2_, R3_, WCH_, true>(a, dtype, lean, g, b, lds, st)       \
                : launch_typed<L2, T_, R1_, R2_, R3_, WCH_, false>(a, dtype, lean, g, b, lds, st);     \
  }
  FDOCT_PLANS(FDOCT_CASE)
#undef FDOCT_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_minmax(const void* frames, int dtype, long long pitch_bytes, int W, int H, int nframes,
                         const float* yd, int yd_2d, float2* out, hipStream_t st) {
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
  dim3 b(32, 8), g((cols + 31) / 32, (rows + 31) / 32, groups);
  hipLaunchKernelGGL(transpose_kernel, g, b, 0, st, in, out, rows, cols);
  return hipGetLastError();
}

hipError_t launch_f64_to_f32(const double* in, long long pitch_elems, float* out, int W, long long rows, hipStream_t st) {
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3(2048), dim3(256), 0, st, in, pitch_elems, out, W, rows);
  return hipGetLastError();
}

}  // namespace fdoct
