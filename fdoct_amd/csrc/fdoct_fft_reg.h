// fdoct_fft_reg.h -- register-level building blocks shared by the fused kernels (fdoct_kernels.hip) and the
// any-configuration kernel (fdoct_generic.hip): 2-vector (packed f32) complex arithmetic and in-register
// radix-R DFTs for CDNA4 (gfx950).  Device code only; include inside namespace-less translation units after
// <hip/hip_runtime.h>.
#pragma once
#ifndef __HIPCC_RTC__  // (the run-time compiler of fdoct_jit.cpp brings its own HIP declarations and has no system headers)
#include <hip/hip_runtime.h>

#include <type_traits>
#endif

#include "fft_consts.h"

namespace fdoct {

typedef float v2f __attribute__((ext_vector_type(2)));  // one complex value / two adjacent samples

// a compile-time index as a value (what std::integral_constant<int, I> is, without the header)
template <int I>
struct IC {
  static constexpr int value = I;
  constexpr operator int() const { return I; }
};

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(IC<B>{});
    static_for<B + 1, E>(f);
  }
}

__device__ __forceinline__ v2f mk(float x, float y) { return (v2f){x, y}; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// A4's re-packing (main:215-241), for any width.  The padded spectrum has n = W + 2 floor((M W - W) / 2) bins (M W, or M W - 1
// for an odd width under an even multiplier) and cv::dft(DFT_INVERSE | DFT_REAL_OUTPUT) reads bins 0 .. n/2 of it, the rest
// by Hermitian symmetry.  With c = floor(W / 2), bin k of the padded spectrum is F[k] for k < c -- the row's Nyquist bin is
// dropped (fftshift put it on the negative side) -- and, for an ODD width, bin c is F[W - 1]: the fftshift of main:215-227
// swaps two halves of c columns and leaves the last column where it is.  Returns the source bin of position `pos` (or -1:
// zero) and whether it is the mirror image (conjugate).  BscanDark's band-pass (dark:218-236) keeps 3 <= k < W / 10 -- and the
// odd width's stray column, which lies outside the ranges it blanks.  (One rule for generic_kernel's full-length zero-pad
// stage and the long-row path.)
__device__ __forceinline__ int pad_source(int W, int n, int bandpass, int pos, bool* mirror) {
  const int c = W >> 1, odd = W & 1, klim = odd ? c + 1 : c;
  const int kk = (pos < klim) ? pos : ((pos != 0 && n - pos < klim) ? n - pos : -1);
  *mirror = pos >= klim;
  if (kk < 0) return -1;
  const bool stray = odd && kk == c;
  if (bandpass && !stray && (kk < 3 || kk >= W / 10)) return -1;
  return stray ? W - 1 : kk;
}

// s = fl(a - b) and err = (a - b) - s exactly (Knuth's TwoSum on a and -b: six operations, no assumption on the magnitudes).
// The reference subtracts the dark frame, the normalisation's minimum and the pi frame in double (dark:1269, main:1126-1132);
// here each of those f32 differences hands its residual to the sample's low word, so none of them rounds at the size of the
// DC level (round 6: a non-integer dark frame under fringes of 1e-3 of the DC level on a 96-sample row was 0.55 x the tolerance
// away from the chain evaluated in double; the files are built with -ffp-contract=off and without -ffast-math).
__device__ __forceinline__ float two_diff(float a, float b, float& err) {
  const float s = a - b;
  const float bb = s - a;
  err = (a - (s - bb)) - (b + bb);
  return s;
}
__device__ __forceinline__ v2f two_diff(v2f a, v2f b, v2f& err) {
  const v2f s = a - b;
  const v2f bb = s - a;
  err = (a - (s - bb)) - (b + bb);
  return s;
}

// One bin of a real row's DFT in double, F[k] = sum_m x[m] e^(-2 pi i k m / W), summed T samples at a time (BscanDark's band-pass
// keeps a few bins of the row's spectrum; what is displayed afterwards is tiny against the row, so those bins are evaluated in
// double, directly -- fdoct_generic.hip, fdoct_wave_dev.h).  The T phasors e^(-2 pi i k t / W), t < T, stay in registers: a chunk
// costs two fmas per sample for c = sum_t x[m + t] p_t, four for the sum += b c with b = e^(-2 pi i k m / W) and four to advance b
// by e^(-2 pi i k T / W) (recurrences in double: their error after W steps is W x 1e-16).
template <int T>
struct DftBinF64 {
  double pr[T], pi[T], br, bi, sr, si, ar, ai;
  // bin k of a W-sample row; this accumulator starts at sample m0
  __device__ __forceinline__ void init(int k, int m0, int W) {
    const double inv_w = 1.0 / (double)W;
    double s1r, s1i;
    sincospi(-2.0 * (double)k * inv_w, &s1i, &s1r);
    pr[0] = 1.0;
    pi[0] = 0.0;
#pragma unroll
    for (int t = 1; t < T; t++) {
      pr[t] = fma(pr[t - 1], s1r, -pi[t - 1] * s1i);
      pi[t] = fma(pr[t - 1], s1i, pi[t - 1] * s1r);
    }
    sr = fma(pr[T - 1], s1r, -pi[T - 1] * s1i);
    si = fma(pr[T - 1], s1i, pi[T - 1] * s1r);
    sincospi(-2.0 * (double)((k * m0) % W) * inv_w, &bi, &br);
    ar = ai = 0.0;
  }
  __device__ __forceinline__ void chunk(const double* x) {  // x[0 .. T-1], zeros past the end of the row
    double cr = x[0], ci = 0.0;
#pragma unroll
    for (int t = 1; t < T; t++) {
      cr = fma(x[t], pr[t], cr);
      ci = fma(x[t], pi[t], ci);
    }
    ar = fma(br, cr, fma(-bi, ci, ar));
    ai = fma(br, ci, fma(bi, cr, ai));
    const double nr = fma(br, sr, -bi * si);
    bi = fma(br, si, bi * sr);
    br = nr;
  }
};

// Hardware v_sqrt_f32 / v_log_f32 (1 ulp) without the library's denormal-range fix-ups:
// magnitudes are sums of >= 512 products and the log argument is >= epsilon = 1e-6.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// Complex multiply of two run-time values in two packed instructions:
//   t = (a.x*b.x, a.x*b.y);  r = (a.y*(-b.y) + t.x, a.y*b.x + t.y)
// The second one needs a half swap and a negation on b that hipcc does not fold into the
// v_pk_fma_f32 modifiers from C++ (it emits v_xor + v_mov instead), hence the asm.
// (a.x + b.x, a.y - b.y) and (a.x - b.x, a.y + b.y): a +- conj(b) in one packed add each
#ifdef FDOCT_X_NO_PK  // tuning experiment: the same arithmetic on single-lane-pair VOP2/VOP3 instructions (energy per instruction)
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) { return mk(a.x + b.x, a.y - b.y); }
__device__ __forceinline__ v2f sub_conj(v2f a, v2f b) { return mk(a.x - b.x, a.y + b.y); }
__device__ __forceinline__ v2f add_mulmi(v2f a, v2f b) { return mk(a.x + b.y, a.y - b.x); }
__device__ __forceinline__ v2f sub_mulmi(v2f a, v2f b) { return mk(a.x - b.y, a.y + b.x); }
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
  return mk(__builtin_fmaf(a.y, -b.y, a.x * b.x), __builtin_fmaf(a.y, b.x, a.x * b.y));
}
#else
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2f sub_conj(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a + (-i)*b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2f add_mulmi(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a - (-i)*b = a + i*b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2f sub_mulmi(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
  v2f t = a.xx * b;
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=v"(r)
      : "v"(a), "v"(b), "v"(t));
  return r;
}

#endif

// multiply by the compile-time constant exp(+-2*pi*i*J/R)  (R divides 64)
template <int J, int R, bool INV>
__device__ __forceinline__ v2f twc(v2f v) {
  constexpr int j = ((J % R) + R) % R;
  if constexpr (j == 0) {
    return v;
  } else if constexpr (2 * j == R) {
    return -v;
  } else if constexpr (4 * j == R) {
    return INV ? mk(-v.y, v.x) : mk(v.y, -v.x);
  } else if constexpr (4 * j == 3 * R) {
    return INV ? mk(v.y, -v.x) : mk(-v.y, v.x);
  } else {
    constexpr int idx = j * (64 / R);
    constexpr float c = COS64[idx];
    constexpr float s = INV ? SIN64[idx] : -SIN64[idx];
    // v.x*(c, s) + v.y*(-s, c): both constant pairs are literals, no swizzle of v is needed
    return pk_fma(v.yy, mk(-s, c), v.xx * mk(c, s));
  }
}

// In-register R-point DFT, natural order in and out.  R in {1,2,4,8,16,32}.
template <int R, bool INV>
__device__ __forceinline__ void fft_reg(v2f* v) {
  if constexpr (R == 1) {
  } else if constexpr (R == 2) {
    v2f a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
  } else if constexpr (R == 4) {
    v2f t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], d = v[1] - v[3];
    // t1 +- i*d (inverse) / t1 -+ i*d (forward): the quarter turn rides on the packed add's modifiers
    v[0] = t0 + t2;
    v[1] = INV ? sub_mulmi(t1, d) : add_mulmi(t1, d);
    v[2] = t0 - t2;
    v[3] = INV ? add_mulmi(t1, d) : sub_mulmi(t1, d);
  } else {
    constexpr int Rb = R / 4;
    static_for<0, Rb>([&](auto n2c) {
      constexpr int n2 = decltype(n2c)::value;
      v2f t[4] = {v[n2], v[Rb + n2], v[2 * Rb + n2], v[3 * Rb + n2]};
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
    v2f o[R];
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


// 3- and 5-point DFTs (exponent sign + for INV), natural order in and out -- for N with factors 3 and 5.
template <bool INV>
__device__ __forceinline__ void fft_reg3(v2f* v) {
  constexpr float s = INV ? 0.86602540378443864676f : -0.86602540378443864676f;
  const v2f t = v[1] + v[2], d = v[1] - v[2];
  const v2f m = pk_fma(t, mk(-0.5f, -0.5f), v[0]);
  const v2f r = mk(-s * d.y, s * d.x);  // i*s*d
  v[0] = v[0] + t;
  v[1] = m + r;
  v[2] = m - r;
}
template <bool INV>
__device__ __forceinline__ void fft_reg5(v2f* v) {
  constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
  constexpr float s1 = INV ? 0.95105651629515357212f : -0.95105651629515357212f;
  constexpr float s2 = INV ? 0.58778525229247312917f : -0.58778525229247312917f;
  const v2f a1 = v[1] + v[4], b1 = v[1] - v[4], a2 = v[2] + v[3], b2 = v[2] - v[3];
  const v2f m1 = pk_fma(a2, mk(c2, c2), pk_fma(a1, mk(c1, c1), v[0]));
  const v2f m2 = pk_fma(a2, mk(c1, c1), pk_fma(a1, mk(c2, c2), v[0]));
  const v2f u1 = pk_fma(b2, mk(s2, s2), b1 * mk(s1, s1));   // s1 b1 + s2 b2
  const v2f u2 = pk_fma(b2, mk(-s1, -s1), b1 * mk(s2, s2));  // s2 b1 - s1 b2
  const v2f r1 = mk(-u1.y, u1.x), r2 = mk(-u2.y, u2.x);      // i*u
  v[0] = v[0] + (a1 + a2);
  v[1] = m1 + r1;
  v[4] = m1 - r1;
  v[2] = m2 + r2;
  v[3] = m2 - r2;
}

// 9-point DFT in registers: 3 x 3 Cooley-Tukey (inputs v[3 r1 + r2], outputs v[k1 + 3 k2]), four non-trivial twiddles
__device__ __forceinline__ constexpr float cos9(int j) {
  constexpr float c[9] = {1.0f, 0.76604444311897801f, 0.17364817766693041f, -0.49999999999999978f, -0.93969262078590832f, -0.93969262078590843f, -0.50000000000000044f, 0.17364817766692997f, 0.76604444311897779f};
  return c[((j % 9) + 9) % 9];
}
__device__ __forceinline__ constexpr float sin9(int j) {
  constexpr float c[9] = {0.0f, 0.64278760968653925f, 0.98480775301220802f, 0.86602540378443871f, 0.34202014332566888f, -0.34202014332566866f, -0.86602540378443837f, -0.98480775301220813f, -0.64278760968653958f};
  return c[((j % 9) + 9) % 9];
}
template <bool INV>
__device__ __forceinline__ void fft_reg9(v2f* v) {
  v2f a[9];  // a[k1 * 3 + r2]
  static_for<0, 3>([&](auto r2c) {
    constexpr int r2 = decltype(r2c)::value;
    v2f t[3] = {v[r2], v[3 + r2], v[6 + r2]};
    fft_reg3<INV>(t);
    static_for<0, 3>([&](auto k1c) {
      constexpr int k1 = decltype(k1c)::value;
      constexpr int j = (r2 * k1) % 9;
      if constexpr (j == 0) {
        a[k1 * 3 + r2] = t[k1];
      } else {
        constexpr float c = cos9(j), sn = INV ? sin9(j) : -sin9(j);
        a[k1 * 3 + r2] = pk_fma(t[k1].yy, mk(-sn, c), t[k1].xx * mk(c, sn));  // t * (c + i sn)
      }
    });
  });
  static_for<0, 3>([&](auto k1c) {
    constexpr int k1 = decltype(k1c)::value;
    fft_reg3<INV>(a + k1 * 3);
    static_for<0, 3>([&](auto k2c) {
      constexpr int k2 = decltype(k2c)::value;
      v[k1 + 3 * k2] = a[k1 * 3 + k2];
    });
  });
}

// 15-point DFT in registers: 3 x 5 Cooley-Tukey (inputs v[5 r1 + r2], outputs v[k1 + 3 k2])
__device__ __forceinline__ constexpr float cos15(int j) {
  constexpr float c[15] = {1.0f, 0.91354545764260087f, 0.66913060635885824f, 0.30901699437494745f, -0.10452846326765333f, -0.49999999999999978f, -0.80901699437494734f, -0.97814760073380569f, -0.97814760073380569f, -0.80901699437494756f, -0.50000000000000044f, -0.10452846326765423f, 0.30901699437494723f, 0.66913060635885846f, 0.91354545764260098f};
  return c[((j % 15) + 15) % 15];
}
__device__ __forceinline__ constexpr float sin15(int j) {
  constexpr float c[15] = {0.0f, 0.40673664307580015f, 0.74314482547739413f, 0.95105651629515353f, 0.9945218953682734f, 0.86602540378443871f, 0.58778525229247325f, 0.20791169081775931f, -0.20791169081775907f, -0.58778525229247303f, -0.86602540378443837f, -0.99452189536827329f, -0.95105651629515364f, -0.74314482547739402f, -0.40673664307580015f};
  return c[((j % 15) + 15) % 15];
}
template <bool INV>
__device__ __forceinline__ void fft_reg15(v2f* v) {
  v2f a[15];  // a[k1 * 5 + r2]
  static_for<0, 5>([&](auto r2c) {
    constexpr int r2 = decltype(r2c)::value;
    v2f t[3] = {v[r2], v[5 + r2], v[10 + r2]};
    fft_reg3<INV>(t);
    static_for<0, 3>([&](auto k1c) {
      constexpr int k1 = decltype(k1c)::value;
      constexpr int j = (r2 * k1) % 15;
      if constexpr (j == 0) {
        a[k1 * 5 + r2] = t[k1];
      } else {
        constexpr float c = cos15(j), sn = INV ? sin15(j) : -sin15(j);
        a[k1 * 5 + r2] = pk_fma(t[k1].yy, mk(-sn, c), t[k1].xx * mk(c, sn));  // t * (c + i sn)
      }
    });
  });
  static_for<0, 3>([&](auto k1c) {
    constexpr int k1 = decltype(k1c)::value;
    fft_reg5<INV>(a + k1 * 5);
    static_for<0, 5>([&](auto k2c) {
      constexpr int k2 = decltype(k2c)::value;
      v[k1 + 3 * k2] = a[k1 * 5 + k2];
    });
  });
}

// 20-point DFT in registers (natural order in and out) as 4 x 5: input r = 5 r1 + r2, output k = k1 + 4 k2,
//   X[k1 + 4 k2] = sum_r2 W_5^(r2 k2) [ W_20^(r2 k1) sum_r1 x[5 r1 + r2] W_4^(r1 k1) ].
// This is two Stockham passes (radix 5 with stride 1, then radix 4 with stride 5) whose data stays in one lane when the
// transform length is a multiple of 64*20, fused: no LDS round trip and compile-time twiddles between them.
__device__ __forceinline__ constexpr float cos20(int j) {
  constexpr float c[20] = {1.f, 0.95105651629515353f, 0.80901699437494745f, 0.58778525229247314f, 0.30901699437494745f, 0.f,
                           -0.30901699437494745f, -0.58778525229247314f, -0.80901699437494745f, -0.95105651629515353f, -1.f,
                           -0.95105651629515353f, -0.80901699437494745f, -0.58778525229247314f, -0.30901699437494745f, 0.f,
                           0.30901699437494745f, 0.58778525229247314f, 0.80901699437494745f, 0.95105651629515353f};
  return c[((j % 20) + 20) % 20];
}
__device__ __forceinline__ constexpr float sin20(int j) { return cos20(j - 5); }
// MIDZERO: inputs 5 .. 14 are zero (the first pass of the zero-pad stage's inverse transform: a low band, zeros, a high band) --
// every radix-4 butterfly of the first stage has two inputs, (a, 0, 0, d) -> a + d, a -+ i d, a - d, a +- i d: four packed adds
// instead of eight, the same values bit for bit.
template <bool INV, bool MIDZERO = false>
__device__ __forceinline__ void fft_reg20(v2f* v) {
  v2f a[20];  // a[k1 * 5 + r2]
  static_for<0, 5>([&](auto r2c) {
    constexpr int r2 = decltype(r2c)::value;
    v2f t[4] = {v[r2], v[5 + r2], v[10 + r2], v[15 + r2]};
    if constexpr (MIDZERO) {
      const v2f x0 = v[r2], x3 = v[15 + r2];
      t[0] = x0 + x3;
      t[1] = INV ? add_mulmi(x0, x3) : sub_mulmi(x0, x3);
      t[2] = x0 - x3;
      t[3] = INV ? sub_mulmi(x0, x3) : add_mulmi(x0, x3);
    } else {
      fft_reg<4, INV>(t);
    }
    static_for<0, 4>([&](auto k1c) {
      constexpr int k1 = decltype(k1c)::value;
      constexpr int j = (r2 * k1) % 20;
      if constexpr (j == 0) {
        a[k1 * 5 + r2] = t[k1];
      } else if constexpr (j == 5) {
        a[k1 * 5 + r2] = INV ? mk(-t[k1].y, t[k1].x) : mk(t[k1].y, -t[k1].x);
      } else if constexpr (j == 10) {
        a[k1 * 5 + r2] = -t[k1];
      } else if constexpr (j == 15) {
        a[k1 * 5 + r2] = INV ? mk(t[k1].y, -t[k1].x) : mk(-t[k1].y, t[k1].x);
      } else {
        constexpr float c = cos20(j), sn = INV ? sin20(j) : -sin20(j);
        a[k1 * 5 + r2] = pk_fma(t[k1].yy, mk(-sn, c), t[k1].xx * mk(c, sn));  // t * (c + i sn)
      }
    });
  });
  static_for<0, 4>([&](auto k1c) {
    constexpr int k1 = decltype(k1c)::value;
    fft_reg5<INV>(a + k1 * 5);
    static_for<0, 5>([&](auto k2c) {
      constexpr int k2 = decltype(k2c)::value;
      v[k1 + 4 * k2] = a[k1 * 5 + k2];
    });
  });
}

}  // namespace fdoct
