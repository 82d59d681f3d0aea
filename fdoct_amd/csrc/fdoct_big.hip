// fdoct_big.hip -- the long-row path of the FD-OCT chain (gfx950): rows in HBM between the steps.
//
// cv::dft takes any length and so does zeropadrowwise (BscanFFT.cpp:211, 241, 1185).  The any-configuration kernel of
// fdoct_generic.hip keeps a row and its two DFT buffers in LDS, which ends near 8000 points (4000 when the length needs
// Bluestein), and its zero-pad stage wants lengths that factor into 2, 3 and 5.  What lies beyond -- e.g. 4096 samples
// upsampled x4 to 16384, or a 322-sample row (W/2 = 7 * 23) with the zero-pad on -- runs here: the same steps in the same
// f32 arithmetic (row mean in f64, float DFTs), each step a streaming kernel over all rows of a chunk with the rows held in
// global memory, the DFTs as Stockham passes of radix 8/4/2/5/3 over two ping-pong buffers (one kernel launch per pass)
// and, for lengths with a prime factor above 5, Bluestein's algorithm around two power-of-two transforms.  Full-length
// complex transforms, no half-length tricks: this is the fallback that makes every width acceptable, not a fast path
// (each pass moves the whole chunk through HBM: ~20 passes of 16 bytes per point for a 16384-point row).
//
//   big_pre       A2/A3: dark, row / frame normalisation, pi frame, background, row mean, window  -> y[row][W]
//   (M > 1)       A4: F = conj(IDFT_W(y)) / W (y is real), Hermitian re-packing with the Nyquist bin dropped and Im F[0]
//                 ignored as cv::dft(DFT_REAL_OUTPUT) reads it, IDFT of length M W, real part       -> yup[row][M W]
//   big_resample  A5/A6/A6': slope step and lambda -> k gather with the reference's indexing, phase -> z[row][N]
//   IDFT_N        A7
//   big_post      A8-A10: magnitude of the first D bins, average over the group's frames, epsilon, dB, DC mask
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fdoct_big.h"
#include "fdoct_fft_reg.h"

namespace fdoct {

namespace {

__device__ __forceinline__ float big_load_sample(const void* row, int dtype, int i) {
  switch (dtype) {
    case FDOCT_K_U8: return (float)static_cast<const uint8_t*>(row)[i];
    case FDOCT_K_U16: return (float)static_cast<const uint16_t*>(row)[i];
    default: return static_cast<const float*>(row)[i];
  }
}

template <typename T, typename OP>
__device__ __forceinline__ T big_block_reduce(T v, T* red, OP op) {
  for (int m = 32; m >= 1; m >>= 1) v = op(v, __shfl_xor(v, m, 64));
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  T r = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = op(r, red[i]);
  return r;
}

// A2/A3 of one input A-scan per workgroup (the same expressions, in the same order, as generic_kernel).
__global__ __launch_bounds__(256) void big_pre_kernel(const BigArgs a, float* y) {
  __shared__ double redd[4];
  __shared__ float redf[4];
  const int W = a.W, tid = threadIdx.x, nt = blockDim.x;
  const unsigned char* frames = static_cast<const unsigned char*>(a.frames);
  for (long long ir = blockIdx.x; ir < a.in_rows; ir += gridDim.x) {
    const long long f = ir / a.H;  // input frame of the chunk
    const int r = (int)(ir - f * a.H);
    const void* row = frames + ir * a.pitch_bytes;
    float* yr = y + ir * W;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < W; i += nt) {
      float x = big_load_sample(row, a.dtype, i);
      if (a.yd) x -= a.yd[(a.yd_2d ? (size_t)r * W : 0) + i];
      yr[i] = x;
      mn = fminf(mn, x);
      mx = fmaxf(mx, x);
    }
    // min-max normalisation (main:88-97, 1126-1129): the normalised sample as two floats, as in generic_kernel
    const bool norm_on = a.rowwisenormalize || a.minmax;
    float nmn = 0.f, nsc = 1.f;
    if (a.rowwisenormalize) {
      mn = big_block_reduce<float>(mn, redf, [](float p, float q) { return fminf(p, q); });
      mx = big_block_reduce<float>(mx, redf, [](float p, float q) { return fmaxf(p, q); });
      nmn = mn;
      nsc = (mx - mn > 2.220446049250313e-16f) ? 1.f / (mx - mn) : 0.f;
    } else if (a.minmax) {
      const float2 mmx = a.minmax[f];
      nmn = mmx.x;
      nsc = (mmx.y - mmx.x > 2.220446049250313e-16f) ? 1.f / (mmx.y - mmx.x) : 0.f;
    }
    // main:1132 through the host-side reciprocal (x / 0 = 0), rounded at the size of the deviation from c0, the row's middle
    // sample (a block-uniform estimate of the mean), as in generic_kernel
    __syncthreads();
    float c0;
    {
      const int im = W >> 1;
      float xm = yr[im];
      if (norm_on) xm = (xm - nmn) * nsc;
      if (a.yp) xm -= a.yp[(a.yp_2d ? (size_t)r * W : 0) + im];
      c0 = xm * a.ib[(a.ib_2d ? (size_t)r * W : 0) + im];
    }
    __syncthreads();
    double sum = 0.0;
    const float* lo_row = a.frames_lo ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.frames_lo) + ir * a.pitch_bytes) : nullptr;
    for (int i = tid; i < W; i += nt) {
      float x = yr[i], xlo = lo_row ? lo_row[i] : 0.f;   // (f64 frames: the samples' low words, as in generic_kernel)
      // (round 6, as in generic_kernel: the exact residuals of the dark, normalisation and pi differences join the low word)
      if (a.yd) {
        float e;
        (void)two_diff(big_load_sample(row, a.dtype, i), a.yd[(a.yd_2d ? (size_t)r * W : 0) + i], e);
        xlo += e;
      }
      if (norm_on) {
        float e;
        const float vm = two_diff(x, nmn, e);
        xlo += e;
        x = vm * nsc;
        xlo = fmaf(xlo, nsc, fmaf(vm, nsc, -x));
      }
      if (a.yp) {
        float e;
        x = two_diff(x, a.yp[(a.yp_2d ? (size_t)r * W : 0) + i], e);
        xlo += e;
      }
      const size_t bi = (a.ib_2d ? (size_t)r * W : 0) + i;
      x = fmaf(xlo, a.ib[bi], fmaf(x, a.il[bi], fmaf(x, a.ib[bi], -c0)));  // 1/yb = ib + il (fdoct_capi.cpp::reciprocal_words): nothing rounds at the size of the DC level
      yr[i] = x;
      sum += (double)x;
    }
    sum = big_block_reduce<double>(sum, redd, [](double p, double q) { return p + q; });
    const double mean = sum / (double)W;  // main:1138
    const float mh = (float)mean, ml = (float)(mean - (double)mh);
    for (int i = tid; i < W; i += nt) yr[i] = ((yr[i] - mh) - ml) * a.win[i];  // main:1139, 1142
    __syncthreads();
  }
}

// real rows -> complex rows of the same length (imaginary part 0)
__global__ void big_real_to_complex_kernel(const float* y, long long total, float2* z) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
    z[e] = make_float2(y[e], 0.f);
}

// One Stockham pass of radix R over every row of the chunk: butterfly j of a row takes src[j + r nb], multiplies by
// exp(+2 pi i r k / (Ns R)) (k = j mod Ns; tw[m] = exp(+2 pi i m / n)) and writes dst[(j div Ns) Ns R + k + r Ns].
template <int R>
__global__ __launch_bounds__(256) void big_fft_pass_kernel(const float2* src_, float2* dst_, long long rows, int n, int Ns,
                                                           const float2* tw_) {
  const v2f* tw = reinterpret_cast<const v2f*>(tw_);
  const int nb = n / R, twstep = nb / Ns;
  const long long total = rows * nb;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / nb;
    const int j = (int)(e - row * nb);
    const v2f* src = reinterpret_cast<const v2f*>(src_) + row * n;
    v2f* dst = reinterpret_cast<v2f*>(dst_) + row * n;
    const int q = j / Ns, k = j - q * Ns;
    v2f v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = src[j + r * nb];
    if (Ns > 1) {
      v2f w[R];
      w[1] = tw[k * twstep];
#pragma unroll
      for (int r = 2; r < R; r++) w[r] = (r & 1) ? cmul(w[r - 1], w[1]) : cmul(w[r / 2], w[r / 2]);
#pragma unroll
      for (int r = 1; r < R; r++) v[r] = cmul(v[r], w[r]);
    }
    if constexpr (R == 3)
      fft_reg3<true>(v);
    else if constexpr (R == 5)
      fft_reg5<true>(v);
    else
      fft_reg<R, true>(v);
    v2f* d = dst + (q * Ns * R + k);
#pragma unroll
    for (int r = 0; r < R; r++) d[r * Ns] = v[r];
  }
}

// A4's re-packing (main:215-241), for any width: fdoct_fft_reg.h::pad_source (shared with generic_kernel's full-length stage)
__device__ __forceinline__ int big_pad_source(int W, int n, int bandpass, int pos, bool* mirror) { return pad_source(W, n, bandpass, pos, mirror); }

// ---- several passes per launch: one group of a transform's Stockham passes with the data in LDS (fdoct_big.h) ----------
// value of element `pos` of row `row` as the group's loader sees it
template <int LOAD>
__device__ __forceinline__ v2f big_group_load(const BigGroup& a, long long row, int pos) {
  if constexpr (LOAD == BIG_LOAD_CPLX) {
    return reinterpret_cast<const v2f*>(a.src)[row * a.n + pos];
  } else if constexpr (LOAD == BIG_LOAD_REAL) {
    return mk(a.yr[row * a.n + pos], 0.f);
  } else if constexpr (LOAD == BIG_LOAD_PAD) {  // big_pad_kernel's expression, read on the fly
    const int W = a.W;
    bool mirror;
    const int ks = big_pad_source(W, a.n, a.bandpass, pos, &mirror);
    if (ks < 0) return mk(0.f, 0.f);
    const float2 s = a.src[row * W + ks];
    const float inv_w = 1.f / (float)W;
    const float fx = s.x * inv_w, fy = (ks == 0) ? 0.f : -s.y * inv_w;
    return mirror ? mk(fx, -fy) : mk(fx, fy);
  } else {  // big_resample_kernel's expression
    const int q = pos, N = a.n;
    float yl = 0.f;
    if (q >= 1 && q <= N - 2) {
      const int i = a.idx[q];
      // (s >= ylen: the column an odd width's upsampled row lacks -- M W - 1 columns under an even multiplier -- reads as 0)
      auto at = [&](int s) { return s >= a.ylen ? 0.f : (a.yc ? a.yc[row * a.ylen + s].x : a.yr[row * a.ylen + s]); };
      const float yi = at(i);
      const float slope = (i == 0) ? (at(1) - at(0)) : (yi - at(i - 1));
      yl = fmaf(a.g[i], slope, yi);
    }
    return a.phase ? mk(yl * a.phase[q].x, yl * a.phase[q].y) : mk(yl, 0.f);
  }
}

// one local pass of radix R over the tile: every butterfly's inputs are read before any output is written (in place).
// The butterflies' twiddles come from global memory (L2): their loads are issued first, so that they are in flight under
// the LDS reads and the barrier.
template <int R>
__device__ __forceinline__ void big_group_pass(v2f* lds, const BigGroup& a, int ts, int s0, int Ns, int log2ns, const v2f* tw) {
  const int TS = 1 << a.log2ts, TSP = TS + 1, Q = a.Q, nbl = Q / R, nbut = nbl << a.log2ts;
  constexpr int MAXI = (BIG_GROUP_TILE_VALUES / 256 + R - 1) / R;  // butterflies per thread (rounded up): a tile holds at most BIG_GROUP_TILE_VALUES values
  const int Nsg = a.P * Ns;                    // the pass's Ns in the whole transform
  const int twstep = (a.n / R) / Nsg;
  v2f v[MAXI][R], w1[MAXI];
  int dsto[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; i++) {
    const int b = threadIdx.x + 256 * i;
    if (b < nbut) {
      const int sl = b & (TS - 1), j = b >> a.log2ts;
      // (Ns is a power of two in all but the odd-radix passes: a shift then)
      const int q = log2ns >= 0 ? (j >> log2ns) : j / Ns, k = j - q * Ns;
      dsto[i] = (q * Ns * R + k) * TSP + sl;
      w1[i] = mk(1.f, 0.f);
      if (Nsg > 1 && sl < ts) {
        const int k1 = a.P > 1 ? (s0 + sl) % a.P : 0;
        w1[i] = tw[(long long)(k1 + a.P * k) * twstep];   // the butterfly's k in the whole transform
      }
#pragma unroll
      for (int r = 0; r < R; r++) v[i][r] = lds[(j + r * nbl) * TSP + sl];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MAXI; i++) {
    const int b = threadIdx.x + 256 * i;
    if (b < nbut) {
      if (Nsg > 1) {
        v2f w[R];
        w[1] = w1[i];
#pragma unroll
        for (int r = 2; r < R; r++) w[r] = (r & 1) ? cmul(w[r - 1], w[1]) : cmul(w[r / 2], w[r / 2]);
#pragma unroll
        for (int r = 1; r < R; r++) v[i][r] = cmul(v[i][r], w[r]);
      }
      if constexpr (R == 3)
        fft_reg3<true>(v[i]);
      else if constexpr (R == 5)
        fft_reg5<true>(v[i]);
      else
        fft_reg<R, true>(v[i]);
      v2f* d = lds + dsto[i];
#pragma unroll
      for (int r = 0; r < R; r++) d[r * Ns * TSP] = v[i][r];
    }
  }
  __syncthreads();
}

template <int LOAD>
__global__ __launch_bounds__(256, 6) void big_fft_group_kernel(const BigGroup a) {
  extern __shared__ __align__(16) unsigned char big_lds_raw[];
  v2f* lds = reinterpret_cast<v2f*>(big_lds_raw);
  const v2f* tw = reinterpret_cast<const v2f*>(a.tw);
  const int TS = 1 << a.log2ts, TSP = TS + 1, Q = a.Q, S = a.P * a.F;
  const int tiles_per_row = (S + TS - 1) >> a.log2ts;
  const long long tiles = a.rows * tiles_per_row;
  v2f* dst = reinterpret_cast<v2f*>(a.dst);
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long row = tile / tiles_per_row;
    const int s0 = (int)(tile - row * tiles_per_row) << a.log2ts;
    const int ts = S - s0 < TS ? S - s0 : TS;
    // element t of sub-problem s0 + sl sits at s0 + sl + S t: TS adjacent values per t
    for (int e = threadIdx.x; e < (Q << a.log2ts); e += 256) {
      const int sl = e & (TS - 1), t = e >> a.log2ts;
      lds[t * TSP + sl] = sl < ts ? big_group_load<LOAD>(a, row, s0 + sl + S * t) : mk(0.f, 0.f);
    }
    __syncthreads();
    int Ns = 1;
    for (int p = 0; p < a.npass; p++) {
      const int l2 = (Ns & (Ns - 1)) == 0 ? 31 - __builtin_clz((unsigned)Ns) : -1;
      switch (a.rad[p]) {
        case 8: big_group_pass<8>(lds, a, ts, s0, Ns, l2, tw); break;
        case 4: big_group_pass<4>(lds, a, ts, s0, Ns, l2, tw); break;
        case 2: big_group_pass<2>(lds, a, ts, s0, Ns, l2, tw); break;
        case 5: big_group_pass<5>(lds, a, ts, s0, Ns, l2, tw); break;
        default: big_group_pass<3>(lds, a, ts, s0, Ns, l2, tw); break;
      }
      Ns *= a.rad[p];
    }
    // result e of sub-problem s = k1 + P a goes to k1 + P (a Q + e)
    v2f* drow = dst + row * a.n;
    if (a.P == 1) {  // blocks of Q contiguous values per sub-problem: threads run along e
      for (int e = threadIdx.x; e < (Q << a.log2ts); e += 256) {
        const int sl = e / Q, el = e - sl * Q;
        const int pos = (s0 + sl) * Q + el;
        if (sl < ts && pos < a.out_limit) drow[pos] = lds[el * TSP + sl];
      }
    } else {         // TS adjacent values per e: threads run along the sub-problems
      for (int e = threadIdx.x; e < (Q << a.log2ts); e += 256) {
        const int sl = e & (TS - 1), el = e >> a.log2ts;
        const int s = s0 + sl, k1 = s % a.P, aa = s / a.P;
        const int pos = k1 + a.P * (aa * Q + el);
        if (sl < ts && pos < a.out_limit) drow[pos] = lds[el * TSP + sl];
      }
    }
    __syncthreads();
  }
}

// Bluestein, step 1: conj(x[m] c[m]) into a zero-padded row of length mb.  Only +i passes exist here, so the forward
// transform of the convolution is taken as conj(IDFT(conj u)): this kernel supplies the inner conj, big_conj_mul the outer.
__global__ void big_chirp_in_kernel(const float2* x, long long rows, int n, int mb, const float2* chirp, float2* out) {
  const long long total = rows * mb;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / mb;
    const int i = (int)(e - row * mb);
    float2 v = make_float2(0.f, 0.f);
    if (i < n) {
      const float2 s = x[row * n + i], c = chirp[i];
      v = make_float2(fmaf(-s.y, c.y, s.x * c.x), -fmaf(s.y, c.x, s.x * c.y));
    }
    out[e] = v;
  }
}
// conj (the forward transform of the convolution as conj(IDFT(conj .))) and, when bhat is given, the product with it
__global__ void big_conj_mul_kernel(float2* z, long long rows, int mb, const float2* bhat) {
  const long long total = rows * mb;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    float2 x = z[e];
    x.y = -x.y;
    if (bhat) {
      const float2 b = bhat[e % mb];
      x = make_float2(fmaf(-x.y, b.y, x.x * b.x), fmaf(x.y, b.x, x.x * b.y));
    }
    z[e] = x;
  }
}
// Bluestein, last step: X[k] = c[k] C[k], k < n, packed to rows of length n
__global__ void big_chirp_out_kernel(const float2* cbuf, long long rows, int n, int mb, const float2* chirp, float2* out) {
  const long long total = rows * n;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / n;
    const int k = (int)(e - row * n);
    const float2 x = cbuf[row * mb + k], c = chirp[k];
    out[e] = make_float2(fmaf(-x.y, c.y, x.x * c.x), fmaf(x.y, c.x, x.x * c.y));
  }
}

// A4, between the two transforms: spec = IDFT_W(y) (so F = conj(spec) / W), re-packed into the Hermitian spectrum of length
// M W that cv::dft(DFT_INVERSE | DFT_REAL_OUTPUT) reads: bins 0 <= k < W/2 and their mirrors, Im of bin 0 ignored, the
// Nyquist bin of the row dropped (fftshift put it on the negative side only); BscanDark's band-pass keeps 3 <= k < W/10.
__global__ void big_pad_kernel(const float2* spec, long long rows, int W, int MW, int bandpass, float2* z) {
  const long long total = rows * MW;   // (MW: the padded spectrum's length, big_pad_source's n)
  const float inv_w = 1.f / (float)W;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / MW;
    const int k = (int)(e - row * MW);
    bool mirror;
    const int ks = big_pad_source(W, MW, bandpass, k, &mirror);  // the source bin of the row's spectrum, or none
    float2 v = make_float2(0.f, 0.f);
    if (ks >= 0) {
      const float2 s = spec[row * W + ks];
      const float fx = s.x * inv_w, fy = (ks == 0) ? 0.f : -s.y * inv_w;  // F = conj(spec) / W
      v = mirror ? make_float2(fx, -fy) : make_float2(fx, fy);            // mirror: conj F
    }
    z[e] = v;
  }
}

// A5 / A6 / A6': data_ylin[q] = y[i] + fractionalk[i] (y[i] - y[i-1]), i = nearestkindex[q], q = 1 .. N-2 (0 elsewhere);
// the row is the real part of `yc` (upsampled rows, stride ylen complex) or `yr` (plain rows, stride ylen floats)
__global__ void big_resample_kernel(const float* yr, const float2* yc, long long rows, int ylen, int N, const int32_t* idx,
                                    const float* g, const float2* phase, float2* z) {
  const long long total = rows * N;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / N;
    const int q = (int)(e - row * N);
    float yl = 0.f;
    if (q >= 1 && q <= N - 2) {
      const int i = idx[q];
      auto at = [&](int s) { return s >= ylen ? 0.f : (yc ? yc[row * ylen + s].x : yr[row * ylen + s]); };  // (see big_group_load)
      const float yi = at(i);
      const float slope = (i == 0) ? (at(1) - at(0)) : (yi - at(i - 1));  // main:1153-1161
      yl = fmaf(g[i], slope, yi);                                          // main:1164-1173
    }
    z[e] = phase ? make_float2(yl * phase[q].x, yl * phase[q].y) : make_float2(yl, 0.f);
  }
}

// A8-A10
__global__ void big_post_kernel(const float2* X, const BigArgs a, float* out_mag, float* out_db) {
  const long long total = a.out_rows * a.D;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long o = e / a.D;
    const int b = (int)(e - o * a.D);
    const long long g = o / a.H;
    const int r = (int)(o - g * a.H);
    auto mean_mag = [&](int bin) {
      float acc = 0.f;
      for (int ai = 0; ai < a.A; ai++) {
        const float2 x = X[((g * a.A + ai) * a.H + r) * (long long)a.N + bin];
        const float m = sqrtf(fmaf(x.x, x.x, x.y * x.y));
        acc = (ai == 0) ? m : acc + m;
      }
      return fmaf(acc, a.inv_A, a.eps);
    };
    const float v = mean_mag(b);
    if (out_mag) out_mag[e] = v;
    if (out_db) out_db[e] = a.db_scale * log2f((a.dcmask && a.D > 4 && b < 2) ? mean_mag(4) : v);
  }
}

int grid_for(long long total, int per_block) {
  long long g = (total + per_block - 1) / per_block;
  if (g > 65535 * 16) g = 65535 * 16;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

hipError_t big_launch_pre(const BigArgs& a, float* y, hipStream_t st) {
  hipLaunchKernelGGL(big_pre_kernel, dim3((unsigned)(a.in_rows < 1048576 ? a.in_rows : 1048576)), dim3(256), 0, st, a, y);
  return hipGetLastError();
}
hipError_t big_launch_real_to_complex(const float* y, long long total, float2* z, hipStream_t st) {
  hipLaunchKernelGGL(big_real_to_complex_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, y, total, z);
  return hipGetLastError();
}
hipError_t big_launch_fft_pass(const float2* src, float2* dst, long long rows, int n, int radix, int Ns, const float2* tw, hipStream_t st) {
  const dim3 g(grid_for(rows * (n / radix), 256)), b(256);
  switch (radix) {
    case 8: hipLaunchKernelGGL(big_fft_pass_kernel<8>, g, b, 0, st, src, dst, rows, n, Ns, tw); break;
    case 4: hipLaunchKernelGGL(big_fft_pass_kernel<4>, g, b, 0, st, src, dst, rows, n, Ns, tw); break;
    case 2: hipLaunchKernelGGL(big_fft_pass_kernel<2>, g, b, 0, st, src, dst, rows, n, Ns, tw); break;
    case 5: hipLaunchKernelGGL(big_fft_pass_kernel<5>, g, b, 0, st, src, dst, rows, n, Ns, tw); break;
    case 3: hipLaunchKernelGGL(big_fft_pass_kernel<3>, g, b, 0, st, src, dst, rows, n, Ns, tw); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
size_t big_group_lds_bytes(const BigGroup& g) { return (size_t)g.Q * ((1u << g.log2ts) + 1) * sizeof(float2); }
hipError_t big_launch_fft_group(const BigGroup& g, hipStream_t st) {
  const int S = g.P * g.F, TS = 1 << g.log2ts;
  const long long tiles = g.rows * ((S + TS - 1) / TS);
  const dim3 grid((unsigned)(tiles < 65535LL * 16 ? (tiles < 1 ? 1 : tiles) : 65535LL * 16)), block(256);
  const size_t lds = big_group_lds_bytes(g);
  if (lds > 64 * 1024 || (g.Q << g.log2ts) > BIG_GROUP_TILE_VALUES) return hipErrorInvalidValue;
  switch (g.load) {
    case BIG_LOAD_CPLX: hipLaunchKernelGGL(big_fft_group_kernel<BIG_LOAD_CPLX>, grid, block, lds, st, g); break;
    case BIG_LOAD_REAL: hipLaunchKernelGGL(big_fft_group_kernel<BIG_LOAD_REAL>, grid, block, lds, st, g); break;
    case BIG_LOAD_PAD: hipLaunchKernelGGL(big_fft_group_kernel<BIG_LOAD_PAD>, grid, block, lds, st, g); break;
    case BIG_LOAD_RESAMPLE: hipLaunchKernelGGL(big_fft_group_kernel<BIG_LOAD_RESAMPLE>, grid, block, lds, st, g); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t big_launch_chirp_in(const float2* x, long long rows, int n, int mb, const float2* chirp, float2* out, hipStream_t st) {
  hipLaunchKernelGGL(big_chirp_in_kernel, dim3(grid_for(rows * mb, 256)), dim3(256), 0, st, x, rows, n, mb, chirp, out);
  return hipGetLastError();
}
hipError_t big_launch_conj_mul(float2* z, long long rows, int mb, const float2* bhat, hipStream_t st) {
  hipLaunchKernelGGL(big_conj_mul_kernel, dim3(grid_for(rows * mb, 256)), dim3(256), 0, st, z, rows, mb, bhat);
  return hipGetLastError();
}
hipError_t big_launch_chirp_out(const float2* cbuf, long long rows, int n, int mb, const float2* chirp, float2* out, hipStream_t st) {
  hipLaunchKernelGGL(big_chirp_out_kernel, dim3(grid_for(rows * n, 256)), dim3(256), 0, st, cbuf, rows, n, mb, chirp, out);
  return hipGetLastError();
}
hipError_t big_launch_pad(const float2* spec, long long rows, int W, int MW, int bandpass, float2* z, hipStream_t st) {
  hipLaunchKernelGGL(big_pad_kernel, dim3(grid_for(rows * MW, 256)), dim3(256), 0, st, spec, rows, W, MW, bandpass, z);
  return hipGetLastError();
}
hipError_t big_launch_resample(const float* yr, const float2* yc, long long rows, int ylen, int N, const int32_t* idx, const float* g,
                               const float2* phase, float2* z, hipStream_t st) {
  hipLaunchKernelGGL(big_resample_kernel, dim3(grid_for(rows * N, 256)), dim3(256), 0, st, yr, yc, rows, ylen, N, idx, g, phase, z);
  return hipGetLastError();
}
hipError_t big_launch_post(const float2* X, const BigArgs& a, float* out_mag, float* out_db, hipStream_t st) {
  hipLaunchKernelGGL(big_post_kernel, dim3(grid_for(a.out_rows * a.D, 256)), dim3(256), 0, st, X, a, out_mag, out_db);
  return hipGetLastError();
}

}  // namespace fdoct
