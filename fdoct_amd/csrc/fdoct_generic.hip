// fdoct_generic.hip -- the any-configuration kernels of the FD-OCT path (gfx950).
//
// fdoct_kernels.hip holds the specialised kernels (power-of-two N, M = 1, row widths that are a multiple
// of 8, D <= N/2).  Everything else the reference's block accepts runs here: any N = 2^a 3^b 5^c (the
// shipped ini uses 2560), the zero-pad spectral upsampling `zeropadrowwise` (increasefftpointsmultiplier
// M > 1, BscanFFT.cpp:180-245), any row width, numdisplaypoints up to N, every input type.  One workgroup
// owns one output A-scan at a time and keeps the whole row in LDS; the DFTs are mixed-radix Stockham passes
// (radix 4/2/3/5) over LDS ping-pong buffers with host-built twiddle tables.  Same arithmetic types as the
// specialised path (f32, row mean in f64); simpler and slower (no register-resident FFT, full complex DFT
// even for real rows), but it is the same math step for step, so the two paths agree to rounding.
//
// `smoothmovavg` (BscanFFT.cpp:247-304) is a separate elementwise pre-kernel here (movavg_kernel) that
// both paths share.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fdoct_kernels.h"

namespace fdoct {

namespace {

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
  return make_float2(fmaf(-a.y, b.y, a.x * b.x), fmaf(a.y, b.x, a.x * b.y));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by sgn*i
__device__ __forceinline__ float2 muli(float2 a, float sgn) { return make_float2(-sgn * a.y, sgn * a.x); }

// R-point DFT with exponent sign sgn (+1: the reference's DFT_INVERSE, -1: forward), R in {2,3,4,5}
__device__ __forceinline__ void dft_small(float2* v, int R, float sgn) {
  if (R == 2) {
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if (R == 4) {
    const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = muli(csub(v[1], v[3]), sgn);
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
  } else if (R == 3) {
    const float c = -0.5f, s = 0.86602540378443864676f * sgn;
    const float2 t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
    const float2 m = make_float2(v[0].x + c * t.x, v[0].y + c * t.y);
    const float2 r = make_float2(-s * d.y, s * d.x);  // i*s*d
    v[0] = cadd(v[0], t);
    v[1] = cadd(m, r);
    v[2] = csub(m, r);
  } else {  // R == 5
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f * sgn, s2 = 0.58778525229247312917f * sgn;
    const float2 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]), a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
    const float2 m1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    const float2 m2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    const float2 r1 = make_float2(-(s1 * b1.y + s2 * b2.y), s1 * b1.x + s2 * b2.x);  // i*(s1 b1 + s2 b2)
    const float2 r2 = make_float2(-(s2 * b1.y - s1 * b2.y), s2 * b1.x - s1 * b2.x);  // i*(s2 b1 - s1 b2)
    v[0] = cadd(v[0], cadd(a1, a2));
    v[1] = cadd(m1, r1);
    v[4] = csub(m1, r1);
    v[2] = cadd(m2, r2);
    v[3] = csub(m2, r2);
  }
}

// In-LDS mixed-radix Stockham DFT of length n.  src/dst are ping-pong buffers; returns the buffer that
// holds the result.  tw[j] = exp(+2*pi*i*j/n); sgn selects the exponent sign.
__device__ float2* fft_lds(float2* src, float2* dst, int n, const int* radices, int npass, const float2* tw, float sgn) {
  int Ns = 1;
  for (int p = 0; p < npass; p++) {
    const int R = radices[p];
    const int nb = n / R;
    const int twstep = n / (Ns * R);
    for (int j = threadIdx.x; j < nb; j += blockDim.x) {
      const int k = j % Ns;
      float2 v[5];
      for (int r = 0; r < R; r++) {
        float2 x = src[j + r * nb];
        if (r > 0 && k > 0) {
          float2 w = tw[r * k * twstep];
          w.y *= sgn;
          x = cmulf(x, w);
        }
        v[r] = x;
      }
      dft_small(v, R, sgn);
      const int j0 = (j / Ns) * Ns * R + k;
      for (int r = 0; r < R; r++) dst[j0 + r * Ns] = v[r];
    }
    __syncthreads();
    float2* t = src;
    src = dst;
    dst = t;
    Ns *= R;
  }
  return src;
}

template <typename T>
__device__ __forceinline__ T block_reduce(T v, T* red, T (*op)(T, T)) {
  // 256..1024 threads: wave shuffle then LDS
  for (int m = 32; m >= 1; m >>= 1) v = op(v, __shfl_xor(v, m, 64));
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  T r = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = op(r, red[i]);
  return r;
}
__device__ double op_addd(double a, double b) { return a + b; }
__device__ float op_minf(float a, float b) { return fminf(a, b); }
__device__ float op_maxf(float a, float b) { return fmaxf(a, b); }

__device__ __forceinline__ float load_sample(const void* row, int dtype, int i) {
  switch (dtype) {
    case FDOCT_K_U8: return (float)static_cast<const uint8_t*>(row)[i];
    case FDOCT_K_U16: return (float)static_cast<const uint16_t*>(row)[i];
    default: return static_cast<const float*>(row)[i];
  }
}

}  // namespace

// One workgroup per output A-scan (persistent: strides over rows).  See the file header.
__global__ __launch_bounds__(256) void generic_kernel(const GenericArgs a) {
  extern __shared__ __align__(16) unsigned char gsm[];
  const int W = a.W, M = a.M, MW = a.W * a.M, N = a.N, D = a.D, L = a.L;
  float* ybuf = reinterpret_cast<float*>(gsm);                  // [max(W, MW)] the row (then the upsampled row)
  float2* bufA = reinterpret_cast<float2*>(ybuf + a.ybuf_len);  // [L]
  float2* bufB = bufA + L;                                      // [L]
  __shared__ double redd[16];
  __shared__ float redf[16];
  __shared__ float bcast[2];
  const int tid = threadIdx.x, nt = blockDim.x;
  const unsigned char* frames = static_cast<const unsigned char*>(a.frames);

  for (long long o = blockIdx.x; o < a.total_out_rows; o += gridDim.x) {
    const long long g = o / a.H;
    const int r = (int)(o - g * a.H);
    float acc[GENERIC_MAX_BINS_PER_THREAD];
#pragma unroll
    for (int j = 0; j < GENERIC_MAX_BINS_PER_THREAD; j++) acc[j] = 0.f;

    for (int ai = 0; ai < a.A; ai++) {
      const long long in_frame = g * a.A + ai;
      const void* row = frames + (in_frame * a.H + r) * a.pitch_bytes;
      // ---- A2: dark, row / frame normalisation, pi frame, background
      float mn = INFINITY, mx = -INFINITY;
      for (int i = tid; i < W; i += nt) {
        float x = load_sample(row, a.dtype, i);
        if (a.yd) x -= a.yd[(a.yd_2d ? (size_t)r * W : 0) + i];
        ybuf[i] = x;
        mn = fminf(mn, x);
        mx = fmaxf(mx, x);
      }
      if (a.rowwisenormalize) {
        mn = block_reduce<float>(mn, redf, op_minf);
        mx = block_reduce<float>(mx, redf, op_maxf);
        const float sc = (mx - mn > 2.220446049250313e-16f) ? 1.f / (mx - mn) : 0.f;
        const float sh = -mn * sc;
        for (int i = tid; i < W; i += nt) ybuf[i] = fmaf(ybuf[i], sc, sh);
      }
      float nsc = 1.f, nsh = 0.f;
      if (a.minmax) {
        const float2 mmx = a.minmax[in_frame];
        nsc = (mmx.y - mmx.x > 2.220446049250313e-16f) ? 1.f / (mmx.y - mmx.x) : 0.f;
        nsh = -mmx.x * nsc;
      }
      double sum = 0.0;
      for (int i = tid; i < W; i += nt) {
        float x = ybuf[i];
        if (a.minmax) x = fmaf(x, nsc, nsh);
        if (a.yp) x -= a.yp[(a.yp_2d ? (size_t)r * W : 0) + i];
        x *= a.ib[(a.ib_2d ? (size_t)r * W : 0) + i];
        ybuf[i] = x;
        sum += (double)x;
      }
      // ---- A3: DC removal (mean in double), window
      sum = block_reduce<double>(sum, redd, op_addd);
      const double mean = sum / (double)W;
      const float mh = (float)mean, ml = (float)(mean - (double)mh);
      for (int i = tid; i < W; i += nt) ybuf[i] = ((ybuf[i] - mh) - ml) * a.win[i];
      __syncthreads();

      // ---- A4: zero-pad spectral upsampling (main:180-245), float DFTs as in the reference
      if (M > 1) {
        for (int i = tid; i < W; i += nt) bufA[i] = make_float2(ybuf[i], 0.f);
        __syncthreads();
        float2* F = fft_lds(bufA, bufB, W, a.rad_w, a.npass_w, a.tw_w, -1.f);  // forward
        float2* G = (F == bufA) ? bufB : bufA;
        const float inv_w = 1.f / (float)W;  // DFT_SCALE
        // the real-output inverse reads bins 0..n/2 only (Hermitian extension, imaginary part of bin 0
        // ignored), so of the shifted/padded spectrum only F[0 .. W/2-1] survive; the Nyquist bin is dropped
        for (int k = tid; k < MW; k += nt) {
          float2 v = make_float2(0.f, 0.f);
          if (k < W / 2) {
            v = make_float2(F[k].x * inv_w, k == 0 ? 0.f : F[k].y * inv_w);
          } else if (k > MW - W / 2) {
            const float2 f = F[MW - k];
            v = make_float2(f.x * inv_w, -f.y * inv_w);
          }
          G[k] = v;
        }
        __syncthreads();
        float2* Y = fft_lds(G, (G == bufA) ? bufB : bufA, MW, a.rad_mw, a.npass_mw, a.tw_mw, 1.f);
        for (int i = tid; i < MW; i += nt) ybuf[i] = Y[i].x;
        __syncthreads();
      }

      // ---- A5: lambda -> k resample with the reference's indexing (main:1151-1177), A6/A6'
      for (int q = tid; q < N; q += nt) {
        float yl = 0.f;
        if (q >= 1 && q <= N - 2) {
          const int i = a.idx[q];
          const float yi = ybuf[i];
          const float slope = (i == 0) ? (ybuf[1] - ybuf[0]) : (yi - ybuf[i - 1]);
          yl = fmaf(a.g[i], slope, yi);
        }
        bufA[q] = a.phase ? make_float2(yl * a.phase[q].x, yl * a.phase[q].y) : make_float2(yl, 0.f);
      }
      __syncthreads();
      // ---- A7: N-point inverse DFT (unscaled), A8: magnitude of the first D bins
      const float2* X = fft_lds(bufA, bufB, N, a.rad_n, a.npass_n, a.tw_n, 1.f);
#pragma unroll
      for (int j = 0; j < GENERIC_MAX_BINS_PER_THREAD; j++) {
        const int b = tid + j * nt;
        if (b < D) {
          const float2 x = X[b];
          acc[j] += sqrtf(fmaf(x.x, x.x, x.y * x.y));
        }
      }
      __syncthreads();
    }

    // ---- A9/A10
    float* om = a.out_mag ? a.out_mag + (size_t)o * D : nullptr;
    float* od = a.out_db ? a.out_db + (size_t)o * D : nullptr;
    float db4 = 0.f;
    if (od && a.dcmask && D > 4) {
      if (tid == 4) bcast[0] = a.db_scale * log2f(fmaf(acc[0], a.inv_A, a.eps));
      __syncthreads();
      db4 = bcast[0];
    }
#pragma unroll
    for (int j = 0; j < GENERIC_MAX_BINS_PER_THREAD; j++) {
      const int b = tid + j * nt;
      if (b < D) {
        const float v = fmaf(acc[j], a.inv_A, a.eps);
        if (om) om[b] = v;
        if (od) od[b] = (a.dcmask && D > 4 && b < 2) ? db4 : a.db_scale * log2f(v);
      }
    }
    __syncthreads();
  }
}

// smoothmovavg (main:247-304): (2n+1) taps, taps outside the row replaced by the centre sample, centre
// counted twice, divisor 2(n+1).  Any input type -> packed f32 frames.
__global__ void movavg_kernel(const void* frames, int dtype, long long pitch_bytes, int W, long long rows, int n,
                              float* out) {
  const long long total = rows * W;
  const float inv = 1.f / (2.f * (float)(n + 1));
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / W;
    const int j = (int)(e - r * W);
    const void* row = static_cast<const unsigned char*>(frames) + r * pitch_bytes;
    const float c = load_sample(row, dtype, j);
    float s = c;  // the extra centre weight
    for (int k = -n; k <= n; k++) {
      const int jj = j + k;
      s += (jj >= 0 && jj < W) ? load_sample(row, dtype, jj) : c;
    }
    out[e] = s * inv;
  }
}

hipError_t launch_generic(const GenericArgs& a, int grid, size_t lds, hipStream_t st) {
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(generic_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_set = lds;
  }
  hipLaunchKernelGGL(generic_kernel, dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_movavg(const void* frames, int dtype, long long pitch_bytes, int W, long long rows, int n, float* out,
                         hipStream_t st) {
  hipLaunchKernelGGL(movavg_kernel, dim3(2048), dim3(256), 0, st, frames, dtype, pitch_bytes, W, rows, n, out);
  return hipGetLastError();
}

}  // namespace fdoct

// ------------------------------------------------------------------------------------------------
// Frame-source tail: cv::medianBlur(mraw, m, n) (BscanFFT.cpp:953-956) and the
// software binning cv::resize(..., 1/binx, 1/biny, INTER_AREA) (BscanFFT.cpp:958,
// BscanFFTspinjnt.cpp:1553) on the integer camera samples, before the path proper.
namespace fdoct {

// n x n median (n odd, <= 7), BORDER_REPLICATE as cv::medianBlur.
template <typename T>
__global__ void median_kernel(const T* in, long long in_pitch, T* out, long long out_pitch, int w, int h, int n, int nframes) {
  const long long total = (long long)nframes * h * w;
  const int r = n / 2;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(e % w);
    const long long fy = e / w;  // frame*h + y
    const int y = (int)(fy % h);
    const long long f = fy / h;
    unsigned v[49];
    int c = 0;
    for (int dy = -r; dy <= r; dy++) {
      const int yy = min(max(y + dy, 0), h - 1);
      const T* row = reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(in) + (f * h + yy) * in_pitch);
      for (int dx = -r; dx <= r; dx++) v[c++] = row[min(max(x + dx, 0), w - 1)];
    }
    // partial selection up to the middle element
    const int mid = (n * n) / 2;
    for (int i = 0; i <= mid; i++) {
      int mi = i;
      for (int j = i + 1; j < n * n; j++)
        if (v[j] < v[mi]) mi = j;
      const unsigned t = v[i];
      v[i] = v[mi];
      v[mi] = t;
    }
    reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(out) + fy * out_pitch)[x] = (T)v[mid];
  }
}

// box mean over binx x biny, rounded to the sample type as cv::resize(INTER_AREA) does for integer
// factors: (s + 2) >> 2 for 2x2 (the vectorised 8u/16u path), round-half-even of s * (1/area) otherwise.
template <typename T>
__global__ void bin_kernel(const T* in, long long in_pitch, T* out, long long out_pitch, int ow, int oh, int binx, int biny,
                           int nframes) {
  const long long total = (long long)nframes * oh * ow;
  const float scale = 1.f / (float)(binx * biny);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(e % ow);
    const long long fy = e / ow;
    const int y = (int)(fy % oh);
    const long long f = fy / oh;
    unsigned s = 0;
    for (int dy = 0; dy < biny; dy++) {
      const T* row = reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(in) +
                                                (f * (long long)oh * biny + (long long)y * biny + dy) * in_pitch);
      for (int dx = 0; dx < binx; dx++) s += row[x * binx + dx];
    }
    unsigned o;
    if (binx == 2 && biny == 2)
      o = (s + 2) >> 2;
    else
      o = (unsigned)rintf((float)s * scale);
    reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(out) + fy * out_pitch)[x] = (T)o;
  }
}

hipError_t launch_median(const void* in, long long in_pitch, void* out, long long out_pitch, int dtype, int w, int h, int n,
                         int nframes, hipStream_t st) {
  if (dtype == FDOCT_K_U8)
    hipLaunchKernelGGL(median_kernel<uint8_t>, dim3(4096), dim3(256), 0, st, static_cast<const uint8_t*>(in), in_pitch,
                       static_cast<uint8_t*>(out), out_pitch, w, h, n, nframes);
  else if (dtype == FDOCT_K_U16)
    hipLaunchKernelGGL(median_kernel<uint16_t>, dim3(4096), dim3(256), 0, st, static_cast<const uint16_t*>(in), in_pitch,
                       static_cast<uint16_t*>(out), out_pitch, w, h, n, nframes);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_bin(const void* in, long long in_pitch, void* out, long long out_pitch, int dtype, int ow, int oh, int binx,
                      int biny, int nframes, hipStream_t st) {
  if (dtype == FDOCT_K_U8)
    hipLaunchKernelGGL(bin_kernel<uint8_t>, dim3(4096), dim3(256), 0, st, static_cast<const uint8_t*>(in), in_pitch,
                       static_cast<uint8_t*>(out), out_pitch, ow, oh, binx, biny, nframes);
  else if (dtype == FDOCT_K_U16)
    hipLaunchKernelGGL(bin_kernel<uint16_t>, dim3(4096), dim3(256), 0, st, static_cast<const uint16_t*>(in), in_pitch,
                       static_cast<uint16_t*>(out), out_pitch, ow, oh, binx, biny, nframes);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace fdoct
