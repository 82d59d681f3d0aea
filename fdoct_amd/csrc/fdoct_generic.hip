// fdoct_generic.hip -- the any-configuration kernels of the FD-OCT path (gfx950).
//
// fdoct_kernels.hip holds the specialised kernels (power-of-two N, M = 1, row widths that are a multiple
// of 8, D <= N/2).  Everything else the reference's block accepts runs here: any N = 2^a 3^b 5^c (the
// shipped ini uses 2560), the zero-pad spectral upsampling `zeropadrowwise` (increasefftpointsmultiplier
// M > 1, BscanFFT.cpp:180-245), any row width, numdisplaypoints up to N, every input type.  One workgroup
// (256, 512 or 1024 threads, by how many rows a CU's LDS holds) owns one output A-scan at a time and keeps the whole row in
// LDS; the DFTs are mixed-radix Stockham passes (radix 16/8/4/2/5/3, butterflies in registers) over LDS ping-pong buffers --
// or in place in ONE buffer for rows of 9000 ... 16384 complex points -- with host-built twiddle tables, real rows at half
// length.  Same arithmetic types as the specialised path (f32; row mean in f64 as in its any-option kernels -- the fast-path
// ones carry it as two floats); simpler and slower (no register-resident FFT), but the same math step for step, so the two
// paths agree to rounding.
//
// `smoothmovavg` (BscanFFT.cpp:247-304) is a separate elementwise pre-kernel here (movavg_kernel: exact tap sums, the
// divisor is folded into the chain's planes on the host) that every path shares.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "fdoct_fft_reg.h"
#include "fdoct_kernels.h"

namespace fdoct {

namespace {

// One Stockham pass of radix R over the LDS ping-pong buffers: butterfly j takes src[j + r*nb], multiplies by
// exp(+-2*pi*i*r*k/(Ns*R)) (k = j mod Ns; table tw[m] = exp(+2*pi*i*m/n)), transforms in registers and writes
// dst[(j div Ns)*Ns*R + k + r*Ns].  j div Ns by multiplication: magic = ceil(2^32/Ns) is exact for j, Ns < 2^16.
template <int R, bool INV>
__device__ __forceinline__ void fft_pass(const v2f* src, v2f* dst, int n, int Ns, unsigned magic, const v2f* tw) {
  const int nb = n / R;
  const int twstep = nb / Ns;
  for (int j = threadIdx.x; j < nb; j += blockDim.x) {
    int q = j, k = 0;
    if (Ns > 1) {
      q = (int)__umulhi((unsigned)j, magic);
      k = j - q * Ns;
    }
    v2f v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = src[j + r * nb];
    if (Ns > 1) {
      // w^r, r = 1..R-1, from ONE table read: powers by repeated squaring / products (depth <= 4 multiplies,
      // a few ulp), instead of R-1 scattered reads of the table in global memory
      v2f w[R];
      w[1] = tw[k * twstep];
      if (!INV) w[1].y = -w[1].y;
#pragma unroll
      for (int r = 2; r < R; r++) w[r] = (r & 1) ? cmul(w[r - 1], w[1]) : cmul(w[r / 2], w[r / 2]);
#pragma unroll
      for (int r = 1; r < R; r++) v[r] = cmul(v[r], w[r]);
    }
    if constexpr (R == 3)
      fft_reg3<INV>(v);
    else if constexpr (R == 5)
      fft_reg5<INV>(v);
    else
      fft_reg<R, INV>(v);
    v2f* d = dst + (q * Ns * R + k);
#pragma unroll
    for (int r = 0; r < R; r++) d[r * Ns] = v[r];
  }
}

// The same pass IN PLACE (one buffer): every butterfly's inputs are read into registers, a barrier, then the outputs are
// written -- for rows whose two ping-pong buffers do not fit the LDS but one does (the caller's workgroup is large enough that
// a thread holds at most GENERIC_IP_VALUES values: n <= GENERIC_IP_VALUES / R * R * blockDim.x).
constexpr int GENERIC_IP_VALUES = 16;
template <int R, bool INV>
__device__ __forceinline__ void fft_pass_ip(v2f* buf, int n, int Ns, unsigned magic, const v2f* tw, int tid) {
  constexpr int MAXB = GENERIC_IP_VALUES / R;
  constexpr int nt = 1024;
  const int nb = n / R, twstep = nb / Ns;
  v2f v[MAXB][R];
#pragma unroll
  for (int i = 0; i < MAXB; i++) {
    const int j = tid + i * nt;
    if (j < nb) {
#pragma unroll
      for (int r = 0; r < R; r++) v[i][r] = buf[j + r * nb];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MAXB; i++) {
    const int j = tid + i * nt;
    if (j < nb) {
      int q = j, k = 0;
      if (Ns > 1) {
        q = (int)__umulhi((unsigned)j, magic);
        k = j - q * Ns;
        v2f w[R];
        w[1] = tw[k * twstep];
        if (!INV) w[1].y = -w[1].y;
#pragma unroll
        for (int r = 2; r < R; r++) w[r] = (r & 1) ? cmul(w[r - 1], w[1]) : cmul(w[r / 2], w[r / 2]);
#pragma unroll
        for (int r = 1; r < R; r++) v[i][r] = cmul(v[i][r], w[r]);
      }
      if constexpr (R == 3)
        fft_reg3<INV>(v[i]);
      else if constexpr (R == 5)
        fft_reg5<INV>(v[i]);
      else
        fft_reg<R, INV>(v[i]);
      v2f* d = buf + (q * Ns * R + k);
#pragma unroll
      for (int r = 0; r < R; r++) d[r * Ns] = v[i][r];
    }
  }
}

// In-LDS mixed-radix Stockham DFT of length n (radices 16/8/4/2/5/3, butterflies in registers).  src/dst are
// ping-pong buffers; returns the buffer that holds the result.  INV: exponent +i (the reference's DFT_INVERSE).
// IP: src == dst, the passes run in place (fft_pass_ip).
template <bool INV, bool IP = false, bool R16 = false>
__device__ float2* fft_lds(float2* src_, float2* dst_, int n, const int* radices, const unsigned* magics, int npass,
                           const float2* tw_, int tid = 0) {
  v2f* src = reinterpret_cast<v2f*>(src_);
  v2f* dst = reinterpret_cast<v2f*>(dst_);
  const v2f* tw = reinterpret_cast<const v2f*>(tw_);
  int Ns = 1;
  if constexpr (IP) {
    // (Tried: the buffer padded by one value per 32 between the passes, against the bank conflicts of the radix-strided stores of
    // the early passes -- 4.2 -> 3.4e6 A-scans/s on 4096 x8 -> 32768: an address per element instead of one base per butterfly.)
    for (int p = 0; p < npass; p++) {
      const int R = radices[p];
      const unsigned magic = magics[p];
      switch (R) {
        case 16: fft_pass_ip<16, INV>(src, n, Ns, magic, tw, tid); break;
        case 8: fft_pass_ip<8, INV>(src, n, Ns, magic, tw, tid); break;
        case 4: fft_pass_ip<4, INV>(src, n, Ns, magic, tw, tid); break;
        case 2: fft_pass_ip<2, INV>(src, n, Ns, magic, tw, tid); break;
        case 5: fft_pass_ip<5, INV>(src, n, Ns, magic, tw, tid); break;
        default: fft_pass_ip<3, INV>(src, n, Ns, magic, tw, tid); break;
      }
      __syncthreads();
      Ns *= R;
    }
    return src_;
  }
  for (int p = 0; p < npass; p++) {
    const int R = radices[p];
    const unsigned magic = magics[p];
    switch (R) {
      case 16:  // (the 1024-thread kernels: 128 registers per thread)
        if constexpr (R16 || GENERIC_MAX_RADIX >= 16) fft_pass<16, INV>(src, dst, n, Ns, magic, tw);
        break;
      case 8: fft_pass<8, INV>(src, dst, n, Ns, magic, tw); break;
      case 4: fft_pass<4, INV>(src, dst, n, Ns, magic, tw); break;
      case 2: fft_pass<2, INV>(src, dst, n, Ns, magic, tw); break;
      case 5: fft_pass<5, INV>(src, dst, n, Ns, magic, tw); break;
      default: fft_pass<3, INV>(src, dst, n, Ns, magic, tw); break;
    }
    __syncthreads();
    v2f* t = src;
    src = dst;
    dst = t;
    Ns *= R;
  }
  return reinterpret_cast<float2*>(src);
}

// The +i DFT of length n for lengths with a prime factor above 5 (cv::dft takes any length, main:1185): Bluestein.
//   X[k] = c[k] * sum_m (x[m] c[m]) * conj(c[k - m]),  c[m] = e^(+i pi m^2 / n)
// i.e. a circular convolution of length Mb >= 2n - 1 done with two power-of-two DFTs; bhat (host, double precision)
// is the transformed kernel with the 1/Mb of the unscaled inverse folded in.  x in `fin` (n values), both buffers
// hold Mb values; returns the buffer whose first n entries are X.
__device__ float2* bluestein_inverse(float2* fin, float2* fout, int n, int Mb, const float2* chirp, const float2* bhat,
                                     const int* rad, const unsigned* mag, int npass, const float2* tw) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < Mb; i += nt) {
    float2 v = make_float2(0.f, 0.f);
    if (i < n) {
      const float2 x = fin[i], c = chirp[i];
      v = make_float2(fmaf(-x.y, c.y, x.x * c.x), fmaf(x.y, c.x, x.x * c.y));
    }
    fin[i] = v;
  }
  __syncthreads();
  float2* A = fft_lds<false>(fin, fout, Mb, rad, mag, npass, tw);
  for (int i = tid; i < Mb; i += nt) {
    const float2 x = A[i], b = bhat[i];
    A[i] = make_float2(fmaf(-x.y, b.y, x.x * b.x), fmaf(x.y, b.x, x.x * b.y));
  }
  __syncthreads();
  float2* C = fft_lds<true>(A, (A == fin) ? fout : fin, Mb, rad, mag, npass, tw);
  for (int k = tid; k < n; k += nt) {
    const float2 x = C[k], c = chirp[k];
    C[k] = make_float2(fmaf(-x.y, c.y, x.x * c.x), fmaf(x.y, c.x, x.x * c.y));
  }
  __syncthreads();
  return C;
}
__device__ float2* bluestein_inverse(float2* fin, float2* fout, int n, const GenericArgs& a) {
  return bluestein_inverse(fin, fout, n, a.blu_m, a.blu_chirp, a.blu_bhat, a.rad_blu, a.mag_blu, a.npass_blu, a.tw_blu);
}
// the +i transform of any length over the two ping-pong buffers (the full-length zero-pad stage): Stockham passes or Bluestein
__device__ float2* dft_any_inverse(float2* fin, float2* fout, const GenericDft& p) {
  if (p.blu_m) return bluestein_inverse(fin, fout, p.n, p.blu_m, p.chirp, p.bhat, p.rad, p.mag, p.npass, p.tw);
  return fft_lds<true>(fin, fout, p.n, p.rad, p.mag, p.npass, p.tw);
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
// NT threads per workgroup, MINB workgroups per CU the register budget is cut for: 256 x 6 where the LDS holds three or more
// rows per CU; long rows, of which it holds two or one (4096 samples upsampled x4: 152 KB), get 512 x 2 / 1024 x 1 -- the row's
// loops all stride by blockDim.x, and a CU with one 256-thread workgroup is one wave per SIMD waiting on its own barriers.
// IP: ONE DFT buffer instead of two (rows of 8000 ... 16000 complex points: 4096 samples upsampled x8): every step that would
// read one buffer and write the other reads its inputs into registers, meets at a barrier, then writes.
template <int NT, int MINB, bool IP = false>
__global__ __launch_bounds__(NT, MINB) void generic_kernel(const GenericArgs a) {
  extern __shared__ __align__(16) unsigned char gsm[];
  const int W = a.W, M = a.M, MW = a.W * a.M, N = a.N, D = a.D, L = a.L;
  float* ybuf = reinterpret_cast<float*>(gsm);                  // [W] the row (the upsampled row lives in a DFT buffer)
  float2* bufA = reinterpret_cast<float2*>(ybuf + a.ybuf_len);  // [L]
  float2* bufB = IP ? bufA : bufA + L;                          // [L] (IP: the same buffer)
  float* accbuf = reinterpret_cast<float*>(bufB + L);  // [D] magnitudes summed over the averaged frames
  __shared__ double redd[16];
  __shared__ float redf[16];
  __shared__ float bcast[2];
  constexpr bool R16 = NT == 1024;  // radix-16 passes where a thread has 128 registers (the host plans them for these kernels only)
  const int tid0 = threadIdx.x;
  const int nt = IP ? NT : (int)blockDim.x;
  const unsigned char* frames = static_cast<const unsigned char*>(a.frames);

  // Rows are CLAIMED (round 6, as fused_kernel and wave_kernel do): a workgroup's first row is its index, every further one comes
  // from a launch-wide counter in global memory (zeroed by the host in front of the launch; null: the static stride of rounds
  // 1-5).  With a static deal the workgroups a CU favours finish their share early and leave the CU short of rows in flight
  // for the rest of the launch.  Thread 0 fetches the next row at the TOP of a row (the atomic's latency hides under the row's
  // work) into one of two LDS slots; everybody reads it behind the row's last barrier.
  __shared__ long long next_row[2];
  int parity = 0;
  for (long long o = blockIdx.x; o < a.total_out_rows;) {
    if (a.row_ticket && threadIdx.x == 0) next_row[parity] = (long long)gridDim.x + (long long)atomicAdd(a.row_ticket, 1u);
    const long long g = o / a.H;
    const int r = (int)(o - g * a.H);

    for (int ai = 0; ai < a.A; ai++) {
      // The thread index is re-read per row behind an empty asm, so that the index arithmetic of the unrolled pass bodies is
      // computed where it is used instead of once in front of the row loop and kept in spilled registers: 380 of them in the
      // one-buffer kernel (2.5 -> 4.2e6 A-scans/s on 4096 x8 -> 32768), 16 in the 256-thread one (+2 ... 6 % on every shape)
      int tid = tid0;
      asm volatile("" : "+v"(tid));
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
      // Min-max normalisation to [0, 1] (main:88-97, 1126-1129; sim:845), row-wise or of the whole frame (from the pre-pass):
      // p = (x - min) * scale.  The product is not a float, and rounding it would be a rounding at the size of the DC level
      // (3e-8 of full scale, random from sample to sample: above the tolerance once the fringes are weaker than 0.1 % of it),
      // so it is carried as TWO floats into the division: p_hi = fl((x - min) scale), p_lo = fma(x - min, scale, -p_hi), the
      // exact residual; x - min is exact for the camera's integer samples.
      const bool norm_on = a.rowwisenormalize || a.minmax;
      float nmn = 0.f, nsc = 1.f;
      if (a.rowwisenormalize) {
        mn = block_reduce<float>(mn, redf, op_minf);
        mx = block_reduce<float>(mx, redf, op_maxf);
        nmn = mn;
        nsc = (mx - mn > 2.220446049250313e-16f) ? 1.f / (mx - mn) : 0.f;
      } else if (a.minmax) {
        const float2 mmx = a.minmax[in_frame];
        nmn = mmx.x;
        nsc = (mmx.y - mmx.x > 2.220446049250313e-16f) ? 1.f / (mmx.y - mmx.x) : 0.f;
      }
      // x = (y - yp) / yb (main:1132) with no DC-sized rounding: c0, the value of the row's middle sample, is a block-uniform
      // estimate of the row mean; d = fma(y - yp, 1/yb, -c0) is the exact product minus c0, rounded at the size of the
      // deviation from it, and x - mean = d - mean(d) (as the fast path of fdoct_kernels.hip does).  1/yb is two floats,
      // ib + il (fdoct_capi.cpp::reciprocal_words): the second fma adds what the f32 reciprocal alone leaves out -- up to
      // 6e-8 of the quotient, a DC-sized fixed pattern -- again rounded at the size of the deviation.
      __syncthreads();  // every thread reads the middle sample
      float c0;
      {
        const int im = W >> 1;
        float xm = ybuf[im];
        if (norm_on) xm = (xm - nmn) * nsc;
        if (a.yp) xm -= a.yp[(a.yp_2d ? (size_t)r * W : 0) + im];
        c0 = xm * a.ib[(a.ib_2d ? (size_t)r * W : 0) + im];
      }
      __syncthreads();  // ... before anyone overwrites it
      double sum = 0.0;
      // BscanDark's band-pass (dark:218-236) keeps 3 <= k < W/10 of the row's spectrum: whatever the fringes put elsewhere is
      // blanked, and what is displayed is the little the window leaks into those few bins -- every float rounding in front of
      // the blanking is a rounding at the size of the WHOLE row (the f32 restatement itself sits up to 3 x the tolerance from
      // the chain in double on such rows).  With the band-pass on, the row is therefore formed in double (xd, in the first DFT
      // buffer) and the kept bins are evaluated directly in double below, instead of the float forward transform.
      const bool bp_direct = a.bandpass && M > 1;
      double* xd = reinterpret_cast<double*>(bufA);  // [W] (bufA holds L >= M W / 2 >= W complex floats)
      double sumd = 0.0;
      // (frames handed over as doubles, main:987: the samples' low words ride along like the normalisation's)
      const float* lo_row = a.frames_lo ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.frames_lo) + (in_frame * a.H + r) * a.pitch_bytes) : nullptr;
      for (int i = tid; i < W; i += nt) {
        float x = ybuf[i], xlo = lo_row ? lo_row[i] : 0.f;
        // Round 6: NO subtraction on the way rounds at the size of the DC level.  The sample less a non-integer dark frame
        // (dark:1269), less the normalisation's minimum, less a pi frame (main:1132) are f32 differences of DC-sized numbers;
        // each one's exact residual (two_diff) joins the low word, which enters the division like the normalisation's.
        if (a.yd) {
          float e;
          (void)two_diff(load_sample(row, a.dtype, i), a.yd[(a.yd_2d ? (size_t)r * W : 0) + i], e);   // x (= ybuf[i]) is that difference, rounded
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
        if (bp_direct) {
          double v = (double)load_sample(row, a.dtype, i) + (lo_row ? (double)lo_row[i] : 0.0);
          if (a.yd) v -= (double)a.yd[(a.yd_2d ? (size_t)r * W : 0) + i] + (double)a.yd_lo[(a.yd_2d ? (size_t)r * W : 0) + i];
          if (norm_on) v = (v - (double)nmn) * (double)nsc;
          if (a.yp) v -= (double)a.yp[(a.yp_2d ? (size_t)r * W : 0) + i] + (double)a.yp_lo[(a.yp_2d ? (size_t)r * W : 0) + i];
          v *= (double)a.ib[bi] + (double)a.il[bi];
          xd[i] = v;
          sumd += v;
        }
        x = fmaf(xlo, a.ib[bi], fmaf(x, a.il[bi], fmaf(x, a.ib[bi], -c0)));
        ybuf[i] = x;
        sum += (double)x;
      }
      // ---- A3: DC removal (mean of the deviations in double), window
      sum = block_reduce<double>(sum, redd, op_addd);
      const double mean = sum / (double)W;
      const float mh = (float)mean, ml = (float)(mean - (double)mh);
      for (int i = tid; i < W; i += nt) ybuf[i] = ((ybuf[i] - mh) - ml) * a.win[i];
      __syncthreads();
      // band-pass: X[j] = F[k] / W (DFT_SCALE) of the kept bins k = 3 + j, j < KB -- and, for an odd width, of the stray column
      // k = W - 1 the blanking spares (pad_source) -- one thread per bin (DftBinF64, fdoct_fft_reg.h), every lane of a wave reading
      // the same xd[m].  X takes the place of the float row.
      const int KB = W / 10 - 3 > 0 ? W / 10 - 3 : 0, NBIN = KB + (W & 1);
      float2* const Xbp = reinterpret_cast<float2*>(ybuf);  // [NBIN] (NBIN <= W / 10 - 2: inside the row's W floats)
      if (bp_direct) {
        sumd = block_reduce<double>(sumd, redd, op_addd);
        const double meand = sumd / (double)W;
        for (int i = tid; i < W; i += nt) xd[i] = (xd[i] - meand) * ((double)a.win[i] + (double)a.win_lo[i]);  // main:1138-1142 in double
        __syncthreads();
        const double inv_wd = 1.0 / (double)W;
        for (int j0 = 0; j0 < NBIN; j0 += nt) {
          const int j = j0 + tid;
          constexpr int T = 8;
          DftBinF64<T> bin;
          bin.init(j < KB ? 3 + j : W - 1, 0, W);
          if (j < NBIN) {
            for (int m = 0; m < W; m += T) {
              double x[T];
#pragma unroll
              for (int t = 0; t < T; t++) x[t] = m + t < W ? xd[m + t] : 0.0;
              bin.chunk(x);
            }
            Xbp[j] = make_float2((float)(bin.ar * inv_wd), (float)(bin.ai * inv_wd));
          }
        }
        __syncthreads();
      }

      // ---- A4: zero-pad spectral upsampling (main:180-245), float DFTs as in the reference.
      // zeropadrowwise = forward DFT of the real row (/W), spectrum re-packed Hermitian with the Nyquist bin dropped
      // and the imaginary part of bin 0 ignored (cv::dft DFT_REAL_OUTPUT reads bins 0..n/2 only), inverse DFT of length
      // M*W, real part.  Real row in, real row out: both transforms run at HALF length --
      //   forward: z[n] = y[2n] + i*y[2n+1], Zf = DFT_{W/2}(z), F[k] = ((Zf[k] + conj Zf[W/2-k]) - i*e^(-2*pi*i*k/W)*(Zf[k] - conj Zf[W/2-k]))/2
      //   inverse: with L = M*W and Hermitian X, Z[k] = (X[k] + conj X[L/2-k]) + i*w^k*(X[k] - conj X[L/2-k]), w = e^(+2*pi*i/L);
      //            IDFT_{L/2}(Z)[n] = x[2n] + i*x[2n+1]: the complex result buffer read as floats IS the upsampled row.
      // X is F/W on k < W/2 (bin 0 real) and zero up to L/2, so Z has a low band X[k]*(1 + i*w^k), k < W/2, and a high band
      // Z[L/2-k] = conj(X[k])*(1 - i*w^(L/2-k)), 0 < k < W/2.
      const float* yrow = ybuf;   // the row the resample reads
      float2* fin = bufA;         // where the resample writes (the other buffer when yrow lives in one of them)
      float2* fout = bufB;
      if (M > 1 && a.zp_full) {
        // FULL-length form (round 6; what fdoct_big.hip does with the rows in HBM, here in the two LDS buffers): spec = the
        // W-point +i transform of the row (so F = conj(spec) / W: DFT_SCALE), the padded spectrum by pad_source's rule --
        // an odd width's stray column, the dropped Nyquist bin, Im F[0] ignored, BscanDark's band-pass --, the zn-point +i
        // transform of it (main:241), and its real parts as the upsampled row; column M W - 1, which an odd width under an even
        // multiplier lacks, reads as 0 like data_ylin[0].
        if constexpr (!IP) {
          const int zn = a.zn;
          const float2* S = nullptr;
          if (!bp_direct) {
            for (int i = tid; i < W; i += nt) bufA[i] = make_float2(ybuf[i], 0.f);
            __syncthreads();
            S = dft_any_inverse(bufA, bufB, a.zf);
          }
          float2* Zb = (S == bufA) ? bufB : bufA;
          const float inv_w = 1.f / (float)W;
          for (int pos = tid; pos < zn; pos += nt) {
            bool mirror;
            const int ks = pad_source(W, zn, a.bandpass, pos, &mirror);
            float2 v = make_float2(0.f, 0.f);
            if (ks >= 0) {
              float fx, fy;
              if (bp_direct) {  // (the band-pass keeps 3 <= ks < W / 10 and an odd width's stray column W - 1)
                const float2 x = Xbp[ks == W - 1 ? KB : ks - 3];
                fx = x.x;
                fy = x.y;
              } else {
                const float2 sp = S[ks];
                fx = sp.x * inv_w;
                fy = (ks == 0) ? 0.f : -sp.y * inv_w;
              }
              v = mirror ? make_float2(fx, -fy) : make_float2(fx, fy);
            }
            Zb[pos] = v;
          }
          __syncthreads();
          float2* Y = dft_any_inverse(Zb, (Zb == bufA) ? bufB : bufA, a.zi);
          float2* other = (Y == bufA) ? bufB : bufA;
          float* yr = reinterpret_cast<float*>(other);
          for (int i = tid; i < MW; i += nt) yr[i] = i < zn ? Y[i].x : 0.f;
          __syncthreads();
          yrow = yr;
          fin = Y;
          fout = other;
        }
      } else if (M > 1) {
        const int Wh = W >> 1, Lh = MW >> 1;
        const float2* Zf = bufA;
        if (!bp_direct) {
          for (int i = tid; i < W; i += nt) reinterpret_cast<float*>(bufA)[i] = ybuf[i];
          __syncthreads();
          Zf = fft_lds<false, IP, R16>(bufA, bufB, Wh, a.rad_wh, a.mag_wh, a.npass_wh, a.tw_wh, tid);  // forward, half length
        }
        float2* Zb = IP ? bufA : ((Zf == bufA) ? bufB : bufA);
        const float inv_w = 1.f / (float)W;  // DFT_SCALE
        // BscanDark.cpp's band-pass (dark:218-236) blanks the shifted spectrum's outer 40 % on both sides and 3 bins either
        // side of DC: of the bins that survive the Hermitian read, 3 <= k < floor(W/10) remain
        const int bp_lo = a.bandpass ? 3 : 0, bp_hi = a.bandpass ? W / 10 : Wh;
        auto spectrum = [&](int k) -> float2 {  // X[k] = F[k]/W for 0 <= k < W/2
          if (k < bp_lo || k >= bp_hi) return make_float2(0.f, 0.f);
          if (bp_direct) return Xbp[k - 3];
          const float2 zk = Zf[k], zp = Zf[k == 0 ? 0 : Wh - k];
          const float ax = zk.x + zp.x, ay = zk.y - zp.y, bx = zk.x - zp.x, by = zk.y + zp.y;  // A = zk + conj zp, B = zk - conj zp
          const float2 t = a.tw_w[k];                                                        // e^(+2*pi*i*k/W); we need its conjugate
          const float qx = fmaf(t.y, by, t.x * bx), qy = fmaf(-t.y, bx, t.x * by);           // q = conj(t) * B
          // F = (A - i*q)/2 = (ax + qy, ay - qx)/2
          return make_float2(0.5f * (ax + qy) * inv_w, k == 0 ? 0.f : 0.5f * (ay - qx) * inv_w);
        };
        // both bands come from the same X[k] and w^k (w^(L/2-k) = -conj(w^k)): one sweep over k < W/2, zeros in between
        auto bands = [&](int k, float2& lo, float2& hi) {
          const float2 x = spectrum(k), w = a.tw_mw[k];
          const float px = fmaf(-x.y, w.y, x.x * w.x), py = fmaf(x.y, w.x, x.x * w.y);  // x * w
          lo = make_float2(x.x - py, x.y + px);                                           // x * (1 + i*w)
          const float cx = x.x, cy = -x.y;                                                // c = conj X[k], w' = (-w.x, w.y)
          const float qx = fmaf(-cy, w.y, cx * -w.x), qy = fmaf(cy, -w.x, cx * w.y);      // c * w'
          hi = make_float2(cx + qy, cy - qx);                                             // c * (1 - i*w')
        };
        if constexpr (IP) {  // the spectrum is re-packed where it lies: all of it is read (registers) before any of it is written
          constexpr int KMAX = 8;  // W/2 <= KMAX * blockDim.x (host)
          float2 lo[KMAX], hi[KMAX];
#pragma unroll
          for (int i = 0; i < KMAX; i++) {
            const int k = tid + i * nt;
            if (k < Wh) bands(k, lo[i], hi[i]);
          }
          __syncthreads();
#pragma unroll
          for (int i = 0; i < KMAX; i++) {
            const int k = tid + i * nt;
            if (k < Wh) {
              Zb[k] = lo[i];
              if (k > 0) Zb[Lh - k] = hi[i];
            }
          }
        } else {
          for (int k = tid; k < Wh; k += nt) {
            float2 lo, hi;
            bands(k, lo, hi);
            Zb[k] = lo;
            if (k > 0) Zb[Lh - k] = hi;
          }
        }
        for (int k = Wh + tid; k <= Lh - Wh; k += nt) Zb[k] = make_float2(0.f, 0.f);
        __syncthreads();
        float2* other = (Zb == bufA) ? bufB : bufA;
        float2* Y = fft_lds<true, IP, R16>(Zb, other, Lh, a.rad_mwh, a.mag_mwh, a.npass_mwh, a.tw_mwh, tid);
        yrow = reinterpret_cast<const float*>(Y);
        fin = IP ? bufA : ((Y == bufA) ? bufB : bufA);
        fout = Y;
      }

      // ---- A5: lambda -> k resample with the reference's indexing (main:1151-1177), A6/A6'
      auto resampled = [&](int q) -> float {
        float yl = 0.f;
        if (q >= 1 && q <= N - 2) {
          const int i = a.idx[q];
          const float yi = yrow[i];
          const float slope = (i == 0) ? (yrow[1] - yrow[0]) : (yi - yrow[i - 1]);
          yl = fmaf(a.g[i], slope, yi);
        }
        return yl;
      };
      auto put = [&](int q, float yl) {
        if (a.real_half)
          reinterpret_cast<float*>(fin)[q] = yl;  // z[n] = ylin[2n] + i*ylin[2n+1]
        else
          fin[q] = a.phase ? make_float2(yl * a.phase[q].x, yl * a.phase[q].y) : make_float2(yl, 0.f);
      };
      if constexpr (IP) {  // the upsampled row and the transform's input share the buffer: gather into registers first
        constexpr int QMAX = 32;  // numfftpoints <= QMAX * blockDim.x (host)
        float yl[QMAX];
#pragma unroll
        for (int i = 0; i < QMAX; i++) {
          const int q = tid + i * nt;
          yl[i] = q < N ? resampled(q) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < QMAX; i++) {
          const int q = tid + i * nt;
          if (q < N) put(q, yl[i]);
        }
      } else {
        for (int q = tid; q < N; q += nt) put(q, resampled(q));
      }
      __syncthreads();
      // ---- A7: N-point inverse DFT (unscaled), A8: magnitude of the first D bins
      if (a.real_half) {
        // real row: Z = IDFT_{N/2}(z), then X[k] = (A - i*w^k*B)/2 with A = Z[k] + conj Z[N/2-k], B = Z[k] - conj Z[N/2-k],
        // w = exp(+2*pi*i/N) (indices mod N/2); bins above N/2 mirror: |X[b]| = |X[N-b]|
        const int NC = N >> 1;
        const float2* Z = (a.blu_m && !IP) ? bluestein_inverse(fin, fout, NC, a) : fft_lds<true, IP, R16>(fin, fout, NC, a.rad_nh, a.mag_nh, a.npass_nh, a.tw_nh, tid);
        for (int b = tid; b < D; b += nt) {
          const int k = (b <= NC) ? b : N - b;
          const float2 zk = Z[k == NC ? 0 : k];
          const float2 zp = Z[(k == 0 || k == NC) ? 0 : NC - k];
          const float2 w = a.tw_n[k];
          const float ax = zk.x + zp.x, ay = zk.y - zp.y, bx = zk.x - zp.x, by = zk.y + zp.y;
          const float qx = fmaf(-w.y, by, w.x * bx), qy = fmaf(w.y, bx, w.x * by);
          const float xr = ax + qy, xi = ay - qx;
          const float m = 0.5f * sqrtf(fmaf(xr, xr, xi * xi));
          accbuf[b] = (ai == 0) ? m : accbuf[b] + m;  // each bin belongs to one thread: no race
        }
      } else {
        const float2* X = (a.blu_m && !IP) ? bluestein_inverse(fin, fout, N, a) : fft_lds<true, IP, R16>(fin, fout, N, a.rad_n, a.mag_n, a.npass_n, a.tw_n, tid);
        for (int b = tid; b < D; b += nt) {
          const float2 x = X[b];
          const float m = sqrtf(fmaf(x.x, x.x, x.y * x.y));
          accbuf[b] = (ai == 0) ? m : accbuf[b] + m;
        }
      }
      __syncthreads();
    }

    // ---- A9/A10
    float* om = a.out_mag ? a.out_mag + (size_t)o * D : nullptr;
    float* od = a.out_db ? a.out_db + (size_t)o * D : nullptr;
    float db4 = 0.f;
    if (od && a.dcmask && D > 4) {
      if (tid0 == 4) bcast[0] = a.db_scale * log2f(fmaf(accbuf[4], a.inv_A, a.eps));
      __syncthreads();
      db4 = bcast[0];
    }
    for (int b = tid0; b < D; b += nt) {
      const float v = fmaf(accbuf[b], a.inv_A, a.eps);
      if (om) om[b] = v;
      if (od) od[b] = (a.dcmask && D > 4 && b < 2) ? db4 : a.db_scale * log2f(v);
    }
    __syncthreads();
    o = a.row_ticket ? next_row[parity] : o + gridDim.x;
    parity ^= 1;
  }
}

// smoothmovavg (main:247-304): (2n+1) taps, taps outside the row replaced by the centre sample, centre
// counted twice.  Any input type -> packed f32 frames of the tap SUMS (exact for the camera's integer samples: an f32 quotient
// by 2(n+1) would round at the size of the DC level); the divisor is folded into the planes the chain divides and subtracts by
// (fdoct_capi.cpp::plane_scales).
__global__ void movavg_kernel(const void* frames, int dtype, long long pitch_bytes, int W, long long rows, int n,
                              float* out) {
  const long long total = rows * W;
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
    out[e] = s;
  }
}

// One LdsGrant per KERNEL: the kernel is a non-type template argument, so every instantiation has its own static (the four
// kernels share one function-pointer TYPE; a generic lambda over `auto k` would be instantiated once and share one table --
// ADVICE r4).
template <auto K>
static hipError_t launch_generic_one(const GenericArgs& a, int grid, int threads, size_t lds, hipStream_t st) {
  static LdsGrant grant;
  if (hipError_t e = grant.ensure(K, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(K, dim3(grid), dim3(threads), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_generic(const GenericArgs& a, int grid, size_t lds, hipStream_t st) {
  static const int force_nt = [] { const char* e = std::getenv("FDOCT_GENERIC_THREADS"); return e ? std::atoi(e) : 0; }();  // measurement: 256 / 512 / 1024
  const int per_cu = (int)((160 * 1024 - 1024) / lds);  // rows (workgroups) the LDS holds per CU
  int nt = per_cu >= 3 ? 256 : (per_cu == 2 ? 512 : 1024);
  if (force_nt == 256 || force_nt == 512 || force_nt == 1024) nt = force_nt;
  if (a.radix16) nt = 1024;  // (the plan holds radix-16 passes: only the 1024-thread kernels have them)
  if (a.inplace) return launch_generic_one<generic_kernel<1024, 1, true>>(a, grid, 1024, lds, st);  // (one DFT buffer: the host has checked its limits)
  if (nt == 1024) return launch_generic_one<generic_kernel<1024, 1>>(a, grid, 1024, lds, st);
  if (nt == 512) return launch_generic_one<generic_kernel<512, 2>>(a, grid, 512, lds, st);
  return launch_generic_one<generic_kernel<256, 6>>(a, grid, 256, lds, st);
}

// The same on frames handed over as doubles (main:987): the tap sums in double -- f32 sums of non-integer samples would round
// at the size of the DC level -- split into two f32 planes, sum = hi + lo, which the chain carries into the division.
template <typename IN>
__global__ void movavg_f64_kernel(const IN* in, long long pitch_elems, int W, long long rows, int n, float* hi, float* lo) {
  const long long total = rows * W;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / W;
    const int j = (int)(e - r * W);
    const IN* row = in + r * pitch_elems;
    const double c = (double)row[j];
    double s = c;  // the extra centre weight
    for (int k = -n; k <= n; k++) {
      const int jj = j + k;
      s += (jj >= 0 && jj < W) ? (double)row[jj] : c;
    }
    const float h = (float)s;
    hi[e] = h;
    const double l = s - (double)h;
    lo[e] = (l == l && h - h == 0.f) ? (float)l : 0.f;
  }
}

hipError_t launch_movavg_f64(const double* in, long long pitch_elems, int W, long long rows, int n, float* hi, float* lo, hipStream_t st) {
  hipLaunchKernelGGL(movavg_f64_kernel<double>, dim3(2048), dim3(256), 0, st, in, pitch_elems, W, rows, n, hi, lo);
  return hipGetLastError();
}
hipError_t launch_movavg_f32_wide(const float* in, long long pitch_elems, int W, long long rows, int n, float* hi, float* lo, hipStream_t st) {
  hipLaunchKernelGGL(movavg_f64_kernel<float>, dim3(2048), dim3(256), 0, st, in, pitch_elems, W, rows, n, hi, lo);
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

// n x n median (n = 3, 5, 7), BORDER_REPLICATE as cv::medianBlur.  One pixel per thread, the n*n neighbours in
// registers; the median by "forgetful selection": of any K/2 + 2 values neither the smallest nor the largest can be
// the median of all K, so keep a window of that size, drop its min and max, take in the next value, repeat.  All
// loops unroll at compile time (min/max pairs only, no data-dependent branches, no scratch).
__device__ __forceinline__ void sort2(unsigned& a, unsigned& b) {
  const unsigned lo = min(a, b), hi = max(a, b);
  a = lo;
  b = hi;
}

template <int S>
__device__ __forceinline__ void drop_min_max(unsigned* w) {  // afterwards w[0] = min, w[S-1] = max of w[0..S-1]
  static_for<0, S - 1>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    sort2(w[i], w[i + 1]);
  });
  static_for<0, S - 2>([&](auto ic) {
    constexpr int i = S - 2 - decltype(ic)::value;  // S-2 .. 1
    sort2(w[i - 1], w[i]);
  });
}

template <int K, int S, int NEXT>
__device__ __forceinline__ unsigned forgetful_median(unsigned* w, const unsigned* v) {
  // w[0..S-1] is the live window, v[NEXT..K-1] still to come
  if constexpr (S == 1) {
    return w[0];
  } else {
    drop_min_max<S>(w);
    // survivors are w[1..S-2]: shift down, append the next input if there is one
    static_for<0, S - 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      w[i] = w[i + 1];
    });
    if constexpr (NEXT < K) {
      w[S - 2] = v[NEXT];
      return forgetful_median<K, S - 1, NEXT + 1>(w, v);
    } else {
      return forgetful_median<K, S - 2, NEXT>(w, v);
    }
  }
}

template <typename T, int N>
__global__ __launch_bounds__(256) void median_kernel(const T* in, long long in_pitch, T* out, long long out_pitch, int w, int h,
                                                     int nframes) {
  constexpr int K = N * N, R = N / 2, M = K / 2 + 2;
  const long long total = (long long)nframes * h * w;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(e % w);
    const long long fy = e / w;  // frame*h + y
    const int y = (int)(fy % h);
    const long long f = fy / h;
    unsigned v[K];
    static_for<0, N>([&](auto dyc) {
      constexpr int dy = decltype(dyc)::value - R;
      const int yy = min(max(y + dy, 0), h - 1);
      const T* row = reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(in) + (f * h + yy) * in_pitch);
      static_for<0, N>([&](auto dxc) {
        constexpr int dx = decltype(dxc)::value - R;
        v[(dy + R) * N + dx + R] = row[min(max(x + dx, 0), w - 1)];
      });
    });
    unsigned win[M];
    static_for<0, M>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      win[i] = v[i];
    });
    const unsigned med = forgetful_median<K, M, M>(win, v);
    reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(out) + fy * out_pitch)[x] = (T)med;
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

// 3 x 3 median, eight pixels of a row per thread (cv::medianBlur(ksize 3), replicated border; main:987).  The three rows are
// held as pairs of 16-bit pixels in two packings -- M_j = (p[2j], p[2j+1]) and P_j = (p[2j-1], p[2j]) -- so that an output
// pair's left neighbours are P_j, itself M_j, its right neighbours P_(j+1), and everything is v_pk_min/max_u16 on whole
// registers: the columns are sorted once (low / middle / high of each column, shared by the three windows that contain
// it), then  median = med3(max of the lows, med3 of the middles, min of the highs).  ~16 packed operations and 9/8 loads
// per pixel where the generic kernel spends ~60 scalar ones and 9 loads; bit-exact (the median is a value of the input).
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 pmin(us2 a, us2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ us2 pmax(us2 a, us2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ us2 pmed3(us2 a, us2 b, us2 c) { return pmax(pmin(a, b), pmin(pmax(a, b), c)); }
__device__ __forceinline__ us2 us2_from(unsigned v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ unsigned us2_bits(us2 v) { return __builtin_bit_cast(unsigned, v); }

template <typename T>
struct Row8;  // eight pixels starting at x0 (a multiple of 8) as M_0 .. M_3
template <>
struct Row8<uint16_t> {
  static __device__ __forceinline__ void load(const uint16_t* row, int x0, unsigned* m) {
    const uint4 v = *reinterpret_cast<const uint4*>(row + x0);
    m[0] = v.x; m[1] = v.y; m[2] = v.z; m[3] = v.w;
  }
  static __device__ __forceinline__ void store(uint16_t* row, int x0, const unsigned* m) {
    *reinterpret_cast<uint4*>(row + x0) = make_uint4(m[0], m[1], m[2], m[3]);
  }
};
template <>
struct Row8<uint8_t> {
  static __device__ __forceinline__ void load(const uint8_t* row, int x0, unsigned* m) {
    const uint2 v = *reinterpret_cast<const uint2*>(row + x0);
    m[0] = (v.x & 0xffu) | ((v.x & 0xff00u) << 8);
    m[1] = ((v.x >> 16) & 0xffu) | ((v.x >> 8) & 0xff0000u);
    m[2] = (v.y & 0xffu) | ((v.y & 0xff00u) << 8);
    m[3] = ((v.y >> 16) & 0xffu) | ((v.y >> 8) & 0xff0000u);
  }
  static __device__ __forceinline__ void store(uint8_t* row, int x0, const unsigned* m) {
    uint2 v;
    v.x = (m[0] & 0xffu) | ((m[0] >> 8) & 0xff00u) | ((m[1] & 0xffu) << 16) | ((m[1] << 8) & 0xff000000u);
    v.y = (m[2] & 0xffu) | ((m[2] >> 8) & 0xff00u) | ((m[3] & 0xffu) << 16) | ((m[3] << 8) & 0xff000000u);
    *reinterpret_cast<uint2*>(row + x0) = v;
  }
};

template <typename T>
__global__ __launch_bounds__(256) void median3_fast_kernel(const T* in, long long in_pitch, T* out, long long out_pitch, int w, int h,
                                                           long long nrows) {
  const int xq = blockIdx.x * blockDim.x + threadIdx.x;
  if (xq * 8 >= w) return;
  const int x0 = xq * 8;
  const int xl = x0 > 0 ? x0 - 1 : 0, xr = x0 + 8 < w ? x0 + 8 : w - 1;
  for (long long fy = blockIdx.y; fy < nrows; fy += gridDim.y) {
    const int y = (int)(fy % h);
    const unsigned char* base = reinterpret_cast<const unsigned char*>(in);
    const T* rows[3] = {reinterpret_cast<const T*>(base + (fy - (y > 0 ? 1 : 0)) * in_pitch), reinterpret_cast<const T*>(base + fy * in_pitch),
                        reinterpret_cast<const T*>(base + (fy + (y < h - 1 ? 1 : 0)) * in_pitch)};
    us2 M[3][4], P[3][5];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      unsigned m[4];
      Row8<T>::load(rows[r], x0, m);
      const unsigned left = rows[r][xl], right = rows[r][xr];
#pragma unroll
      for (int j = 0; j < 4; j++) M[r][j] = us2_from(m[j]);
      P[r][0] = us2_from((m[0] << 16) | left);
      P[r][1] = us2_from(__builtin_amdgcn_alignbit(m[1], m[0], 16));
      P[r][2] = us2_from(__builtin_amdgcn_alignbit(m[2], m[1], 16));
      P[r][3] = us2_from(__builtin_amdgcn_alignbit(m[3], m[2], 16));
      P[r][4] = us2_from((m[3] >> 16) | (right << 16));
    }
    // column sorts: afterwards row 0 <= row 1 <= row 2 in every column
    auto sort_cols = [](us2& a, us2& b, us2& c) {
      us2 t = pmin(a, b);
      b = pmax(a, b);
      a = t;
      t = pmin(b, c);
      c = pmax(b, c);
      b = t;
      t = pmin(a, b);
      b = pmax(a, b);
      a = t;
    };
#pragma unroll
    for (int j = 0; j < 4; j++) sort_cols(M[0][j], M[1][j], M[2][j]);
#pragma unroll
    for (int j = 0; j < 5; j++) sort_cols(P[0][j], P[1][j], P[2][j]);
    unsigned res[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const us2 lo = pmax(pmax(P[0][j], M[0][j]), P[0][j + 1]);
      const us2 mid = pmed3(P[1][j], M[1][j], P[1][j + 1]);
      const us2 hi = pmin(pmin(P[2][j], M[2][j]), P[2][j + 1]);
      res[j] = us2_bits(pmed3(lo, mid, hi));
    }
    Row8<T>::store(reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(out) + fy * out_pitch), x0, res);
  }
}

template <typename T>
static hipError_t launch_median_t(const void* in, long long in_pitch, void* out, long long out_pitch, int w, int h, int n, int nframes,
                                  hipStream_t st) {
  if (n == 3 && w % 8 == 0 && in_pitch % 16 == 0 && out_pitch % 16 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0) {
    const long long nrows = (long long)nframes * h;
    const int chunks = w / 8, bx = chunks < 256 ? ((chunks + 63) / 64) * 64 : 256;
    const dim3 g((chunks + bx - 1) / bx, (unsigned)(nrows < 32768 ? nrows : 32768));
    hipLaunchKernelGGL((median3_fast_kernel<T>), g, dim3(bx), 0, st, static_cast<const T*>(in), in_pitch, static_cast<T*>(out), out_pitch, w, h, nrows);
    return hipGetLastError();
  }
  const dim3 g(8192), b(256);
  const T* i = static_cast<const T*>(in);
  T* o = static_cast<T*>(out);
  switch (n) {
    case 3: hipLaunchKernelGGL((median_kernel<T, 3>), g, b, 0, st, i, in_pitch, o, out_pitch, w, h, nframes); break;
    case 5: hipLaunchKernelGGL((median_kernel<T, 5>), g, b, 0, st, i, in_pitch, o, out_pitch, w, h, nframes); break;
    case 7: hipLaunchKernelGGL((median_kernel<T, 7>), g, b, 0, st, i, in_pitch, o, out_pitch, w, h, nframes); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_median(const void* in, long long in_pitch, void* out, long long out_pitch, int dtype, int w, int h, int n,
                         int nframes, hipStream_t st) {
  if (dtype == FDOCT_K_U8) return launch_median_t<uint8_t>(in, in_pitch, out, out_pitch, w, h, n, nframes, st);
  if (dtype == FDOCT_K_U16) return launch_median_t<uint16_t>(in, in_pitch, out, out_pitch, w, h, n, nframes, st);
  return hipErrorInvalidValue;
}

// 2x2 binning of 16-byte-aligned rows, the common case (binvalue = 2): 16 bytes from each of the two raw rows in,
// 8 bytes out per thread.  u16: (a + b + c + d + 2) >> 2 per output sample, horizontal pairs are the halves of a dword.
template <typename T>
__global__ __launch_bounds__(256) void bin2x2_kernel(const unsigned char* in, long long in_pitch, unsigned char* out,
                                                     long long out_pitch, int vecs_per_row, long long out_rows) {
  const long long total = out_rows * vecs_per_row;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long oy = e / vecs_per_row;  // frame*oh + y (raw frames hold 2*oh rows, so raw row = 2*oy)
    const int v = (int)(e - oy * vecs_per_row);
    const uint4 r0 = reinterpret_cast<const uint4*>(in + (2 * oy) * in_pitch)[v];
    const uint4 r1 = reinterpret_cast<const uint4*>(in + (2 * oy + 1) * in_pitch)[v];
    const unsigned a[4] = {r0.x, r0.y, r0.z, r0.w}, b[4] = {r1.x, r1.y, r1.z, r1.w};
    if constexpr (sizeof(T) == 2) {
      unsigned o[4];
#pragma unroll
      for (int i = 0; i < 4; i++) o[i] = ((a[i] & 0xffffu) + (a[i] >> 16) + (b[i] & 0xffffu) + (b[i] >> 16) + 2u) >> 2;
      reinterpret_cast<uint2*>(out + oy * out_pitch)[v] = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
    } else {
      unsigned o[8];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        o[2 * i] = ((a[i] & 0xffu) + ((a[i] >> 8) & 0xffu) + (b[i] & 0xffu) + ((b[i] >> 8) & 0xffu) + 2u) >> 2;
        o[2 * i + 1] = (((a[i] >> 16) & 0xffu) + (a[i] >> 24) + ((b[i] >> 16) & 0xffu) + (b[i] >> 24) + 2u) >> 2;
      }
      reinterpret_cast<uint2*>(out + oy * out_pitch)[v] =
          make_uint2(o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24), o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24));
    }
  }
}

hipError_t launch_bin(const void* in, long long in_pitch, void* out, long long out_pitch, int dtype, int ow, int oh, int binx,
                      int biny, int nframes, hipStream_t st) {
  const int es = dtype == FDOCT_K_U16 ? 2 : 1;
  // fast path: 2x2, rows of whole 16-byte input vectors (= 8-byte output vectors), aligned pointers and pitches
  if (binx == 2 && biny == 2 && (dtype == FDOCT_K_U8 || dtype == FDOCT_K_U16) && (2 * ow * es) % 16 == 0 && in_pitch % 16 == 0 &&
      out_pitch % 8 == 0 && reinterpret_cast<uintptr_t>(in) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0) {
    const int vecs = 2 * ow * es / 16;
    const long long out_rows = (long long)nframes * oh;
    const auto* i8 = static_cast<const unsigned char*>(in);
    auto* o8 = static_cast<unsigned char*>(out);
    if (dtype == FDOCT_K_U16)
      hipLaunchKernelGGL(bin2x2_kernel<uint16_t>, dim3(8192), dim3(256), 0, st, i8, in_pitch, o8, out_pitch, vecs, out_rows);
    else
      hipLaunchKernelGGL(bin2x2_kernel<uint8_t>, dim3(8192), dim3(256), 0, st, i8, in_pitch, o8, out_pitch, vecs, out_rows);
    return hipGetLastError();
  }
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
