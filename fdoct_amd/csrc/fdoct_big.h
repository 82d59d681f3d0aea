// fdoct_big.h -- interface of the long-row path (fdoct_big.hip): rows in HBM between the steps of the chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fdoct_kernels.h"

namespace fdoct {

// Arguments of the elementwise steps for one chunk of whole averaging groups.  All pointers are device pointers.
struct BigArgs {
  const void* frames;        // first input row of the chunk
  long long pitch_bytes;
  long long in_rows;         // input A-scans of the chunk (groups * A * H)
  long long out_rows;        // output A-scans of the chunk (groups * H)
  int dtype;                 // FDOCT_K_*
  int W, H, N, D, M, A;
  const float* ib; int ib_2d;
  const float* il;           // low word of 1/background, indexed like ib (fdoct_capi.cpp::reciprocal_words)
  const float* yp; int yp_2d;
  const float* yd; int yd_2d;
  const float* win;          // [W]
  const float2* minmax;      // per input frame of the chunk (min, max) or null
  int rowwisenormalize, dcmask;
  float inv_A, eps, db_scale;
};

hipError_t big_launch_pre(const BigArgs& a, float* y, hipStream_t st);
hipError_t big_launch_real_to_complex(const float* y, long long total, float2* z, hipStream_t st);
// one +i Stockham pass of radix 8/4/2/5/3 over `rows` rows of length n; tw[m] = exp(+2 pi i m / n)
hipError_t big_launch_fft_pass(const float2* src, float2* dst, long long rows, int n, int radix, int Ns, const float2* tw, hipStream_t st);
hipError_t big_launch_chirp_in(const float2* x, long long rows, int n, int mb, const float2* chirp, float2* out, hipStream_t st);
hipError_t big_launch_conj_mul(float2* z, long long rows, int mb, const float2* bhat, hipStream_t st);
hipError_t big_launch_chirp_out(const float2* cbuf, long long rows, int n, int mb, const float2* chirp, float2* out, hipStream_t st);
hipError_t big_launch_pad(const float2* spec, long long rows, int W, int MW, int bandpass, float2* z, hipStream_t st);
hipError_t big_launch_resample(const float* yr, const float2* yc, long long rows, int ylen, int N, const int32_t* idx, const float* g,
                               const float2* phase, float2* z, hipStream_t st);
hipError_t big_launch_post(const float2* X, const BigArgs& a, float* out_mag, float* out_db, hipStream_t st);

}  // namespace fdoct
