// fdoct_big.h -- interface of the long-row path (fdoct_big.hip): rows in HBM between the steps of the chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fdoct_kernels.h"

namespace fdoct {

// Arguments of the elementwise steps for one chunk of whole averaging groups.  All pointers are device pointers.
struct BigArgs {
  const void* frames;        // first input row of the chunk
  const float* frames_lo;    // low words of f64 frames (same pitch) or null
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
// Several Stockham passes in ONE launch, the data in LDS between them (round 4: a 16384-point transform is 2 launches of
// 16 bytes per point each way instead of 5).  The passes of a transform are cut into groups; group g has the local length Q
// (the product of its radices), P = the product of the earlier groups' lengths, F = n / (P Q).  Its butterflies only ever
// combine the Q elements  s + S t  (t < Q, S = P F)  of sub-problem s = k1 + P a (k1 < P, a < F), as a plain Q-point Stockham
// transform whose twiddles use the global index  k = k1 + P k_local  over  Ns = P Ns_local,  and whose result e goes to
// k1 + P (a Q + e).  A workgroup takes `TS` consecutive sub-problems (adjacent in memory both ways: 8 TS contiguous bytes).
// The first group may read its input through a loader (the elementwise step in front of the transform fused into it), the
// last one may crop its output.
enum { BIG_LOAD_CPLX = 0, BIG_LOAD_REAL = 1, BIG_LOAD_PAD = 2, BIG_LOAD_RESAMPLE = 3 };
constexpr int BIG_GROUP_MAX_PASSES = 6;
constexpr int BIG_GROUP_TILE_VALUES = 2048;  // values of one workgroup's tile (sub-problems x local length): 8 per thread in registers during a pass
struct BigGroup {
  const float2* src;         // BIG_LOAD_CPLX: rows of n complex values; BIG_LOAD_PAD: the W-point spectra (rows of W)
  float2* dst;               // rows of n
  long long rows;
  int n, P, Q, F, log2ts;
  int npass, rad[BIG_GROUP_MAX_PASSES];
  int out_limit;             // elements at positions >= out_limit of a row are not stored (n: all of them)
  const float2* tw;          // exp(+2 pi i m / n)
  int load;                  // BIG_LOAD_*
  const float* yr;           // BIG_LOAD_REAL: rows of n floats; BIG_LOAD_RESAMPLE: the plain rows (or null)
  const float2* yc;          // BIG_LOAD_RESAMPLE: the upsampled rows (real part used) or null
  int ylen;                  //   their length
  const int32_t* idx;        //   nearestkindex
  const float* g;            //   fractionalk by sample
  const float2* phase;       //   dispersion phasors or null
  int W, bandpass;           // BIG_LOAD_PAD: the row width (n = M W) and BscanDark's band-pass
};
size_t big_group_lds_bytes(const BigGroup& g);
hipError_t big_launch_fft_group(const BigGroup& g, hipStream_t st);
hipError_t big_launch_chirp_in(const float2* x, long long rows, int n, int mb, const float2* chirp, float2* out, hipStream_t st);
hipError_t big_launch_conj_mul(float2* z, long long rows, int mb, const float2* bhat, hipStream_t st);
hipError_t big_launch_chirp_out(const float2* cbuf, long long rows, int n, int mb, const float2* chirp, float2* out, hipStream_t st);
hipError_t big_launch_pad(const float2* spec, long long rows, int W, int MW, int bandpass, float2* z, hipStream_t st);
hipError_t big_launch_resample(const float* yr, const float2* yc, long long rows, int ylen, int N, const int32_t* idx, const float* g,
                               const float2* phase, float2* z, hipStream_t st);
hipError_t big_launch_post(const float2* X, const BigArgs& a, float* out_mag, float* out_db, hipStream_t st);

}  // namespace fdoct
