// Micro-benchmark: do LDS instructions of one wave and VALU instructions of another wave on the same
// SIMD overlap?  Three kernels at 4 waves/SIMD (1024 threads, 1 block/CU): all waves VALU, all waves LDS,
// and half/half (waves 0-7 VALU, waves 8-15 LDS: two of each per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 4096;
__global__ void k(float* out, int mode, float a, float b) {
  __shared__ float2 buf[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = make_float2(i, 1.f);
  __syncthreads();
  const bool do_valu = mode == 0 || (mode == 2 && wave < (int)(blockDim.x >> 7));
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = lane * 0.01f + i;
  float2 acc = make_float2(0.f, 0.f);
  if (do_valu) {
    for (int it = 0; it < ITER; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = __builtin_fmaf(x[i], a, b);
    }
  } else {
    const float2* p = buf + lane;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        float2 v = p[(i * 64 + it * 8) & 4032];
        x[i] += v.x;   // 8 independent chains: the loads of one iteration are all in flight together
      }
    }
  }
  float s = acc.x;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 1024 * sizeof(float));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[3] = {"all waves VALU (8 fma/iter)", "all waves LDS (8 ds_read_b64 + 8 add/iter)", "half VALU / half LDS"};
  for (int mode = 0; mode < 3; mode++) {
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, mode, 1.0001f, 0.5f);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-45s %.3f ms\n", names[mode], ms);
  }
  return 0;
}
