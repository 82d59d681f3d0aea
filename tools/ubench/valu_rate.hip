// Micro-benchmark: VALU issue rate on gfx950 for scalar v_fma_f32 vs packed v_pk_fma_f32 at 1..4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 2048, UNR = 16;

__global__ void k_scalar(float* out, float a, float b) {
  float x[UNR];
  for (int i = 0; i < UNR; i++) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < UNR; i++) x[i] = __builtin_fmaf(x[i], a, b);
  }
  float s = 0;
  for (int i = 0; i < UNR; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_packed(float* out, float a, float b) {
  v2f x[UNR];
  for (int i = 0; i < UNR; i++) x[i] = (v2f){threadIdx.x * 0.001f + i, 1.0f * i};
  const v2f va = {a, a}, vb = {b, b};
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < UNR; i++) x[i] = __builtin_elementwise_fma(x[i], va, vb);
  }
  float s = 0;
  for (int i = 0; i < UNR; i++) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mix(float* out, float a, float b) {  // dependent chain of 1 (latency-bound per wave)
  float x = threadIdx.x * 0.001f;
  for (int it = 0; it < ITER * UNR; it++) x = __builtin_fmaf(x, a, b);
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
  float* d; hipMalloc(&d, 1024 * 256 * 4 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("clock %d kHz, CUs %d\n", p.clockRate, p.multiProcessorCount);
  for (int kind = 0; kind < 3; kind++)
    for (int waves = 1; waves <= 4; waves++) {
      dim3 g(256), b(256 * waves);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k_scalar, g, b, 0, 0, d, 1.0001f, 0.5f);
        if (kind == 1) hipLaunchKernelGGL(k_packed, g, b, 0, 0, d, 1.0001f, 0.5f);
        if (kind == 2) hipLaunchKernelGGL(k_mix, g, b, 0, 0, d, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double instr_per_simd = (double)ITER * UNR * waves;  // wave-instructions issued per SIMD
      printf("%s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cycles at 2.4 GHz)\n",
             kind == 0 ? "v_fma_f32   " : kind == 1 ? "v_pk_fma_f32" : "dependent   ", waves, ms,
             ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
  return 0;
}
