// Micro-benchmark: package power and clock while ALL CUs run a loop of one instruction kind (2 waves per SIMD, like the
// fused fast path), to price the instruction kinds of the fused kernel in ENERGY -- the chip is at its power cap under
// that kernel (profiles/r02_power_clock.txt), so what an instruction costs is joules, not issue slots.
// usage (GPU box): power_probe <kind> <seconds>; prints instructions/s; tools/power_probe.sh samples rocm-smi meanwhile.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define BODY_LOOP(NAME, BODY) BODY_LOOP_X(NAME, BODY, "s_nop 0\n")
#define BODY_LOOP_X(NAME, BODY, PRE)                                                                                   \
  __global__ void __launch_bounds__(512) NAME(float* sink, int iters) {                                        \
    __shared__ float lds[8192];                                                                                 \
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 0.25f;                                    \
    __syncthreads();                                                                                            \
    const unsigned laddr = (unsigned)(uintptr_t)lds + (threadIdx.x & 63) * 8u + (threadIdx.x >> 6) * 2048u;    \
    asm volatile("v_mov_b32 v48, %0\n v_mov_b32 v10, 1.5\n v_mov_b32 v11, 2.5\n v_mov_b32 v12, 0.75\n v_mov_b32 v13, 0.25\n"  \
                 "v_mov_b32 v14, 1.25\n v_mov_b32 v15, 2.0\n v_mov_b32 v16, 0.5\n v_mov_b32 v17, 0.125\n v_mov_b32 v18, 1.0\n"  \
                 "v_mov_b32 v19, 3.0\n v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.25\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 2.0\n"   \
                 "v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.25\n v_mov_b32 v44, 0.999\n v_mov_b32 v45, 1.001\n v_mov_b32 v26, 0.001\n v_mov_b32 v27, 0.002\n" \
                 ::"v"(laddr) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22",     \
                   "v23", "v24", "v25", "v26", "v27", "v44", "v45", "v48");                                      \
    asm volatile(PRE);  /* e.g. a narrower exec mask: the loop control is scalar and does not see it */           \
    for (int it = 0; it < iters; it++) {                                                                        \
      asm volatile(".rept 16\n" BODY ".endr\n" ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",  \
                   "v20", "v21", "v22", "v23", "v24", "v25", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "memory"); \
    }                                                                                                           \
    float r;                                                                                                    \
    asm volatile("s_mov_b64 exec, -1\n v_add_f32 %0, v10, v12" : "=v"(r));                                                          \
    if (r == 123.456f) sink[0] = r;                                                                             \
  }

BODY_LOOP(k_pk_add, "v_pk_add_f32 v[10:11], v[10:11], v[26:27]\n v_pk_add_f32 v[12:13], v[12:13], v[26:27]\n v_pk_add_f32 v[14:15], v[14:15], v[26:27]\n v_pk_add_f32 v[16:17], v[16:17], v[26:27]\n"
                    "v_pk_add_f32 v[18:19], v[18:19], v[26:27]\n v_pk_add_f32 v[20:21], v[20:21], v[26:27]\n v_pk_add_f32 v[22:23], v[22:23], v[26:27]\n v_pk_add_f32 v[24:25], v[24:25], v[26:27]\n")
BODY_LOOP(k_pk_mul, "v_pk_mul_f32 v[10:11], v[10:11], v[44:45]\n v_pk_mul_f32 v[12:13], v[12:13], v[44:45]\n v_pk_mul_f32 v[14:15], v[14:15], v[44:45]\n v_pk_mul_f32 v[16:17], v[16:17], v[44:45]\n"
                    "v_pk_mul_f32 v[18:19], v[18:19], v[44:45]\n v_pk_mul_f32 v[20:21], v[20:21], v[44:45]\n v_pk_mul_f32 v[22:23], v[22:23], v[44:45]\n v_pk_mul_f32 v[24:25], v[24:25], v[44:45]\n")
BODY_LOOP(k_pk_fma, "v_pk_fma_f32 v[10:11], v[10:11], v[44:45], v[26:27]\n v_pk_fma_f32 v[12:13], v[12:13], v[44:45], v[26:27]\n v_pk_fma_f32 v[14:15], v[14:15], v[44:45], v[26:27]\n v_pk_fma_f32 v[16:17], v[16:17], v[44:45], v[26:27]\n"
                    "v_pk_fma_f32 v[18:19], v[18:19], v[44:45], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[44:45], v[26:27]\n v_pk_fma_f32 v[22:23], v[22:23], v[44:45], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[44:45], v[26:27]\n")
// the same packed fma with 32, 8 and 1 of the 64 lanes enabled: does the energy of a VALU instruction follow its active lanes?
BODY_LOOP_X(k_pk_fma_32, "v_pk_fma_f32 v[10:11], v[10:11], v[44:45], v[26:27]\n v_pk_fma_f32 v[12:13], v[12:13], v[44:45], v[26:27]\n v_pk_fma_f32 v[14:15], v[14:15], v[44:45], v[26:27]\n v_pk_fma_f32 v[16:17], v[16:17], v[44:45], v[26:27]\n"
                    "v_pk_fma_f32 v[18:19], v[18:19], v[44:45], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[44:45], v[26:27]\n v_pk_fma_f32 v[22:23], v[22:23], v[44:45], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[44:45], v[26:27]\n", "s_mov_b64 exec, 0xffffffff\n")
BODY_LOOP_X(k_pk_fma_8, "v_pk_fma_f32 v[10:11], v[10:11], v[44:45], v[26:27]\n v_pk_fma_f32 v[12:13], v[12:13], v[44:45], v[26:27]\n v_pk_fma_f32 v[14:15], v[14:15], v[44:45], v[26:27]\n v_pk_fma_f32 v[16:17], v[16:17], v[44:45], v[26:27]\n"
                    "v_pk_fma_f32 v[18:19], v[18:19], v[44:45], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[44:45], v[26:27]\n v_pk_fma_f32 v[22:23], v[22:23], v[44:45], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[44:45], v[26:27]\n", "s_mov_b64 exec, 0xff\n")
BODY_LOOP_X(k_pk_fma_1, "v_pk_fma_f32 v[10:11], v[10:11], v[44:45], v[26:27]\n v_pk_fma_f32 v[12:13], v[12:13], v[44:45], v[26:27]\n v_pk_fma_f32 v[14:15], v[14:15], v[44:45], v[26:27]\n v_pk_fma_f32 v[16:17], v[16:17], v[44:45], v[26:27]\n"
                    "v_pk_fma_f32 v[18:19], v[18:19], v[44:45], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[44:45], v[26:27]\n v_pk_fma_f32 v[22:23], v[22:23], v[44:45], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[44:45], v[26:27]\n", "s_mov_b64 exec, 1\n")
BODY_LOOP(k_add, "v_add_f32 v10, v10, v26\n v_add_f32 v12, v12, v26\n v_add_f32 v14, v14, v26\n v_add_f32 v16, v16, v26\n v_add_f32 v18, v18, v26\n v_add_f32 v20, v20, v26\n v_add_f32 v22, v22, v26\n v_add_f32 v24, v24, v26\n")
BODY_LOOP(k_fma, "v_fma_f32 v10, v10, v44, v26\n v_fma_f32 v12, v12, v44, v26\n v_fma_f32 v14, v14, v44, v26\n v_fma_f32 v16, v16, v44, v26\n v_fma_f32 v18, v18, v44, v26\n v_fma_f32 v20, v20, v44, v26\n v_fma_f32 v22, v22, v44, v26\n v_fma_f32 v24, v24, v44, v26\n")
BODY_LOOP(k_sqrt, "v_sqrt_f32 v30, v10\n v_sqrt_f32 v31, v12\n v_sqrt_f32 v32, v14\n v_sqrt_f32 v33, v16\n v_sqrt_f32 v34, v18\n v_sqrt_f32 v35, v20\n v_sqrt_f32 v36, v22\n v_sqrt_f32 v37, v24\n")
BODY_LOOP(k_nop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
BODY_LOOP(k_ds_read_b64, "ds_read_b64 v[30:31], v48\n ds_read_b64 v[32:33], v48 offset:512\n ds_read_b64 v[34:35], v48 offset:1024\n ds_read_b64 v[36:37], v48 offset:1536\n"
                         "ds_read_b64 v[30:31], v48\n ds_read_b64 v[32:33], v48 offset:512\n ds_read_b64 v[34:35], v48 offset:1024\n ds_read_b64 v[36:37], v48 offset:1536\n s_waitcnt lgkmcnt(0)\n")
BODY_LOOP(k_ds_write_b64, "ds_write_b64 v48, v[10:11]\n ds_write_b64 v48, v[12:13] offset:512\n ds_write_b64 v48, v[14:15] offset:1024\n ds_write_b64 v48, v[16:17] offset:1536\n"
                          "ds_write_b64 v48, v[18:19]\n ds_write_b64 v48, v[20:21] offset:512\n ds_write_b64 v48, v[22:23] offset:1024\n ds_write_b64 v48, v[24:25] offset:1536\n s_waitcnt lgkmcnt(0)\n")

int main(int argc, char** argv) {
  const char* kind = argc > 1 ? argv[1] : "pk_fma";
  const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
  float* sink;
  hipMalloc(&sink, 64);
  typedef void (*kern_t)(float*, int);
  kern_t k = nullptr;
  if (!strcmp(kind, "pk_add")) k = k_pk_add;
  if (!strcmp(kind, "pk_mul")) k = k_pk_mul;
  if (!strcmp(kind, "pk_fma")) k = k_pk_fma;
  if (!strcmp(kind, "pk_fma_32")) k = k_pk_fma_32;
  if (!strcmp(kind, "pk_fma_8")) k = k_pk_fma_8;
  if (!strcmp(kind, "pk_fma_1")) k = k_pk_fma_1;
  if (!strcmp(kind, "add")) k = k_add;
  if (!strcmp(kind, "fma")) k = k_fma;
  if (!strcmp(kind, "sqrt")) k = k_sqrt;
  if (!strcmp(kind, "nop")) k = k_nop;
  if (!strcmp(kind, "ds_read_b64")) k = k_ds_read_b64;
  if (!strcmp(kind, "ds_write_b64")) k = k_ds_write_b64;
  if (!k) { fprintf(stderr, "unknown kind %s\n", kind); return 1; }
  const int iters = 20000;  // 20000 x 16 x 8 = 2.56 M instructions per wave per launch
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int i = 0; i < 4; i++) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, sink, iters);
    hipDeviceSynchronize();
    launches += 4;
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double wave_instr = (double)launches * 256 * 8 * iters * 16 * 8;
  printf("%-14s %.2f s: %.3e wave-instructions/s over the chip (%.2f cycles per instruction per SIMD at 2.4 GHz)\n", kind, dt,
         wave_instr / dt, 1024 * 2.4e9 / (wave_instr / dt));
  return 0;
}
