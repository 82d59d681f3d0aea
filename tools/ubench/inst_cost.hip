// Micro-benchmark: issue cost of the instruction kinds the fused kernel is made of, on gfx950, at 1 and 2 waves per
// SIMD (the fused fast path runs 2).  Each test is a block of 8 independent instructions repeated; cost = shader cycles
// (s_memtime) per wave per instruction, and per SIMD per instruction (what bounds a throughput-limited kernel).
// build: hipcc --offload-arch=gfx950 -O3 inst_cost.hip -o inst_cost ; run on the GPU box (tools/README.md).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REG_CLOBBERS                                                                                                      \
  "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26",  \
      "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",     \
      "v43", "v44", "v45", "v46", "v47", "v48", "v49", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc", "memory"

constexpr int OUTER = 64, REPT = 8, BLOCK_INSTR = 8;

#define DEFINE_TEST(NAME, BODY)                                                                          \
  __global__ void NAME(unsigned long long* out, float* sink, float* gbuf) {                              \
    __shared__ float lds[16384];                                                                         \
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 0.5f;                             \
    __syncthreads();                                                                                     \
    const unsigned laddr = (unsigned)(uintptr_t)lds + (threadIdx.x & 63) * 16u + (threadIdx.x >> 6) * 4096u; \
    const unsigned long long gaddr = (unsigned long long)(gbuf + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4); \
    asm volatile(                                                                                        \
        "v_mov_b32 v48, %0\n v_mov_b32 v46, %1\n v_mov_b32 v47, %2\n"                                    \
        "v_mov_b32 v10, 1.0\n v_mov_b32 v11, 2.0\n v_mov_b32 v12, 0.5\n v_mov_b32 v13, 0.25\n"           \
        "v_mov_b32 v14, 1.0\n v_mov_b32 v15, 2.0\n v_mov_b32 v16, 0.5\n v_mov_b32 v17, 0.25\n"           \
        "v_mov_b32 v18, 1.0\n v_mov_b32 v19, 2.0\n v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.25\n"           \
        "v_mov_b32 v22, 1.0\n v_mov_b32 v23, 2.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.25\n"           \
        "v_mov_b32 v26, 1.0\n v_mov_b32 v27, 2.0\n v_mov_b32 v28, 0.5\n v_mov_b32 v29, 0.25\n"           \
        "v_mov_b32 v30, 1.0\n v_mov_b32 v31, 2.0\n v_mov_b32 v32, 0.5\n v_mov_b32 v33, 0.25\n"           \
        "v_mov_b32 v34, 1.0\n v_mov_b32 v35, 2.0\n v_mov_b32 v36, 0.5\n v_mov_b32 v37, 0.25\n"           \
        "v_mov_b32 v38, 1.0\n v_mov_b32 v39, 2.0\n v_mov_b32 v40, 0.5\n v_mov_b32 v41, 0.25\n"           \
        "v_mov_b32 v42, 0x12345\n v_mov_b32 v43, 0x54321\n v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n"    \
        :: "v"(laddr), "v"((unsigned)gaddr), "v"((unsigned)(gaddr >> 32)) : REG_CLOBBERS);               \
    const unsigned long long gaddr2 = (unsigned long long)(gbuf + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 512 + (threadIdx.x & 63)); \
    asm volatile("v_mov_b32 v34, %0\n v_mov_b32 v35, %1\n s_mov_b64 s[22:23], 1\n s_mov_b32 s20, 1.0\n s_mov_b32 s21, 1.0" :: "v"((unsigned)gaddr2), "v"((unsigned)(gaddr2 >> 32)) : REG_CLOBBERS); \
    unsigned long long t0, t1;                                                                           \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); \
    for (int it = 0; it < OUTER; it++) {                                                                 \
      asm volatile(".rept 8\n" BODY ".endr\n" ::: REG_CLOBBERS);                                         \
    }                                                                                                    \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); \
    float r;                                                                                             \
    asm volatile("v_add_f32 %0, v10, v12\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v26" : "=v"(r)::REG_CLOBBERS); \
    if (r == 12345.678f) sink[0] = r;                                                                    \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;    \
  }

// 8 independent instructions per body
DEFINE_TEST(t_add_f32, "v_add_f32 v10, v10, v44\n v_add_f32 v12, v12, v44\n v_add_f32 v14, v14, v44\n v_add_f32 v16, v16, v44\n"
                       "v_add_f32 v18, v18, v44\n v_add_f32 v20, v20, v44\n v_add_f32 v22, v22, v44\n v_add_f32 v24, v24, v44\n")
DEFINE_TEST(t_pk_add, "v_pk_add_f32 v[10:11], v[10:11], v[44:45]\n v_pk_add_f32 v[12:13], v[12:13], v[44:45]\n v_pk_add_f32 v[14:15], v[14:15], v[44:45]\n v_pk_add_f32 v[16:17], v[16:17], v[44:45]\n"
                      "v_pk_add_f32 v[18:19], v[18:19], v[44:45]\n v_pk_add_f32 v[20:21], v[20:21], v[44:45]\n v_pk_add_f32 v[22:23], v[22:23], v[44:45]\n v_pk_add_f32 v[24:25], v[24:25], v[44:45]\n")
DEFINE_TEST(t_pk_fma, "v_pk_fma_f32 v[10:11], v[10:11], v[44:45], v[26:27]\n v_pk_fma_f32 v[12:13], v[12:13], v[44:45], v[26:27]\n v_pk_fma_f32 v[14:15], v[14:15], v[44:45], v[26:27]\n v_pk_fma_f32 v[16:17], v[16:17], v[44:45], v[26:27]\n"
                      "v_pk_fma_f32 v[18:19], v[18:19], v[44:45], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[44:45], v[26:27]\n v_pk_fma_f32 v[22:23], v[22:23], v[44:45], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[44:45], v[26:27]\n")
DEFINE_TEST(t_fma_f32, "v_fma_f32 v10, v10, v44, v26\n v_fma_f32 v12, v12, v44, v26\n v_fma_f32 v14, v14, v44, v26\n v_fma_f32 v16, v16, v44, v26\n"
                       "v_fma_f32 v18, v18, v44, v26\n v_fma_f32 v20, v20, v44, v26\n v_fma_f32 v22, v22, v44, v26\n v_fma_f32 v24, v24, v44, v26\n")
DEFINE_TEST(t_cvt_sdwa, "v_cvt_f32_u32_sdwa v10, v42 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa v12, v43 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                        "v_cvt_f32_u32_sdwa v14, v42 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa v16, v43 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                        "v_cvt_f32_u32_sdwa v18, v42 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa v20, v43 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                        "v_cvt_f32_u32_sdwa v22, v42 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa v24, v43 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n")
DEFINE_TEST(t_cvt_plain, "v_cvt_f32_u32 v10, v42\n v_cvt_f32_u32 v12, v43\n v_cvt_f32_u32 v14, v42\n v_cvt_f32_u32 v16, v43\n"
                         "v_cvt_f32_u32 v18, v42\n v_cvt_f32_u32 v20, v43\n v_cvt_f32_u32 v22, v42\n v_cvt_f32_u32 v24, v43\n")
DEFINE_TEST(t_and_or, "v_and_or_b32 v10, v42, v43, v44\n v_and_or_b32 v12, v42, v43, v44\n v_and_or_b32 v14, v42, v43, v44\n v_and_or_b32 v16, v42, v43, v44\n"
                      "v_perm_b32 v18, v42, v43, v44\n v_perm_b32 v20, v42, v43, v44\n v_perm_b32 v22, v42, v43, v44\n v_perm_b32 v24, v42, v43, v44\n")
DEFINE_TEST(t_mov_dpp, "v_mov_b32_dpp v10, v26 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v12, v27 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                       "v_mov_b32_dpp v14, v28 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v16, v29 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                       "v_mov_b32_dpp v18, v30 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v20, v31 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                       "v_mov_b32_dpp v22, v32 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v24, v33 row_shr:1 row_mask:0xf bank_mask:0xf\n")
DEFINE_TEST(t_add_dpp_dep, "v_add_f32_dpp v10, v10, v10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp v10, v10, v10 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                           "v_add_f32_dpp v10, v10, v10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp v10, v10, v10 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                           "v_add_f32_dpp v12, v12, v12 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp v12, v12, v12 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                           "v_add_f32_dpp v12, v12, v12 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp v12, v12, v12 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n")
DEFINE_TEST(t_permlane32, "v_permlane32_swap_b32 v10, v11\n v_permlane32_swap_b32 v12, v13\n v_permlane32_swap_b32 v14, v15\n v_permlane32_swap_b32 v16, v17\n"
                          "v_permlane16_swap_b32 v18, v19\n v_permlane16_swap_b32 v20, v21\n v_permlane16_swap_b32 v22, v23\n v_permlane16_swap_b32 v24, v25\n")
DEFINE_TEST(t_cndmask, "v_cndmask_b32 v10, v26, v27, vcc\n v_cndmask_b32 v12, v26, v27, vcc\n v_cndmask_b32 v14, v26, v27, vcc\n v_cndmask_b32 v16, v26, v27, vcc\n"
                       "v_cndmask_b32 v18, v26, v27, vcc\n v_cndmask_b32 v20, v26, v27, vcc\n v_cndmask_b32 v22, v26, v27, vcc\n v_cndmask_b32 v24, v26, v27, vcc\n")
DEFINE_TEST(t_sqrt, "v_sqrt_f32 v10, v26\n v_sqrt_f32 v12, v27\n v_sqrt_f32 v14, v28\n v_sqrt_f32 v16, v29\n v_sqrt_f32 v18, v30\n v_sqrt_f32 v20, v31\n v_sqrt_f32 v22, v32\n v_sqrt_f32 v24, v33\n")
DEFINE_TEST(t_log, "v_log_f32 v10, v26\n v_log_f32 v12, v27\n v_log_f32 v14, v28\n v_log_f32 v16, v29\n v_log_f32 v18, v30\n v_log_f32 v20, v31\n v_log_f32 v22, v32\n v_log_f32 v24, v33\n")
DEFINE_TEST(t_add_f64, "v_add_f64 v[10:11], v[10:11], v[26:27]\n v_add_f64 v[12:13], v[12:13], v[26:27]\n v_add_f64 v[14:15], v[14:15], v[26:27]\n v_add_f64 v[16:17], v[16:17], v[26:27]\n"
                       "v_add_f64 v[18:19], v[18:19], v[26:27]\n v_add_f64 v[20:21], v[20:21], v[26:27]\n v_add_f64 v[22:23], v[22:23], v[26:27]\n v_add_f64 v[24:25], v[24:25], v[26:27]\n")
DEFINE_TEST(t_mul_lo, "v_mul_lo_u32 v10, v42, v43\n v_mul_lo_u32 v12, v42, v43\n v_mul_lo_u32 v14, v42, v43\n v_mul_lo_u32 v16, v42, v43\n"
                      "v_mad_u64_u32 v[18:19], s[20:21], v42, v43, v[26:27]\n v_mad_u64_u32 v[20:21], s[20:21], v42, v43, v[26:27]\n v_mad_u64_u32 v[22:23], s[20:21], v42, v43, v[26:27]\n v_mad_u64_u32 v[24:25], s[20:21], v42, v43, v[26:27]\n")
DEFINE_TEST(t_readlane, "v_readlane_b32 s20, v26, 63\n v_readlane_b32 s21, v27, 63\n v_readlane_b32 s22, v28, 63\n v_readlane_b32 s23, v29, 63\n"
                        "v_readlane_b32 s24, v30, 63\n v_readlane_b32 s25, v31, 63\n v_readlane_b32 s26, v32, 63\n v_readlane_b32 s27, v33, 63\n")
DEFINE_TEST(t_snop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
// LDS: laddr = per-lane 16-byte slots, one 4 KB window per wave
DEFINE_TEST(t_ds_read_b32, "ds_read_b32 v10, v48\n ds_read_b32 v12, v48 offset:1024\n ds_read_b32 v14, v48 offset:2048\n ds_read_b32 v16, v48 offset:3072\n"
                           "ds_read_b32 v18, v48 offset:4\n ds_read_b32 v20, v48 offset:1028\n ds_read_b32 v22, v48 offset:2052\n ds_read_b32 v24, v48 offset:3076\n s_waitcnt lgkmcnt(0)\n")
DEFINE_TEST(t_ds_read_b64, "ds_read_b64 v[10:11], v48\n ds_read_b64 v[12:13], v48 offset:1024\n ds_read_b64 v[14:15], v48 offset:2048\n ds_read_b64 v[16:17], v48 offset:3072\n"
                           "ds_read_b64 v[18:19], v48 offset:8\n ds_read_b64 v[20:21], v48 offset:1032\n ds_read_b64 v[22:23], v48 offset:2056\n ds_read_b64 v[24:25], v48 offset:3080\n s_waitcnt lgkmcnt(0)\n")
DEFINE_TEST(t_ds_read2_b64, "ds_read2_b64 v[10:13], v48 offset1:128\n ds_read2_b64 v[14:17], v48 offset0:1 offset1:129\n ds_read2_b64 v[18:21], v48 offset0:2 offset1:130\n ds_read2_b64 v[22:25], v48 offset0:3 offset1:131\n"
                            "ds_read2_b64 v[26:29], v48 offset0:4 offset1:132\n ds_read2_b64 v[30:33], v48 offset0:5 offset1:133\n ds_read2_b64 v[34:37], v48 offset0:6 offset1:134\n ds_read2_b64 v[38:41], v48 offset0:7 offset1:135\n s_waitcnt lgkmcnt(0)\n")
DEFINE_TEST(t_ds_read_b128, "ds_read_b128 v[10:13], v48\n ds_read_b128 v[14:17], v48 offset:1024\n ds_read_b128 v[18:21], v48 offset:2048\n ds_read_b128 v[22:25], v48 offset:3072\n"
                            "ds_read_b128 v[26:29], v48\n ds_read_b128 v[30:33], v48 offset:1024\n ds_read_b128 v[34:37], v48 offset:2048\n ds_read_b128 v[38:41], v48 offset:3072\n s_waitcnt lgkmcnt(0)\n")
DEFINE_TEST(t_ds_write_b128, "ds_write_b128 v48, v[10:13]\n ds_write_b128 v48, v[14:17] offset:1024\n ds_write_b128 v48, v[18:21] offset:2048\n ds_write_b128 v48, v[22:25] offset:3072\n"
                             "ds_write_b128 v48, v[26:29]\n ds_write_b128 v48, v[30:33] offset:1024\n ds_write_b128 v48, v[34:37] offset:2048\n ds_write_b128 v48, v[38:41] offset:3072\n s_waitcnt lgkmcnt(0)\n")
DEFINE_TEST(t_ds_write_b64, "ds_write_b64 v48, v[10:11]\n ds_write_b64 v48, v[12:13] offset:1024\n ds_write_b64 v48, v[14:15] offset:2048\n ds_write_b64 v48, v[16:17] offset:3072\n"
                            "ds_write_b64 v48, v[18:19] offset:8\n ds_write_b64 v48, v[20:21] offset:1032\n ds_write_b64 v48, v[22:23] offset:2056\n ds_write_b64 v48, v[24:25] offset:3080\n s_waitcnt lgkmcnt(0)\n")
DEFINE_TEST(t_ds_write2_b64, "ds_write2_b64 v48, v[10:11], v[12:13] offset1:128\n ds_write2_b64 v48, v[14:15], v[16:17] offset0:1 offset1:129\n ds_write2_b64 v48, v[18:19], v[20:21] offset0:2 offset1:130\n ds_write2_b64 v48, v[22:23], v[24:25] offset0:3 offset1:131\n"
                             "ds_write2_b64 v48, v[26:27], v[28:29] offset0:4 offset1:132\n ds_write2_b64 v48, v[30:31], v[32:33] offset0:5 offset1:133\n ds_write2_b64 v48, v[34:35], v[36:37] offset0:6 offset1:134\n ds_write2_b64 v48, v[38:39], v[40:41] offset0:7 offset1:135\n s_waitcnt lgkmcnt(0)\n")
DEFINE_TEST(t_bpermute, "ds_bpermute_b32 v10, v48, v26\n ds_bpermute_b32 v12, v48, v27\n ds_bpermute_b32 v14, v48, v28\n ds_bpermute_b32 v16, v48, v29\n"
                        "ds_bpermute_b32 v18, v48, v30\n ds_bpermute_b32 v20, v48, v31\n ds_bpermute_b32 v22, v48, v32\n ds_bpermute_b32 v24, v48, v33\n s_waitcnt lgkmcnt(0)\n")
// global stores to a private 16-byte slot per lane (L2-resident buffer): issue cost of the store path
DEFINE_TEST(t_gstore_b32, "global_store_dword v[46:47], v10, off\n global_store_dword v[46:47], v11, off offset:4\n global_store_dword v[46:47], v12, off offset:8\n global_store_dword v[46:47], v13, off offset:12\n"
                          "global_store_dword v[46:47], v14, off\n global_store_dword v[46:47], v15, off offset:4\n global_store_dword v[46:47], v16, off offset:8\n global_store_dword v[46:47], v17, off offset:12\n")
DEFINE_TEST(t_gstore_b128, "global_store_dwordx4 v[46:47], v[10:13], off\n global_store_dwordx4 v[46:47], v[14:17], off\n global_store_dwordx4 v[46:47], v[18:21], off\n global_store_dwordx4 v[46:47], v[22:25], off\n"
                           "global_store_dwordx4 v[46:47], v[26:29], off\n global_store_dwordx4 v[46:47], v[30:33], off\n global_store_dwordx4 v[46:47], v[34:37], off\n global_store_dwordx4 v[46:47], v[38:41], off\n")

DEFINE_TEST(t_cndmask_e64, "v_cndmask_b32_e64 v10, v26, v27, s[22:23]\n v_cndmask_b32_e64 v12, v26, v27, s[22:23]\n v_cndmask_b32_e64 v14, v26, v27, s[22:23]\n v_cndmask_b32_e64 v16, v26, v27, s[22:23]\n"
                           "v_cndmask_b32_e64 v18, v26, v27, s[22:23]\n v_cndmask_b32_e64 v20, v26, v27, s[22:23]\n v_cndmask_b32_e64 v22, v26, v27, s[22:23]\n v_cndmask_b32_e64 v24, v26, v27, s[22:23]\n")
DEFINE_TEST(t_bfi, "v_bfi_b32 v10, v42, v26, v27\n v_bfi_b32 v12, v42, v26, v27\n v_bfi_b32 v14, v42, v26, v27\n v_bfi_b32 v16, v42, v26, v27\n"
                   "v_bfi_b32 v18, v42, v26, v27\n v_bfi_b32 v20, v42, v26, v27\n v_bfi_b32 v22, v42, v26, v27\n v_bfi_b32 v24, v42, v26, v27\n")
DEFINE_TEST(t_add_sgpr, "v_add_f32 v10, s20, v10\n v_add_f32 v12, s20, v12\n v_add_f32 v14, s20, v14\n v_add_f32 v16, s20, v16\n"
                        "v_add_f32 v18, s20, v18\n v_add_f32 v20, s20, v20\n v_add_f32 v22, s20, v22\n v_add_f32 v24, s20, v24\n")
DEFINE_TEST(t_pk_mul_sgpr, "v_pk_mul_f32 v[10:11], v[10:11], s[20:21]\n v_pk_mul_f32 v[12:13], v[12:13], s[20:21]\n v_pk_mul_f32 v[14:15], v[14:15], s[20:21]\n v_pk_mul_f32 v[16:17], v[16:17], s[20:21]\n"
                           "v_pk_mul_f32 v[18:19], v[18:19], s[20:21]\n v_pk_mul_f32 v[20:21], v[20:21], s[20:21]\n v_pk_mul_f32 v[22:23], v[22:23], s[20:21]\n v_pk_mul_f32 v[24:25], v[24:25], s[20:21]\n")
DEFINE_TEST(t_mul_f32, "v_mul_f32 v10, v10, v44\n v_mul_f32 v12, v12, v44\n v_mul_f32 v14, v14, v44\n v_mul_f32 v16, v16, v44\n"
                       "v_mul_f32 v18, v18, v44\n v_mul_f32 v20, v20, v44\n v_mul_f32 v22, v22, v44\n v_mul_f32 v24, v24, v44\n")
DEFINE_TEST(t_mac_f32, "v_fmac_f32 v10, v26, v44\n v_fmac_f32 v12, v26, v44\n v_fmac_f32 v14, v26, v44\n v_fmac_f32 v16, v26, v44\n"
                       "v_fmac_f32 v18, v26, v44\n v_fmac_f32 v20, v26, v44\n v_fmac_f32 v22, v26, v44\n v_fmac_f32 v24, v26, v44\n")
DEFINE_TEST(t_ds_read_b32_32, "ds_read_b32 v10, v48\n ds_read_b32 v12, v48 offset:1024\n ds_read_b32 v14, v48 offset:2048\n ds_read_b32 v16, v48 offset:3072\n"
                           "ds_read_b32 v18, v48 offset:4\n ds_read_b32 v20, v48 offset:1028\n ds_read_b32 v22, v48 offset:2052\n ds_read_b32 v24, v48 offset:3076\n")
DEFINE_TEST(t_ds_read_b64_nw, "ds_read_b64 v[10:11], v48\n ds_read_b64 v[12:13], v48 offset:1024\n ds_read_b64 v[14:15], v48 offset:2048\n ds_read_b64 v[16:17], v48 offset:3072\n"
                           "ds_read_b64 v[18:19], v48 offset:8\n ds_read_b64 v[20:21], v48 offset:1032\n ds_read_b64 v[22:23], v48 offset:2056\n ds_read_b64 v[24:25], v48 offset:3080\n")
DEFINE_TEST(t_ds_write_b128_nw, "ds_write_b128 v48, v[10:13]\n ds_write_b128 v48, v[14:17] offset:1024\n ds_write_b128 v48, v[18:21] offset:2048\n ds_write_b128 v48, v[22:25] offset:3072\n"
                             "ds_write_b128 v48, v[26:29]\n ds_write_b128 v48, v[30:33] offset:1024\n ds_write_b128 v48, v[34:37] offset:2048\n ds_write_b128 v48, v[38:41] offset:3072\n")
DEFINE_TEST(t_ds_write_b64_nw, "ds_write_b64 v48, v[10:11]\n ds_write_b64 v48, v[12:13] offset:1024\n ds_write_b64 v48, v[14:15] offset:2048\n ds_write_b64 v48, v[16:17] offset:3072\n"
                            "ds_write_b64 v48, v[18:19] offset:8\n ds_write_b64 v48, v[20:21] offset:1032\n ds_write_b64 v48, v[22:23] offset:2056\n ds_write_b64 v48, v[24:25] offset:3080\n")
DEFINE_TEST(t_bpermute_nw, "ds_bpermute_b32 v10, v48, v26\n ds_bpermute_b32 v12, v48, v27\n ds_bpermute_b32 v14, v48, v28\n ds_bpermute_b32 v16, v48, v29\n"
                        "ds_bpermute_b32 v18, v48, v30\n ds_bpermute_b32 v20, v48, v31\n ds_bpermute_b32 v22, v48, v32\n ds_bpermute_b32 v24, v48, v33\n")
// coalesced dword stores: v[34:35] = base + lane*4 (set up below through gaddr2)
DEFINE_TEST(t_gstore_b32_coal, "global_store_dword v[34:35], v10, off\n global_store_dword v[34:35], v11, off offset:256\n global_store_dword v[34:35], v12, off offset:512\n global_store_dword v[34:35], v13, off offset:768\n"
                          "global_store_dword v[34:35], v14, off offset:1024\n global_store_dword v[34:35], v15, off offset:1280\n global_store_dword v[34:35], v16, off offset:1536\n global_store_dword v[34:35], v17, off offset:1792\n")
DEFINE_TEST(t_gstore_b32_coal_nt, "global_store_dword v[34:35], v10, off nt\n global_store_dword v[34:35], v11, off offset:256 nt\n global_store_dword v[34:35], v12, off offset:512 nt\n global_store_dword v[34:35], v13, off offset:768 nt\n"
                          "global_store_dword v[34:35], v14, off offset:1024 nt\n global_store_dword v[34:35], v15, off offset:1280 nt\n global_store_dword v[34:35], v16, off offset:1536 nt\n global_store_dword v[34:35], v17, off offset:1792 nt\n")

typedef void (*kern_t)(unsigned long long*, float*, float*);
// the same instructions with part of the wave enabled (exec narrowed around each block of 8; the two s_mov are counted as 0):
// does a lane-sparse VALU instruction issue faster, or slower?
DEFINE_TEST(t_pk_fma_e32, "s_mov_b64 exec, 0xffffffff\n" "v_pk_fma_f32 v[10:11], v[10:11], v[44:45], v[26:27]\n v_pk_fma_f32 v[12:13], v[12:13], v[44:45], v[26:27]\n v_pk_fma_f32 v[14:15], v[14:15], v[44:45], v[26:27]\n v_pk_fma_f32 v[16:17], v[16:17], v[44:45], v[26:27]\n"
    "v_pk_fma_f32 v[18:19], v[18:19], v[44:45], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[44:45], v[26:27]\n v_pk_fma_f32 v[22:23], v[22:23], v[44:45], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[44:45], v[26:27]\n" "s_mov_b64 exec, -1\n")
DEFINE_TEST(t_pk_fma_e16, "s_mov_b64 exec, 0xffff\n" "v_pk_fma_f32 v[10:11], v[10:11], v[44:45], v[26:27]\n v_pk_fma_f32 v[12:13], v[12:13], v[44:45], v[26:27]\n v_pk_fma_f32 v[14:15], v[14:15], v[44:45], v[26:27]\n v_pk_fma_f32 v[16:17], v[16:17], v[44:45], v[26:27]\n"
    "v_pk_fma_f32 v[18:19], v[18:19], v[44:45], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[44:45], v[26:27]\n v_pk_fma_f32 v[22:23], v[22:23], v[44:45], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[44:45], v[26:27]\n" "s_mov_b64 exec, -1\n")
DEFINE_TEST(t_pk_fma_e8, "s_mov_b64 exec, 0xff\n" "v_pk_fma_f32 v[10:11], v[10:11], v[44:45], v[26:27]\n v_pk_fma_f32 v[12:13], v[12:13], v[44:45], v[26:27]\n v_pk_fma_f32 v[14:15], v[14:15], v[44:45], v[26:27]\n v_pk_fma_f32 v[16:17], v[16:17], v[44:45], v[26:27]\n"
    "v_pk_fma_f32 v[18:19], v[18:19], v[44:45], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[44:45], v[26:27]\n v_pk_fma_f32 v[22:23], v[22:23], v[44:45], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[44:45], v[26:27]\n" "s_mov_b64 exec, -1\n")
DEFINE_TEST(t_pk_fma_e1, "s_mov_b64 exec, 1\n" "v_pk_fma_f32 v[10:11], v[10:11], v[44:45], v[26:27]\n v_pk_fma_f32 v[12:13], v[12:13], v[44:45], v[26:27]\n v_pk_fma_f32 v[14:15], v[14:15], v[44:45], v[26:27]\n v_pk_fma_f32 v[16:17], v[16:17], v[44:45], v[26:27]\n"
    "v_pk_fma_f32 v[18:19], v[18:19], v[44:45], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[44:45], v[26:27]\n v_pk_fma_f32 v[22:23], v[22:23], v[44:45], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[44:45], v[26:27]\n" "s_mov_b64 exec, -1\n")
DEFINE_TEST(t_add_e1, "s_mov_b64 exec, 1\n" "v_add_f32 v10, v10, v44\n v_add_f32 v12, v12, v44\n v_add_f32 v14, v14, v44\n v_add_f32 v16, v16, v44\n v_add_f32 v18, v18, v44\n v_add_f32 v20, v20, v44\n v_add_f32 v22, v22, v44\n v_add_f32 v24, v24, v44\n" "s_mov_b64 exec, -1\n")
DEFINE_TEST(t_add_e17, "s_mov_b64 exec, 0x10001\n" "v_add_f32 v10, v10, v44\n v_add_f32 v12, v12, v44\n v_add_f32 v14, v14, v44\n v_add_f32 v16, v16, v44\n v_add_f32 v18, v18, v44\n v_add_f32 v20, v20, v44\n v_add_f32 v22, v22, v44\n v_add_f32 v24, v24, v44\n" "s_mov_b64 exec, -1\n")
struct Test {
  const char* name;
  kern_t k;
};

int main() {
  unsigned long long* d_out;
  float *d_sink, *d_g;
  hipMalloc(&d_out, 256 * 16 * sizeof(unsigned long long));
  hipMalloc(&d_sink, 64);
  hipMalloc(&d_g, (size_t)64 << 20);  // >= 256 blocks x 16 waves x 512 floats (coalesced-store windows) and 256 x 1024 x 4 floats
  const Test tests[] = {{"v_add_f32", t_add_f32}, {"v_fma_f32", t_fma_f32}, {"v_pk_add_f32", t_pk_add}, {"v_pk_fma_f32", t_pk_fma},
                        {"v_pk_fma_f32, 32 lanes on", t_pk_fma_e32}, {"v_pk_fma_f32, 16 lanes on", t_pk_fma_e16}, {"v_pk_fma_f32, 8 lanes on", t_pk_fma_e8},
                        {"v_pk_fma_f32, 1 lane on", t_pk_fma_e1}, {"v_add_f32, 1 lane on", t_add_e1}, {"v_add_f32, lanes 0 and 16 on", t_add_e17},
                        {"v_cvt_f32_u32_sdwa", t_cvt_sdwa}, {"v_cvt_f32_u32", t_cvt_plain}, {"v_and_or/v_perm", t_and_or},
                        {"v_mov_b32_dpp", t_mov_dpp}, {"v_add_f32_dpp+s_nop1 (dep)", t_add_dpp_dep}, {"v_permlane32/16_swap", t_permlane32},
                        {"v_cndmask_b32 (vcc)", t_cndmask}, {"v_cndmask_b32_e64 (sgpr)", t_cndmask_e64}, {"v_bfi_b32", t_bfi}, {"v_add_f32 (sgpr src)", t_add_sgpr},
                        {"v_pk_mul_f32 (sgpr src)", t_pk_mul_sgpr}, {"v_mul_f32", t_mul_f32}, {"v_fmac_f32", t_mac_f32}, {"v_sqrt_f32", t_sqrt}, {"v_log_f32", t_log}, {"v_add_f64", t_add_f64},
                        {"v_mul_lo_u32/v_mad_u64_u32", t_mul_lo}, {"v_readlane_b32", t_readlane}, {"s_nop 0", t_snop},
                        {"ds_read_b32 (8+wait)", t_ds_read_b32}, {"ds_read_b64 (8+wait)", t_ds_read_b64},
                        {"ds_read2_b64 (8+wait)", t_ds_read2_b64}, {"ds_read_b128 (8+wait)", t_ds_read_b128},
                        {"ds_write_b128 (8+wait)", t_ds_write_b128}, {"ds_write_b64 (8+wait)", t_ds_write_b64},
                        {"ds_write2_b64 (8+wait)", t_ds_write2_b64}, {"ds_bpermute_b32 (8+wait)", t_bpermute},
                        {"ds_read_b32 (no wait)", t_ds_read_b32_32}, {"ds_read_b64 (no wait)", t_ds_read_b64_nw}, {"ds_write_b128 (no wait)", t_ds_write_b128_nw},
                        {"ds_write_b64 (no wait)", t_ds_write_b64_nw}, {"ds_bpermute_b32 (no wait)", t_bpermute_nw},
                        {"global_store_dword coalesced", t_gstore_b32_coal}, {"global_store_dword coalesced nt", t_gstore_b32_coal_nt},
                        {"global_store_dword", t_gstore_b32}, {"global_store_dwordx4", t_gstore_b128}};
  printf("%-30s %s\n", "instruction", "cycles per instruction per wave: 1 wave/SIMD | 2 waves/SIMD (=> per SIMD) | 4 waves/SIMD (=> per SIMD)");
  const double n = (double)OUTER * REPT * BLOCK_INSTR;
  for (const Test& t : tests) {
    double res[3];
    int wi = 0;
    for (int waves : {4, 8, 16}) {
      hipLaunchKernelGGL(t.k, dim3(256), dim3(64 * waves), 0, 0, d_out, d_sink, d_g);
      hipLaunchKernelGGL(t.k, dim3(256), dim3(64 * waves), 0, 0, d_out, d_sink, d_g);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(256 * waves);
      hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (auto v : h) s += (double)v;
      res[wi++] = s / h.size() / n;
    }
    printf("%-30s %7.2f | %7.2f (%6.2f) | %7.2f (%6.2f)\n", t.name, res[0], res[1], res[1] / 2, res[2], res[2] / 4);
  }
  return 0;
}
