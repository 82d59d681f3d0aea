// c3_gather_probe.hip -- the bounded experiment VERDICT r5 (next 5) asks for: C3's complex rows (dispersion phase) spend 24 % of
// their cycles in the gather -- 72 LDS reads per lane and row (32 gathered floats, 32 phasors of 8 bytes, 8 table reads) -- because a
// wave that holds a whole 2048-point row in 64 lanes has no registers left for phasors and addresses.  Prototype of the
// alternative, gather + phase multiply + first in-register radix pass ONLY:
//   mode 0  one wave per row, as fused_kernel's 2048-point complex plan does it today: 32 points per lane, addresses and phasors
//           re-read from LDS every row, radix-32 butterfly in registers;
//   mode 1  two waves per row (T = 128): 16 points per lane, the 16 phasors and the gather addresses RESIDENT in registers,
//           radix-16 butterfly in registers; a row is two such half-rows (the cross-wave exchange that would follow is the LDS
//           round trip the plan has anyway, and is left out of both modes).
// Same LDS footprint and waves per CU as C3 (8 waves, 8.3 KB row buffer each, 16 KB of phasors + 4 KB of addresses shared), same
// number of rows; prints wave-cycles per row for both.  Stop rule: mode 1 has to save >= 10 % of C3's 15 370 cycles per row,
// i.e. >= 1540 cycles per row on this phase, before a two-wave plan is worth building.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I fdoct_amd/csrc tools/ubench/c3_gather_probe.hip -o tools/ubench/c3_gather_probe
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <vector>

#include "fdoct_fft_reg.h"

using namespace fdoct;

constexpr int N = 2048, WAVES = 8;
constexpr int ROWF = N + 4 * 16;   // staged samples of a row (padded like the kernel's staging buffer)

template <int MODE>
__global__ __launch_bounds__(64 * WAVES) void probe(const float2* g_ph, const uint16_t* g_gi, int rows_per_wave, unsigned long long* cyc, float* sink) {
  extern __shared__ __align__(16) unsigned char sm[];
  float2* s_ph = reinterpret_cast<float2*>(sm);                       // [N] phasors
  uint16_t* s_gi = reinterpret_cast<uint16_t*>(s_ph + N);             // [N] gather sources (float index into the row buffer)
  float* rowbuf = reinterpret_cast<float*>(s_gi + N) + (threadIdx.x >> 6) * ROWF;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    s_ph[i] = g_ph[i];
    s_gi[i] = g_gi[i];
  }
  for (int i = lane; i < ROWF; i += 64) rowbuf[i] = 0.001f * (float)((i * 37 + wave) & 1023);
  __syncthreads();
  float acc = 0.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  if constexpr (MODE == 0) {
    // lane l holds points l + 64 m, m < 32 (the plan's first pass is a radix 32 over m)
    for (int r = 0; r < rows_per_wave; r++) {
      v2f z[32];
      // 8 table reads: four packed 16-bit sources per 8-byte read -> the kernel's layout is [m/4][lane][m%4]
      const uint2* gt = reinterpret_cast<const uint2*>(s_gi) + lane;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint2 w = gt[64 * q];
        const unsigned a0 = w.x & 0xffffu, a1 = w.x >> 16, a2 = w.y & 0xffffu, a3 = w.y >> 16;
        const float y0 = rowbuf[a0], y1 = rowbuf[a1], y2 = rowbuf[a2], y3 = rowbuf[a3];
        const float2 p0 = s_ph[lane + 64 * (4 * q)], p1 = s_ph[lane + 64 * (4 * q + 1)], p2 = s_ph[lane + 64 * (4 * q + 2)], p3 = s_ph[lane + 64 * (4 * q + 3)];
        z[4 * q + 0] = mk(y0 * p0.x, y0 * p0.y);
        z[4 * q + 1] = mk(y1 * p1.x, y1 * p1.y);
        z[4 * q + 2] = mk(y2 * p2.x, y2 * p2.y);
        z[4 * q + 3] = mk(y3 * p3.x, y3 * p3.y);
      }
      fft_reg<32, true>(z);
#pragma unroll
      for (int m = 0; m < 32; m++) acc += z[m].x + z[m].y;
      asm volatile("" ::: "memory");   // the next row's reads are not merged with this one's
    }
  } else {
    // two waves share a row: this wave's lanes hold points (lane + 64 h) + 128 m, m < 16, h = wave & 1; phasors and sources resident
    const int h = wave & 1;
    v2f ph[16];
    unsigned ad[16];
#pragma unroll
    for (int m = 0; m < 16; m++) {
      const int e = lane + 64 * h + 128 * m;
      ph[m] = mk(s_ph[e].x, s_ph[e].y);
      ad[m] = s_gi[e];
    }
    for (int r = 0; r < 2 * rows_per_wave; r++) {   // (two waves per row: each wave sees twice as many half-rows)
      v2f z[16];
#pragma unroll
      for (int m = 0; m < 16; m++) {
        const float y = rowbuf[ad[m]];
        z[m] = mk(y * ph[m].x, y * ph[m].y);
      }
      fft_reg<16, true>(z);
#pragma unroll
      for (int m = 0; m < 16; m++) acc += z[m].x + z[m].y;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int m = 0; m < 16; m++) asm volatile("" : "+v"(ad[m]));   // (keeps the addresses opaque: no hoisting of the row buffer reads)
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  std::vector<float2> ph(N);
  std::vector<uint16_t> gi(N);
  for (int q = 0; q < N; q++) {
    const double x = (q - N / 2) / (double)(N / 2), phi = 20.0 * x * x + 5.0 * x * x * x;
    ph[q] = make_float2((float)cos(phi), (float)sin(phi));
  }
  // the kernel's gather: data_ylin[q] = s[idx[q]], idx non-increasing in q over a 2048-sample row; laid out [m/4][lane][m%4] for mode 0
  std::vector<uint16_t> src(N);
  for (int q = 0; q < N; q++) {
    int i = N - 1 - (int)(q * 0.96);
    if (i < 0) i = 0;
    src[q] = (uint16_t)(i + 4 * (i / 128));
  }
  std::vector<uint16_t> gi0(N), gi1(N);
  for (int l = 0; l < 64; l++)
    for (int m = 0; m < 32; m++) gi0[((m >> 2) * 64 + l) * 4 + (m & 3)] = src[l + 64 * m];
  for (int q = 0; q < N; q++) gi1[q] = src[q];
  float2* d_ph;
  uint16_t *d_g0, *d_g1;
  unsigned long long* d_c;
  float* d_s;
  const int blocks = 256, rows = 400;
  hipMalloc(&d_ph, N * 8); hipMalloc(&d_g0, N * 2); hipMalloc(&d_g1, N * 2); hipMalloc(&d_c, blocks * WAVES * 8); hipMalloc(&d_s, 4);
  hipMemcpy(d_ph, ph.data(), N * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_g0, gi0.data(), N * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_g1, gi1.data(), N * 2, hipMemcpyHostToDevice);
  const size_t lds = N * 8 + N * 2 + (size_t)WAVES * ROWF * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<unsigned long long> c(blocks * WAVES);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 2; mode++) {
      hipEventRecord(e0);
      if (mode == 0)
        hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64 * WAVES), lds, 0, d_ph, d_g0, rows, d_c, d_s);
      else
        hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64 * WAVES), lds, 0, d_ph, d_g1, rows, d_c, d_s);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost);
      double sum = 0;
      for (auto v : c) sum += (double)v;
      // wave-cycles per ROW: mode 0 one wave spends its cycles on `rows` rows; mode 1 a PAIR of waves spends 2 x the cycles on 2 x rows rows
      const double per_row = sum / c.size() / rows;
      printf("rep %d mode %d (%s): %.0f wave-cycles per row (gather + phase multiply + first radix pass), kernel %.3f ms for %d rows\n", rep, mode,
             mode == 0 ? "one wave per row, tables from LDS" : "two waves per row, phasors and sources resident", per_row, ms, blocks * WAVES * rows);
    }
  return 0;
}
