// store_completion.hip -- how long do the transposed store's scattered 16-byte stores take to be ACKNOWLEDGED?  (EXPERIMENTS.md
// section 5: C2 in D x H is bound by store completion -- a wave's vector-memory operations return in order through one counter, so
// the next row's prefetched samples cannot land before the write-out stores issued ahead of them are done.)  Eight waves per CU on
// every CU; each wave loops over "rows": stream-read 4 KB of input (as the chain does), spin ~one row's time, then issue one write-out
// step's stores -- 4 x 16 bytes per lane into 64-byte segments of a D x H image, the pattern of fused_kernel's tro_step -- and time
// s_waitcnt vmcnt(0) from the moment the stores are issued.  Prints percentiles of that completion time in microseconds, for HBM-sized
// targets (every tile its own place in a 1 GiB image) and for L2-resident ones (all tiles into the same 64 KB).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/store_completion.hip -o tools/ubench/store_completion
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int POL>
__global__ __launch_bounds__(512) void probe(const u4* in, float* out, int H, int D, int rows, int resident, int row_cycles, unsigned* lat, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 8 + wave, nwaves = gridDim.x * 8;
  const int rq = lane & 3, dg = lane >> 2;
  float acc = 0.f;
  const size_t in_words = (size_t)1 << 26;   // 1 GiB of 16-byte words
  for (int r = 0; r < rows; r++) {
    // the chain's input traffic: 4 KB per row and wave, streaming
    const u4 v = __builtin_nontemporal_load(in + ((size_t)(r * nwaves + gw) * 64 + lane) % in_words);
    acc += (float)(v.x + v.y + v.z + v.w);
    // one row's compute
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)row_cycles) __builtin_amdgcn_s_sleep(8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // one write-out step: bins s0 .. s0 + 63 of a 16-row tile; lane (dg, rq): rows 4 rq .. + 3, bins 4 dg .. + 3
    const int tile = resident ? 0 : (r * nwaves + gw) % (H / 16 * 1024);   // tiles of a 1024-frame image
    const int g = tile / (H / 16), r0 = (tile % (H / 16)) * 16;
    const int s0 = resident ? 0 : ((r * 7 + wave) % (D / 64)) * 64;
    float* base = out + (resident ? 0 : ((size_t)g * D) * H + r0);
    const int Hs = resident ? 16 : H;
    const f4 w = {acc, acc + 1.f, acc + 2.f, acc + 3.f};
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
      // POL 4: the ROW-MAJOR pattern for comparison (a wave writes 1 KB contiguous per instruction)
      f4* p = POL == 4 ? reinterpret_cast<f4*>(out + ((size_t)((r * nwaves + gw) % (1000 * 1000)) * D + 256 * bb) + 4 * lane)
                       : reinterpret_cast<f4*>(base + (size_t)(s0 + 4 * dg + bb) * Hs + 4 * rq);
      if (POL == 0 || POL == 4) *p = w;
      else if (POL == 1) __builtin_nontemporal_store(w, p);
      else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
      else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && r >= 8) lat[(size_t)gw * rows + r] = (unsigned)(t2 - t1);
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int H = 1000, D = 1024, rows = 200, blocks = 256;
  u4* d_in;
  float *d_out, *d_sink;
  unsigned* d_lat;
  hipMalloc(&d_in, (size_t)1 << 30);
  hipMalloc(&d_out, (size_t)1024 * D * H * 4);
  hipMalloc(&d_sink, 4);
  hipMalloc(&d_lat, (size_t)blocks * 8 * rows * 4);
  hipMemset(d_in, 1, (size_t)1 << 30);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const char* pol_name[] = {"plain stores", "nt stores", "sc0 sc1 stores", "sc1 stores", "ROW-MAJOR pattern (1 KB contiguous per instruction), plain stores"};
  for (int pol = 0; pol < 5; pol++)
    for (int resident = 0; resident < (pol == 0 ? 2 : 1); resident++)
      for (int row_us : {4, 1}) {
        hipMemset(d_lat, 0, (size_t)blocks * 8 * rows * 4);
        const int row_cycles = row_us * 100;   // __builtin_amdgcn_s_memrealtime() = s_memtime: 100 MHz on this part
        auto go = [&](auto k) { hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d_in, d_out, H, D, rows, resident, row_cycles, d_lat, d_sink); };
        switch (pol) {
          case 0: go(probe<0>); break;
          case 1: go(probe<1>); break;
          case 2: go(probe<2>); break;
          case 3: go(probe<3>); break;
          default: go(probe<4>); break;
        }
        hipDeviceSynchronize();
        std::vector<unsigned> lat((size_t)blocks * 8 * rows);
        hipMemcpy(lat.data(), d_lat, lat.size() * 4, hipMemcpyDeviceToHost);
        std::vector<unsigned> v;
        for (unsigned x : lat)
          if (x) v.push_back(x);
        std::sort(v.begin(), v.end());
        auto pct = [&](double p) { return v.empty() ? 0.0 : v[(size_t)(p * (v.size() - 1))] / 100.0; };   // 100 MHz ticks -> us
        printf("%s, %s targets, %d us of compute per row: completion of one step's stores (4 x 16 B per lane), us: median %.2f  p90 %.2f  p99 %.2f  max %.2f  (%zu samples)\n",
               pol_name[pol], resident ? "L2-resident (64 KB)" : "HBM-sized (4 GB image)", row_us, pct(0.5), pct(0.9), pct(0.99), pct(1.0), v.size());
      }
  return 0;
}
