// rw_mix.hip -- what does the memory system sustain for C2's traffic in the reference's D x H layout?  (EXPERIMENTS.md section 5.)
// Eight waves per CU on every CU, no arithmetic: each wave streams 4 KB of "samples" per row (four 16-byte loads per lane, the
// chain's input) and writes 4 KB of "depth profile" per row, either ROW-MAJOR (four instructions of 1 KB contiguous) or as the
// transposed store does it (four instructions of 16-byte stores into sixteen 64-byte segments each, H * 4 bytes apart, tiles of 16
// rows x 64 bins spread over the image) -- the next row's loads are in flight while a row's stores are issued.  Prints
// (bytes read + bytes written) / time.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/rw_mix.hip -o tools/ubench/rw_mix
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int PATTERN, bool READS>   // 0: row-major, 1: transposed (64-byte segments), 2: transposed with 32-row tiles (128-byte segments),
                                     // 3 / 4: 64-byte segments, a wave writes the two halves of a 128-byte line in CONSECUTIVE rows (nt / write-back stores)
__global__ __launch_bounds__(512) void mix(const u4* in, float* out, int H, int D, int rows, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 8 + wave, nwaves = gridDim.x * 8;
  constexpr int RQN = PATTERN == 2 ? 8 : 4;       // lanes side by side along H: 4 x 16 B = 64-byte segments, 8 x 16 B = 128-byte
  const int rq = lane % RQN, dg = lane / RQN;
  const size_t in_words = (size_t)1 << 26;        // 1 GiB of 16-byte words
  auto load_row = [&](int r, u4* v) {
#pragma unroll
    for (int c = 0; c < 4; c++) v[c] = READS ? __builtin_nontemporal_load(in + ((size_t)(r * nwaves + gw) * 256 + 64 * c + lane) % in_words) : u4{1u, 2u, 3u, 4u};
  };
  u4 cur[4], nxt[4];
  load_row(0, cur);
  float acc = 0.f;
  for (int r = 0; r < rows; r++) {
    load_row(r + 1, nxt);
#pragma unroll
    for (int c = 0; c < 4; c++) acc += (float)(cur[c].x ^ cur[c].w);
    const f4 w = {acc, acc + 1.f, acc + 2.f, acc + 3.f};
    const size_t unit = (size_t)r * nwaves + gw;   // this wave's r-th 4 KB of output
    if (PATTERN == 0) {
#pragma unroll
      for (int bb = 0; bb < 4; bb++) __builtin_nontemporal_store(w, reinterpret_cast<f4*>(out + (unit % ((size_t)1000 * 1000)) * D + 256 * bb + 4 * lane));
    } else {
      // a step of the tile write-out: bins s0 .. of tile (g, r0); 64 / RQN bin groups of 4 bins, RQN row quads
      constexpr int TR = 4 * RQN, BINS = 4 * (64 / RQN);          // 16 rows x 64 bins, or 32 rows x 32 bins: 4 KB either way
      const size_t tiles_per_frame = (size_t)(H / TR) * (D / BINS);
      size_t t = unit % (tiles_per_frame * 1000);
      if (PATTERN >= 3) t = (((size_t)(r >> 1) * nwaves + gw) * 2 + (r & 1)) % (tiles_per_frame * 1000);   // tiles 2 p, 2 p + 1: vertical neighbours
      const size_t g = t / tiles_per_frame, tt = t % tiles_per_frame;
      const int r0 = (int)(tt % (H / TR)) * TR, s0 = (int)(tt / (H / TR)) * BINS;
      float* base = out + (g * D) * (size_t)H + r0;
#pragma unroll
      for (int bb = 0; bb < 4; bb++) {
        f4* q = reinterpret_cast<f4*>(base + (size_t)(s0 + 4 * dg + bb) * H + 4 * rq);
        if (PATTERN == 4) *q = w; else __builtin_nontemporal_store(w, q);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) cur[c] = nxt[c];
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int H = 992, D = 1024, rows = 4000, blocks = 256;   // (H a multiple of 32 so that both tile heights divide it)
  u4* d_in;
  float *d_out, *d_sink;
  (void)hipMalloc(&d_in, (size_t)1 << 30);
  (void)hipMalloc(&d_out, (size_t)1000 * D * 1000 * 4);
  (void)hipMalloc(&d_sink, 4);
  (void)hipMemset(d_in, 1, (size_t)1 << 30);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const char* names[] = {"row-major stores (1 KB contiguous per instruction)", "transposed store, 64-byte segments (16-row tiles)", "transposed store, 128-byte segments (32-row tiles)",
                         "64-byte segments, vertical neighbours in consecutive rows, nt", "64-byte segments, vertical neighbours in consecutive rows, write-back"};
  for (int reads = 1; reads >= 0; reads--)
    for (int p = 0; p < 5; p++) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0, 0);
        auto go = [&](auto k) { hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d_in, d_out, H, D, rows, d_sink); };
        if (reads) {
          if (p == 0) go(mix<0, true>); else if (p == 1) go(mix<1, true>); else if (p == 2) go(mix<2, true>); else if (p == 3) go(mix<3, true>); else go(mix<4, true>);
        } else {
          if (p == 0) go(mix<0, false>); else if (p == 1) go(mix<1, false>); else if (p == 2) go(mix<2, false>); else if (p == 3) go(mix<3, false>); else go(mix<4, false>);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double bytes = (double)blocks * 8 * rows * 4096.0 * (reads ? 2.0 : 1.0);
      printf("%-28s %-72s %.3f ms  %.2f TB/s %s\n", reads ? "4 KB read + 4 KB written:" : "4 KB written (no reads):", names[p], best, bytes / best * 1e-9,
             reads ? "(reads + writes)" : "");
    }
  return 0;
}
