// Micro-benchmark: how fast does the chip take the write pattern of a FUSED transposed B-scan store?  The reference's
// bscan is D x H (depth-major): 16 consecutive A-scans of one frame give, per depth bin, 16 consecutive floats = 64 bytes,
// and consecutive bins are H*4 bytes apart.  Each workgroup writes tiles of [D bins] x [R rows] that way (R = 8, 16, 32:
// 32-, 64-, 128-byte segments), non-temporal, against plain row-major 256-byte-per-wave stores of the same volume.
// build: hipcc -O3 --offload-arch=gfx950 -o scatter64 scatter64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));

// one frame = D x H floats; tile t of frame f covers rows t*R .. t*R+R-1 (H is a multiple of R here)
template <int R>
__global__ void __launch_bounds__(512) k_tiles(float* out, int D, int H, int frames) {
  const int tiles_per_frame = H / R;
  const long long ntiles = (long long)frames * tiles_per_frame;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int f = (int)(tile / tiles_per_frame), t = (int)(tile - (long long)f * tiles_per_frame);
    float* base = out + (size_t)f * D * H + (size_t)t * R;
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
      f4v* p = reinterpret_cast<f4v*>(base + (size_t)k * H);
      const f4v v = {(float)k, (float)t, (float)f, 1.f};
#pragma unroll
      for (int q = 0; q < R / 4; q++) __builtin_nontemporal_store(v, p + q);
    }
  }
}

// the cooperative mapping: R/4 lanes share one bin's segment (16 bytes each), a wave instruction covers 64*4/R bins
template <int R>
__global__ void __launch_bounds__(512) k_tiles_coop(float* out, int D, int H, int frames) {
  constexpr int LPB = R / 4;  // lanes per bin
  const int tiles_per_frame = H / R;
  const long long ntiles = (long long)frames * tiles_per_frame;
  const int part = threadIdx.x % LPB, kb = threadIdx.x / LPB;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int f = (int)(tile / tiles_per_frame), t = (int)(tile - (long long)f * tiles_per_frame);
    float* base = out + (size_t)f * D * H + (size_t)t * R + 4 * part;
    for (int k = kb; k < D; k += blockDim.x / LPB) {
      const f4v v = {(float)k, (float)t, (float)f, 1.f};
      __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(base + (size_t)k * H));
    }
  }
}

// Round 3: the same cooperative pattern as the fused kernel issues it (fused_kernel TRO: write-back stores, the workgroups of
// one XCD on neighbouring tiles so that the halves of a 128-byte line meet in one L2), from 1 .. 8 waves per workgroup.
template <int R, bool NT, bool PAIR>
__global__ void __launch_bounds__(512) k_tiles_coop2(float* out, int D, int H, int frames) {
  constexpr int LPB = R / 4;  // lanes per bin
  const int tiles_per_frame = H / R;
  const long long ntiles = (long long)frames * tiles_per_frame;
  const int part = threadIdx.x % LPB, kb = threadIdx.x / LPB;
  const unsigned blk = blockIdx.x, grid = gridDim.x;
  const unsigned bperm = PAIR ? (blk & 7u) * (grid >> 3) + (blk >> 3) : blk;
  for (long long tile = bperm; tile < ntiles; tile += grid) {
    const int f = (int)(tile / tiles_per_frame), t = (int)(tile - (long long)f * tiles_per_frame);
    float* base = out + (size_t)f * D * H + (size_t)t * R + 4 * part;
    for (int k = kb; k < D; k += blockDim.x / LPB) {
      const f4v v = {(float)k, (float)t, (float)f, 1.f};
      if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(base + (size_t)k * H));
      else *reinterpret_cast<f4v*>(base + (size_t)k * H) = v;
    }
  }
}

__global__ void __launch_bounds__(512) k_rows(float* out, int D, long long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (long long r = (long long)blockIdx.x * nw + wave; r < rows; r += (long long)gridDim.x * nw) {
    float* p = out + (size_t)r * D + lane;
#pragma unroll
    for (int m = 0; m < 16; m++) __builtin_nontemporal_store((float)m, p + 64 * m);
  }
}

template <typename F>
static double time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; i++) launch();
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const int D = 1024, H = 960, frames = 272;  // 960 = 15 * 64 rows: every tile is whole
  const size_t bytes = (size_t)frames * D * H * 4;
  float* out;
  hipMalloc(&out, bytes);
  const double gb = bytes / 1e9;
  double ms = time_ms([&] { hipLaunchKernelGGL(k_rows, dim3(256), dim3(512), 0, 0, out, D, (long long)frames * H); }, 200);
  printf("row-major, 256 B per wave store    %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL(k_tiles<8>, dim3(256), dim3(512), 0, 0, out, D, H, frames); }, 200);
  printf("depth-major,  8 rows = 32 B segs   %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL(k_tiles<16>, dim3(256), dim3(512), 0, 0, out, D, H, frames); }, 200);
  printf("depth-major, 16 rows = 64 B segs   %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL(k_tiles<32>, dim3(256), dim3(512), 0, 0, out, D, H, frames); }, 200);
  printf("depth-major, 32 rows = 128 B segs  %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL(k_tiles<16>, dim3(1024), dim3(512), 0, 0, out, D, H, frames); }, 200);
  printf("depth-major, 16 rows, 1024 blocks  %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL(k_tiles_coop<8>, dim3(256), dim3(512), 0, 0, out, D, H, frames); }, 200);
  printf("depth-major,  8 rows, 2 lanes per 32 B segment   %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL(k_tiles_coop<16>, dim3(256), dim3(512), 0, 0, out, D, H, frames); }, 200);
  printf("depth-major, 16 rows, 4 lanes per 64 B segment   %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL(k_tiles_coop<32>, dim3(256), dim3(512), 0, 0, out, D, H, frames); }, 200);
  printf("depth-major, 32 rows, 8 lanes per 128 B segment  %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL(k_tiles_coop<64>, dim3(256), dim3(512), 0, 0, out, D, H, frames); }, 200);
  printf("depth-major, 64 rows, 16 lanes per 256 B segment %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  printf("-- round 3: write-back stores, XCD pairing, waves per workgroup, aligned rows (H = 1024: frames shrink to fit)\n");
  const int H2 = 1024, frames2 = (int)(bytes / ((size_t)D * H2 * 4));
  const double gb2 = (double)frames2 * D * H2 * 4 / 1e9;
#define RUN(R, NT, PAIR, THREADS, HH, FR, GB, label)                                                                            \
  ms = time_ms([&] { hipLaunchKernelGGL((k_tiles_coop2<R, NT, PAIR>), dim3(256), dim3(THREADS), 0, 0, out, D, HH, FR); }, 100); \
  printf("%-72s %.3f ms  %.0f GB/s = %.1f GB/s per CU\n", label, ms, GB / ms * 1e3, GB / ms * 1e3 / 256);
  RUN(16, true, false, 512, H, frames, gb, "16 rows, nt, 8 waves");
  RUN(16, false, false, 512, H, frames, gb, "16 rows, write-back, 8 waves");
  RUN(16, false, true, 512, H, frames, gb, "16 rows, write-back, neighbouring tiles per XCD, 8 waves");
  RUN(16, false, true, 256, H, frames, gb, "16 rows, write-back, neighbouring tiles per XCD, 4 waves");
  RUN(16, false, true, 128, H, frames, gb, "16 rows, write-back, neighbouring tiles per XCD, 2 waves");
  RUN(16, false, true, 64, H, frames, gb, "16 rows, write-back, neighbouring tiles per XCD, 1 wave");
  RUN(32, false, true, 512, H, frames, gb, "32 rows, write-back, neighbouring tiles per XCD, 8 waves");
  RUN(32, false, true, 64, H, frames, gb, "32 rows, write-back, neighbouring tiles per XCD, 1 wave");
  RUN(16, false, true, 512, H2, frames2, gb2, "H = 1024: 16 rows, write-back, neighbouring tiles per XCD, 8 waves");
  RUN(16, false, true, 64, H2, frames2, gb2, "H = 1024: 16 rows, write-back, neighbouring tiles per XCD, 1 wave");
  RUN(32, false, true, 512, H2, frames2, gb2, "H = 1024: 32 rows, write-back, neighbouring tiles per XCD, 8 waves");
  RUN(32, false, true, 64, H2, frames2, gb2, "H = 1024: 32 rows, write-back, neighbouring tiles per XCD, 1 wave");
  RUN(64, false, true, 512, H2, frames2, gb2, "H = 1024: 64 rows, write-back, neighbouring tiles per XCD, 8 waves");
  return 0;
}
