// CPU check of fdoct_amd/csrc/fdoct_hostcopy.h (the copy threads behind fdoct_process's pinned staging slots): every byte of
// a strided 2-D copy and of a flat copy arrives, nothing outside the rows is written, for 1 ... 7 threads and job after job on
// one pool.  Built by tests/test_hostcopy.py with -fsanitize=thread (and once with address,undefined).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../fdoct_amd/csrc/fdoct_hostcopy.h"

int main() {
  using fdoct_impl::HostCopyPool;
  unsigned seed = 12345u;
  auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  long long jobs = 0;
  for (int threads : {1, 2, 3, 7}) {
    HostCopyPool pool(threads);
    if (pool.threads() != threads) { std::printf("pool of %d has %d\n", threads, pool.threads()); return 1; }
    for (int it = 0; it < 24; it++) {
      const size_t width = 1 + rnd() % 5000, rows = 1 + rnd() % (it % 3 ? 700 : 40);
      const size_t spitch = width + (it & 1 ? rnd() % 64 : 0), dpitch = width + (it & 2 ? rnd() % 64 : 0);
      std::vector<unsigned char> src(spitch * rows), dst(dpitch * rows, 0xA5);
      for (auto& b : src) b = (unsigned char)rnd();
      pool.copy2d(dst.data(), dpitch, src.data(), spitch, width, rows);
      for (size_t r = 0; r < rows; r++) {
        for (size_t i = 0; i < width; i++)
          if (dst[r * dpitch + i] != src[r * spitch + i]) { std::printf("copy2d: row %zu byte %zu differs\n", r, i); return 1; }
        for (size_t i = width; i < dpitch; i++)
          if (dst[r * dpitch + i] != 0xA5) { std::printf("copy2d: wrote into the pad of row %zu\n", r); return 1; }
      }
      jobs++;
    }
    for (size_t bytes : {(size_t)0, (size_t)1, (size_t)65535, (size_t)65536, (size_t)(3 << 20) + 17, (size_t)(9 << 20)}) {
      std::vector<unsigned char> src(bytes + 1), dst(bytes + 1, 0x5A);
      for (auto& b : src) b = (unsigned char)rnd();
      pool.copy(dst.data(), src.data(), bytes);
      for (size_t i = 0; i < bytes; i++)
        if (dst[i] != src[i]) { std::printf("copy: byte %zu of %zu differs\n", i, bytes); return 1; }
      if (dst[bytes] != 0x5A) { std::printf("copy: wrote past %zu bytes\n", bytes); return 1; }
      jobs++;
    }
  }
  std::printf("ok %lld jobs\n", jobs);
  return 0;
}
