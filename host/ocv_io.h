// ocv_io.h -- the reference's raw cv::Mat dump (".ocv", BscanFFTspinj.cpp:672-715 matwrite/matread):
// int32 rows, cols, OpenCV type code (depth + ((channels-1) << 3)), channels, then the row-major payload.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

struct OcvMat {
  int rows = 0, cols = 0, depth = 0, channels = 1;  // depth: 0 = u8, 2 = u16, 5 = f32, 6 = f64 (OpenCV codes)
  std::vector<unsigned char> data;
};

inline int ocv_elem_size(int depth) {
  static const int sz[7] = {1, 1, 2, 2, 4, 4, 8};
  return (depth >= 0 && depth < 7) ? sz[depth] : 0;
}

inline bool ocv_read(const std::string& path, OcvMat* m) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  int32_t hdr[4];
  bool ok = std::fread(hdr, sizeof(int32_t), 4, f) == 4;
  if (ok) {
    m->rows = hdr[0];
    m->cols = hdr[1];
    m->depth = hdr[2] & 7;
    m->channels = hdr[3];
    const size_t n = (size_t)m->rows * m->cols * m->channels * ocv_elem_size(m->depth);
    ok = m->rows > 0 && m->cols > 0 && ((hdr[2] >> 3) + 1) == hdr[3] && ocv_elem_size(m->depth) > 0;
    if (ok) {
      m->data.resize(n);
      ok = std::fread(m->data.data(), 1, n, f) == n;
    }
  }
  std::fclose(f);
  return ok;
}

inline bool ocv_write(const std::string& path, int rows, int cols, int depth, const void* data) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  const int32_t hdr[4] = {rows, cols, depth, 1};
  const size_t n = (size_t)rows * cols * ocv_elem_size(depth);
  const bool ok = std::fwrite(hdr, sizeof(int32_t), 4, f) == 4 && std::fwrite(data, 1, n, f) == n;
  std::fclose(f);
  return ok;
}
