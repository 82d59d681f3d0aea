// fdoct_host.h -- host-side (double precision) set-up arithmetic of the path.
#pragma once
#include <stdint.h>

#include <vector>

namespace fdoct {

// A0, BscanFFT.cpp:615-698.  idx/frac get N entries.
void build_resample_table(int W, int M, int N, double lambdamin, double lambdamax, std::vector<int32_t>& idx,
                          std::vector<double>& frac);
// A1, BscanFFT.cpp:936-944.
void build_barthann(int W, std::vector<double>& win);
// applyColorMap(.., COLORMAP_JET), BscanFFT.cpp:1284: OpenCV's 256-entry B,G,R table, built the way OpenCV builds it.
void build_opencv_jet(unsigned char* bgr256);

struct GatherLayout {
  int split;      // 1: even samples at [0,WC/2), odd at [WC/2,WC)
  int zero_slot;  // float index that always holds 0
};
// LDS byte offset (relative to the row's staging buffer) of sample i.
inline int staging_offset_bytes(int i, int WC, int split) {
  return 4 * (split ? ((i >> 1) + (i & 1) * (WC / 2)) : i);
}

}  // namespace fdoct
