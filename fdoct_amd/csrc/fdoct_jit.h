// fdoct_jit.h -- run-time compilation of the wave-per-row kernel (fdoct_wave_dev.h) for a shape outside the built-in list.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace fdoct {

// Can the wave-per-row kernel template be instantiated for this shape at all (what its static_asserts and the host's LDS
// layout ask for)?  A cheap host-side test; the compile itself is the final word.
bool wave_jit_shape_ok(int W, int M, int N, int D, int opt = 0);

// wave_kernel<W, M, N, sample type of `kdtype` (FDOCT_K_*), depth bins per lane, opt (FDOCT_WAVE_OPT_*), depth bound> for numdisplaypoints D
// (its class: (D + 63) / 64 and wave_depth_bound) on `device`: from the process-wide cache, from the disk
// cache ($FDOCT_JIT_CACHE, else $XDG_CACHE_HOME/fdoct_amd, else $HOME/.cache/fdoct_amd), or compiled now by hipRTC (seconds).
// Returns hipSuccess and the function, or an error with its reason in *why (the caller falls back to generic_kernel).
// A failure is remembered: the same shape is not compiled again in this process.  Thread-safe.
hipError_t wave_jit_get(int W, int M, int N, int kdtype, int D, int opt, int device, hipFunction_t* fn, std::string* why);

// Compile only (no device needed: the build check and the CPU tests): bytes of the code object for `gcn_arch` ("gfx950"), or -1
// with the reason in *why.
long long wave_jit_compile_only(int W, int M, int N, int kdtype, int D, int opt, const char* gcn_arch, std::string* why);

// launch with the same contract as launch_wave (fdoct_wave.h)
struct WaveArgs;
hipError_t wave_jit_launch(hipFunction_t fn, const WaveArgs& a, int grid, int waves, size_t lds, hipStream_t st);

}  // namespace fdoct
