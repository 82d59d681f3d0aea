// fdoct_ctx.h -- what the three translation units of the C-ABI layer share: the handle (fdoct_ctx), the error / device-scope /
// device-memory helpers, and the declarations of
//   fdoct_state.cpp   plan selection and everything a handle uploads to its device (tables, planes, twiddles)
//   fdoct_route.cpp   the dispatch: choose_route, the passes in front of the chain, one launcher per kernel family, enqueue
//   fdoct_capi.cpp    the extern "C" entry points of include/fdoct.h
// (round 5: one 2900-line file until then; the seams are DESIGN.md 3.5's).  Internal: nothing outside fdoct_amd/csrc includes it.
#pragma once
#include <hip/hip_runtime.h>


#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/fdoct.h"
#include "fdoct_big.h"
#include "fdoct_host.h"
#include "fdoct_kernels.h"
#include "fdoct_wave.h"
#include "fdoct_jit.h"
#include "fdoct_hostcopy.h"

using namespace fdoct;

namespace fdoct_impl {

inline const double kPi = 3.141592653589793;  // BscanFFT.cpp:609
inline thread_local std::string g_create_error;  // fdoct_create has no handle to report through

struct RefFrame {  // a caller-supplied reference frame (background / pi / dark), as doubles
  std::vector<double> v;
  int rows = 0;  // 0 = unset, 1 = one spectrum for all rows, H = full frame
};

}  // namespace fdoct_impl
using fdoct_impl::RefFrame;

struct fdoct_ctx {
  fdoct_config cfg{};
  int W = 0, H = 0, N = 0, D = 0, M = 1, A = 1;
  int device = 0, num_cu = 256;
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  std::string err;

  // host state
  std::vector<double> win;
  std::vector<int32_t> idx;
  std::vector<double> frac;
  RefFrame yb, yp, yd;
  std::vector<float> phase;  // N (cos,sin) pairs or empty
  bool custom_win = false, custom_table = false, force_general = false, staged = false, bandpass = false;
  bool dirty = true;

  // derived plan
  bool cplx = false;
  int NC = 0;
  FusedPlan plan{};
  int split = 0, scratch_bytes = 0, tw_count = 0;
  int block_override = 0, grid_override = 0, plan_override = -1;  // plan_override == -2: force the generic path
  bool use_generic = false;   // no specialised kernel for this configuration: fdoct_generic.hip runs it
  bool generic_radix16 = false;  // ... its pass plans hold radix-16 butterflies (the 1024-thread kernels)
  bool generic_inplace = false;  // ... with ONE DFT buffer in LDS (rows whose two ping-pong buffers do not fit: generic_kernel<1024, 1, true>)
  bool generic_tables_ok = false;
  std::vector<int> rad_n, rad_nh, rad_wh, rad_mwh, rad_blu;
  // the zero-pad stage at full length inside generic_kernel (round 6: odd widths, half lengths with a prime factor above 5)
  struct GenericDftPlan {
    int n = 0, blu_m = 0;
    std::vector<int> rad;   // of n, or of blu_m
    float2 *d_tw = nullptr, *d_chirp = nullptr, *d_bhat = nullptr;
  };
  unsigned* d_gen_tickets = nullptr;   // kGenTickets row counters of generic_kernel launches, used round-robin (one per launch in flight)
  unsigned gen_ticket_seq = 0;
  bool zp_full = false;
  int zn = 0;               // W + 2 floor((M W - W) / 2)
  GenericDftPlan gzf, gzi;  // the W-point and the zn-point +i transform
  int blu_m = 0;  // > 0: the final transform (length N or N/2) has a prime factor > 5 and runs as Bluestein's chirp-z of this power-of-two length
  float2 *d_blu_chirp = nullptr, *d_blu_bhat = nullptr, *d_twg_blu = nullptr;

  // device state
  float *d_ib = nullptr, *d_ib2d = nullptr, *d_ib2d_f = nullptr, *d_yp = nullptr, *d_yd = nullptr, *d_yp_lo = nullptr, *d_yd_lo = nullptr, *d_win = nullptr, *d_g = nullptr;
  float *d_il = nullptr, *d_il2d = nullptr, *d_il2d_f = nullptr, *d_il_p = nullptr;  // d_il_p: d_il in the order of the fused kernels' LDS planes  // low words of the reciprocal background, laid out like d_ib / d_ib2d / d_ib2d_f
  uint32_t *d_il16 = nullptr, *d_il16_2d = nullptr;  // the second word as the fast path reads it: il / ib * 2^38 as half-float pairs (fdoct_kernels.h: FDOCT_PREC16)
  // fdoct_set_precise_division.  ON by default (round 5): main:1132 divides in double, and one f32 reciprocal leaves a fixed
  // pattern of 6e-8 of the DC level -- 8 x the tolerance on fringes of 1e-3 of it.  Off (or FDOCT_PRECISE_DIVISION=0) is the
  // opt-out for callers who know their fringes exceed ~1 % of the DC level.
  bool precise_div = true;
  // BscanFFTsim.cpp with averages > 1 (sim:936-947): every frame's magnitudes are COPIED over the last one's (the accumulate
  // is commented out) and what is emitted, undivided, is the last copy -- frame averages - 1 of every group.  The chain then
  // runs with A = 1 on those frames only (sim_last_frames gathers them); sim_group is the group length the caller counts in.
  int sim_group = 1;
  void* ws_sim = nullptr;
  size_t ws_sim_cap = 0;
  uint32_t* d_gidx = nullptr;
  float2 *d_tw = nullptr, *d_utw = nullptr, *d_phase = nullptr, *d_minmax = nullptr;
  // generic path
  float *d_win_g = nullptr, *d_win_lo_g = nullptr, *d_g_g = nullptr;
  int32_t* d_idx_g = nullptr;
  // wave-per-row kernels (fdoct_wave.hip)
  uint32_t* d_wave_gidx = nullptr;
  float2* d_wave_tw = nullptr;
  int wave_tw_count = 0, wave_off[6] = {0, 0, 0, 0, 0, 0};
  bool wave_tables_ok = false;
  float2 *d_twg_n = nullptr, *d_twg_nh = nullptr, *d_twg_w = nullptr, *d_twg_mw = nullptr, *d_twg_wh = nullptr, *d_twg_mwh = nullptr;
  size_t minmax_cap = 0;
  // long-row path (fdoct_big.hip): rows in HBM, one DFT plan per length
  struct BigGroupPlan {        // one launch: a group of the transform's passes with the data in LDS (fdoct_big.h)
    int P = 1, Q = 1, F = 1, log2ts = 0;
    std::vector<int> rad;
  };
  struct BigPlan {
    std::vector<int> rad;      // Stockham radices of the length itself, or (Bluestein) of mb: the one-launch-per-pass form
    std::vector<BigGroupPlan> groups;  // the same transform as a few launches of several passes each (empty: not available)
    int mb = 0;                // > 0: the length has a prime factor above 5 and runs as Bluestein around two mb-point DFTs
    float2 *d_tw = nullptr, *d_chirp = nullptr, *d_bhat = nullptr;  // exp(+2 pi i j / (mb ? mb : n)); e^(+i pi m^2/n); DFT(conj chirp)/mb
  };
  bool use_big = false;
  std::map<int, BigPlan> big_plans;
  float* ws_big_y = nullptr;
  float2 *ws_big_a = nullptr, *ws_big_b = nullptr;
  size_t ws_big_y_cap = 0, ws_big_a_cap = 0, ws_big_b_cap = 0;
  // workspaces
  void* ws_in = nullptr;
  size_t ws_in_cap = 0;
  float *ws_f32 = nullptr, *ws_f32_lo = nullptr;   // f64 frames as two f32 planes (launch_f64_split)
  size_t ws_f32_cap = 0, ws_f32_lo_cap = 0;
  float* ws_mov_lo = nullptr;                       // ... and the moving average of the low plane
  size_t ws_mov_lo_cap = 0;
  float *ws_out0 = nullptr, *ws_out1 = nullptr, *ws_tr = nullptr;
  size_t ws_out0_cap = 0, ws_out1_cap = 0, ws_tr_cap = 0;
  float2* ws_ylin = nullptr;
  size_t ws_ylin_cap = 0;
  long long ylin_rows = 0;  // A-scans the last staged run left in ws_ylin (0: none)
  float* ws_mov = nullptr;
  size_t ws_mov_cap = 0;
  void *ws_front = nullptr, *ws_med = nullptr, *ws_raw = nullptr;
  size_t ws_front_cap = 0, ws_med_cap = 0, ws_raw_cap = 0;
  int fe_median = 0, fe_binx = 1, fe_biny = 1;
  // display post-chain
  // host-pointer pipeline (fdoct_process with host buffers): copy-in / kernels / copy-out on three streams
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t pe_in[2] = {nullptr, nullptr}, pe_k[2] = {nullptr, nullptr}, pe_out[2] = {nullptr, nullptr};
  void* pl_in[2] = {nullptr, nullptr};
  float *pl_mag[2] = {nullptr, nullptr}, *pl_db[2] = {nullptr, nullptr};
  size_t pl_in_cap[2] = {0, 0}, pl_mag_cap[2] = {0, 0}, pl_db_cap[2] = {0, 0};
  // the same pipeline fed from / drained to PAGEABLE caller memory (fdoct_hostcopy.h): pinned staging slots the handle owns
  // and the threads that move a chunk between them and the caller's buffers
  void* pin_in[2] = {nullptr, nullptr};
  float *pin_mag[2] = {nullptr, nullptr}, *pin_db[2] = {nullptr, nullptr};
  size_t pin_in_cap[2] = {0, 0}, pin_mag_cap[2] = {0, 0}, pin_db_cap[2] = {0, 0};
  fdoct_impl::HostCopyPool* copy_pool = nullptr;
  int host_staging = -1;  // fdoct_set_host_staging: -1 = the library decides per buffer (pageable: staged), 0 = never, > 0 = that many copy threads
  unsigned char lut[768];
  bool lut_dirty = true;
  unsigned char* d_lut = nullptr;
  double* d_disp_part = nullptr;
  size_t disp_part_cap = 0;
  void *ws_disp_in = nullptr, *ws_disp_in2 = nullptr, *ws_disp_out = nullptr;
  size_t ws_disp_in_cap = 0, ws_disp_in2_cap = 0, ws_disp_out_cap = 0;

  fdoct_timing timing{};
  bool timing_pending = false, timing_staged = false;
  bool async_timing = false, record_now = false;  // event records cost stream time: async calls opt in
  bool rec_first = true, rec_last = true;         // chunked calls: the first chunk records the start events, the last one the end events
  unsigned* d_tro_fault = nullptr;                // see FusedArgs::tr_fault: one word of pinned, device-visible HOST memory, so that any
                                                  // entry point can look at it without a copy or a synchronisation of its own
  bool tro_used = false;                          // a TRO launch has run on this handle
  bool tro_enabled = true;                        // FDOCT_NO_TRO=1 (tuning / tests): always the two-pass path
  size_t tr_chunk_bytes = (size_t)2 << 30;        // transposed layout, two-pass path: row-major intermediate per chunk (bounds the workspace)
  bool jit = true;                                // fdoct_set_jit / FDOCT_JIT=0: compile the wave-per-row kernel for shapes off the built-in list
  std::string jit_note;                           // why the last run-time compile was refused (the call itself fell back and succeeded)
  int last_kernel = FDOCT_KERNEL_NONE;            // fdoct_last_kernel
};

namespace fdoct_impl {

inline int fail(fdoct_ctx* h, int code, const std::string& msg) {
  if (h)
    h->err = msg;
  else
    g_create_error = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                       \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(h, FDOCT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

// Every entry point works on the handle's device and leaves the calling thread's current device as it found it: a host
// that drives other GPUs through HIP (or torch) on the same thread is not re-pointed behind its back.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) {
      err = hipSetDevice(device);
      switched = (err == hipSuccess);
    }
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};
#define DEVICE_SCOPE(h)                                                                                     \
  DeviceScope device_scope_((h)->device);                                                                   \
  if (device_scope_.err != hipSuccess)                                                                      \
  return fail(h, FDOCT_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(device_scope_.err))

template <typename T>
int dev_alloc(fdoct_ctx* h, T** p, size_t count) {
  if (*p) {
    (void)hipFree(*p);
    *p = nullptr;
  }
  if (count == 0) return FDOCT_OK;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
  if (e != hipSuccess) return fail(h, FDOCT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  return FDOCT_OK;
}

template <typename T>
int dev_reserve(fdoct_ctx* h, T** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes && *p) return FDOCT_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), bytes);
  if (e != hipSuccess) return fail(h, FDOCT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  *cap = bytes;
  return FDOCT_OK;
}

// pinned host memory, grown on demand (the staging slots of fdoct_process's chunk pipeline)
template <typename T>
int host_reserve(fdoct_ctx* h, T** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes && *p) return FDOCT_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(p), bytes, hipHostMallocDefault);
  if (e != hipSuccess) return fail(h, FDOCT_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  *cap = bytes;
  return FDOCT_OK;
}

template <typename T>
int upload(fdoct_ctx* h, T** dptr, const std::vector<T>& v) {
  int rc = dev_alloc(h, dptr, v.size());
  if (rc) return rc;
  if (!v.empty()) HIP_TRY(h, hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return FDOCT_OK;
}

// COLORMAP_JET as OpenCV builds it (fdoct_host.cpp::build_opencv_jet): dark blue (128,0,0 in B,G,R) at 0 through cyan and
// yellow to dark red (0,0,128) at 255.
inline void builtin_jet(unsigned char* bgr) { build_opencv_jet(bgr); }

// ---- fdoct_state.cpp ------------------------------------------------------------------------------------------------------
size_t dtype_size(int dt);
int copy_ref_frame(fdoct_ctx* h, RefFrame& dst, const void* data, fdoct_dtype dtype, int rows, size_t pitch);
bool factor_radices(int n, std::vector<int>& rad, int log2max = 0);
bool generic_real_half(const fdoct_ctx* h);
int generic_buffer_len(const fdoct_ctx* h);
size_t generic_lds_bytes(const fdoct_ctx* h, int buffers = 0);
int select_generic(fdoct_ctx* h);
int select_plan(fdoct_ctx* h);
size_t const_lds_bytes(const fdoct_ctx* h, bool planes, bool il_plane, bool il_half, bool tw3 = true, bool gi = true);
size_t tro_const_lds_bytes(const fdoct_ctx* h, int sample_bytes, bool normalize);  // (of the launch: fused_tro_pf2 depends on both)
void reciprocal_words(const std::vector<double>& yb, std::vector<float>& ib, std::vector<float>& il);
struct PlaneScales { double yb, yp, yd; };
PlaneScales plane_scales(const fdoct_ctx* h);
std::vector<double> scaled_copy(const std::vector<double>& v, double s);
int rebuild_device_state(fdoct_ctx* h);
void build_bluestein_tables(int n, int Mb, std::vector<float2>& chirp, std::vector<float2>& bhat);
int rebuild_generic_state(fdoct_ctx* h);
int rebuild_wave_state(fdoct_ctx* h);

// ---- fdoct_route.cpp ------------------------------------------------------------------------------------------------------
// ---- dispatch ----------------------------------------------------------------------------------------------------------
// Everything a call decides before it launches anything: which passes run in front of the chain, which kernel family takes it
// and with what.  A function of the handle's state and of the call's geometry only (pointers enter through their alignment), so
// that fdoct_prepare makes the same decisions -- and pays for a run-time compile -- without frames.
struct Route {
  int family = FDOCT_KERNEL_NONE;   // fdoct_kernel: who runs the chain
  bool frontend = false;            // medianBlur + binning pass over the raw frames first (main:953-958)
  bool narrow_f64 = false;          // data_y doubles narrowed once to float (main:987)
  bool mov_lo = false;              // smoothmovavg of FLOAT frames: the tap sums in double, handed on as two f32 planes like the doubles'
  bool movavg = false;              // smoothmovavg pass (main:990-991)
  int kdt = -1;                     // sample type the chain's kernel reads (FDOCT_K_*)
  size_t kpitch = 0;                // ... and its row pitch
  bool need_minmax = false;         // whole-frame min / max pre-pass (main:1128)
  bool tro = false;                 // the fused chain writes the D x H layout itself
  bool transpose_pass = false;      // ... or a transpose pass does
  hipFunction_t jit_fn = nullptr;   // FDOCT_KERNEL_WAVE_JIT: the kernel compiled for this handle
  bool bin2_in_kernel = false;      // ... with the 2 x 2 software binning inside its loads (the raw frames go to it as they are)
  int wave_opt = 0;                 // FDOCT_WAVE_OPT_* of that kernel
};

// Quantities of one call that every family's launch needs.
struct Call {
  const void* kframes = nullptr;    // what the chain's kernel reads (the caller's frames, or the last pre-pass's output)
  const float* kframes_lo = nullptr;  // f64 frames: the low words of kframes (same pitch), else null
  int nframes = 0, G = 0;
  long long in_rows = 0, out_rows = 0;
  size_t es = 0;                    // bytes per sample of the CALLER's frames (the algorithmic-bytes figure)
  float *k_mag = nullptr, *k_db = nullptr;          // where the chain's kernel writes (the caller's arrays, or the transpose pass's input)
  float *d_out_bscan = nullptr, *d_out_db = nullptr;
  hipStream_t st = nullptr;
};

int kernel_dtype(int dt);
int run_frontend(fdoct_ctx* h, const void* d_raw, int kdt, int nframes, int raw_w, int raw_h, size_t raw_pitch, int mediann,
                 int binx, int biny, void** out, size_t* out_pitch);
void big_plans_free(fdoct_ctx* h);
int choose_route(fdoct_ctx* h, fdoct_dtype dtype, uintptr_t frames_addr, size_t pitch_bytes, uintptr_t out_bscan_addr,
                 uintptr_t out_db_addr, fdoct_layout layout, int nframes, Route* r);
int enqueue(fdoct_ctx* h, const void* d_frames, fdoct_dtype dtype, int nframes, size_t pitch_bytes,
            float* d_out_bscan, float* d_out_db, fdoct_layout layout);

}  // namespace fdoct_impl
