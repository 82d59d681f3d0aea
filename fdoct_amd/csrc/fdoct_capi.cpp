// fdoct_capi.cpp -- the extern "C" boundary (include/fdoct.h) over the HIP kernels.
//
// Replaces the processing block of the reference's main() loop
// (BscanFFT.cpp:1123-1240 / BscanFFTsim.cpp:842-955) plus its one-time set-up
// (BscanFFT.cpp:544-698, 936-944).  No CPU compute path exists here: every
// fdoct_process* call runs the gfx950 kernels or fails.
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

using namespace fdoct;

namespace {

const double kPi = 3.141592653589793;  // BscanFFT.cpp:609
thread_local std::string g_create_error;

struct RefFrame {  // a caller-supplied reference frame (background / pi / dark), as doubles
  std::vector<double> v;
  int rows = 0;  // 0 = unset, 1 = one spectrum for all rows, H = full frame
};

}  // namespace

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
  int blu_m = 0;  // > 0: the final transform (length N or N/2) has a prime factor > 5 and runs as Bluestein's chirp-z of this power-of-two length
  float2 *d_blu_chirp = nullptr, *d_blu_bhat = nullptr, *d_twg_blu = nullptr;

  // device state
  float *d_ib = nullptr, *d_ib2d = nullptr, *d_ib2d_f = nullptr, *d_yp = nullptr, *d_yd = nullptr, *d_win = nullptr, *d_g = nullptr;
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
  float *d_win_g = nullptr, *d_g_g = nullptr;
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

namespace {

int fail(fdoct_ctx* h, int code, const std::string& msg) {
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

template <typename T>
int upload(fdoct_ctx* h, T** dptr, const std::vector<T>& v) {
  int rc = dev_alloc(h, dptr, v.size());
  if (rc) return rc;
  if (!v.empty()) HIP_TRY(h, hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return FDOCT_OK;
}

// The analytic jet ramp: channel(x) = clamp(1.5 - |4x - c|, 0, 1) with c = 1 (blue), 2 (green), 3 (red), x = i/255;
// dark blue (128,0,0 in B,G,R) at 0 through cyan and yellow to dark red (0,0,128) at 255.
void builtin_jet(unsigned char* bgr) {
  for (int i = 0; i < 256; i++) {
    const double x = i / 255.0;
    for (int ch = 0; ch < 3; ch++) {
      double v = 1.5 - std::fabs(4.0 * x - (ch + 1));
      v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
      bgr[3 * i + ch] = (unsigned char)std::lrint(v * 255.0);
    }
  }
}

size_t dtype_size(int dt) {
  switch (dt) {
    case FDOCT_U8: return 1;
    case FDOCT_U16: return 2;
    case FDOCT_F32: return 4;
    case FDOCT_F64: return 8;
    default: return 0;
  }
}

int copy_ref_frame(fdoct_ctx* h, RefFrame& dst, const void* data, fdoct_dtype dtype, int rows, size_t pitch) {
  if (!data) {
    dst.v.clear();
    dst.rows = 0;
    h->dirty = true;
    return FDOCT_OK;
  }
  const size_t es = dtype_size(dtype);
  if (!es) return fail(h, FDOCT_ERR_INVALID, "bad dtype");
  if (rows != 1 && rows != h->H) return fail(h, FDOCT_ERR_INVALID, "reference frame rows must be 1 or height");
  if (pitch == 0) pitch = es * h->W;
  if (pitch < es * h->W) return fail(h, FDOCT_ERR_INVALID, "pitch smaller than a row");
  dst.v.resize((size_t)rows * h->W);
  for (int r = 0; r < rows; r++) {
    const unsigned char* row = static_cast<const unsigned char*>(data) + (size_t)r * pitch;
    double* o = dst.v.data() + (size_t)r * h->W;
    for (int i = 0; i < h->W; i++) {
      switch (dtype) {
        case FDOCT_U8: o[i] = reinterpret_cast<const uint8_t*>(row)[i]; break;
        case FDOCT_U16: o[i] = reinterpret_cast<const uint16_t*>(row)[i]; break;
        case FDOCT_F32: o[i] = reinterpret_cast<const float*>(row)[i]; break;
        default: o[i] = reinterpret_cast<const double*>(row)[i]; break;
      }
    }
  }
  dst.rows = rows;
  h->dirty = true;
  return FDOCT_OK;
}

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// n = 2^a 3^b 5^c -> Stockham radices (4s first), false if another prime divides n
// Radix plan of the generic kernel's Stockham DFT (radices 16/8/4/2/5/3).  The first pass writes butterfly j's
// outputs R apart (stride R*8 bytes across lanes), so it gets an odd radix -- or a small power of two -- to keep
// those LDS writes off the same banks; it is also the pass without twiddle multiplies.
bool factor_radices(int n, std::vector<int>& rad, int log2max = 0) {
  rad.clear();
  int a = 0, b = 0, c = 0;
  while (n % 2 == 0) { a++; n /= 2; }
  while (n % 3 == 0) { b++; n /= 3; }
  while (n % 5 == 0) { c++; n /= 5; }
  if (n != 1) return false;
  for (int i = 0; i < c; i++) rad.push_back(5);
  for (int i = 0; i < b; i++) rad.push_back(3);
  if (rad.empty() && a > 0) {
    const int first = (a % 2) ? 1 : 2;
    rad.push_back(1 << first);
    a -= first;
  }
  const int kLog2Max = log2max ? log2max : (GENERIC_MAX_RADIX >= 16 ? 4 : 3);
  for (; a >= kLog2Max; a -= kLog2Max) rad.push_back(1 << kLog2Max);
  if (a) rad.push_back(1 << a);
  return (int)rad.size() <= GENERIC_MAX_PASSES;
}

// real rows run the N-point DFT as an N/2-point complex one (see generic_kernel)
bool generic_real_half(const fdoct_ctx* h) { return h->phase.empty() && (h->N % 2) == 0; }

int generic_buffer_len(const fdoct_ctx* h) {
  const int MW = h->W * h->M;
  int L = generic_real_half(h) ? h->N / 2 : h->N;
  if (h->M > 1) L = std::max(L, MW / 2);  // the zero-pad DFTs run at half length (real row, Hermitian spectrum)
  if (h->blu_m > L) L = h->blu_m;         // Bluestein: the transform runs as two power-of-two DFTs of this length
  return L;
}

size_t generic_lds_bytes(const fdoct_ctx* h, int buffers = 0) {
  const int L = generic_buffer_len(h);
  const int ybuf = (h->W + 3) & ~3;
  if (!buffers) buffers = h->generic_inplace ? 1 : 2;
  return (size_t)ybuf * 4 + (size_t)L * 8 * buffers + (size_t)((h->D + 3) & ~3) * 4;  // row, the DFT buffer(s), magnitude sums
}

// The any-configuration path: checks that fdoct_generic.hip can run this geometry.
int select_generic(fdoct_ctx* h) {
  const int MW = h->W * h->M;
  // cv::dft takes any length (main:1185).  Lengths with prime factors up to 5 run as mixed-radix Stockham passes; any
  // other length as Bluestein's algorithm: two power-of-two DFTs of length >= 2n - 1 around a chirp multiplication.
  h->blu_m = 0;
  const int tlen = generic_real_half(h) ? h->N / 2 : h->N;  // the transform the kernel actually runs
  std::vector<int> probe;
  if (!factor_radices(tlen, probe)) {
    int mb = 1;
    while (mb < 2 * tlen - 1) mb <<= 1;
    h->blu_m = mb;
    factor_radices(mb, h->rad_blu);
    h->rad_n.clear();
    h->rad_nh.clear();
  } else {
    if (!factor_radices(h->N, h->rad_n)) h->rad_n.clear();  // (only used when the full-length transform runs)
    if ((h->N % 2) == 0 && !factor_radices(h->N / 2, h->rad_nh)) h->rad_nh.clear();
  }
  h->use_big = false;
  if (h->M > 1) {
    // an odd width (the reference's fftshift leaves the last column of the spectrum where it is and, under an even multiplier,
    // pads to M W - 1 bins, main:215-241) and zero-pad lengths with a prime factor above 5: the long-row path, whose DFTs run at
    // full length and take any length (the LDS kernels halve the transforms of a real row, which needs an even width)
    if ((h->W % 2) || !factor_radices(h->W / 2, h->rad_wh) || !factor_radices(MW / 2, h->rad_mwh)) {
      h->rad_wh.clear();
      h->rad_mwh.clear();
      h->use_big = true;
    }
  }
  // rows whose two DFT buffers do not fit the 160 KB of LDS (half-length transforms beyond about 9000 points): with ONE buffer and
  // every step in place (generic_kernel<1024, 1, true>) up to 16384 points -- 4096 samples upsampled x8 -- as long as a thread of
  // the 1024 holds its share of a pass in 16 registers (radices 5 / 3: 15), the zero-pad spectrum in 8 and the resampled row
  // in 32, and the length needs no Bluestein; what lies beyond runs with the rows in HBM (fdoct_big.hip)
  h->generic_inplace = false;
  // (FDOCT_GENERIC_INPLACE_ABOVE: the two-buffer footprint above which the one-buffer kernel is taken, for measurements)
  static const size_t inplace_above = [] { const char* e = std::getenv("FDOCT_GENERIC_INPLACE_ABOVE"); return e ? (size_t)std::atol(e) : (size_t)160 * 1024; }();
  const bool must_inplace = generic_lds_bytes(h, 2) + 1024 > 160 * 1024;
  if (generic_lds_bytes(h, 2) + 1024 > inplace_above) {
    auto pass_ok = [](const std::vector<int>& rad, int n) {
      for (int R : rad)
        if (R > 16 || n / R > 1024 * (16 / R)) return false;
      return !rad.empty();
    };
    const bool real_half = generic_real_half(h);
    // (the in-place passes take radix-16 butterflies -- one per thread on a 16384-point transform -- and with them a pass less)
    std::vector<int> r_n = h->rad_n, r_nh = h->rad_nh, r_wh = h->rad_wh, r_mwh = h->rad_mwh;
    if (!h->blu_m && !h->use_big) {
      if (!h->rad_n.empty()) factor_radices(h->N, h->rad_n, 4);
      if (!h->rad_nh.empty()) factor_radices(h->N / 2, h->rad_nh, 4);
      if (h->M > 1) {
        factor_radices(h->W / 2, h->rad_wh, 4);
        factor_radices(MW / 2, h->rad_mwh, 4);
      }
    }
    const bool ok = !h->use_big && !h->blu_m && generic_lds_bytes(h, 1) + 1024 <= 160 * 1024 && h->N <= 32 * 1024 &&
                    (real_half ? pass_ok(h->rad_nh, h->N / 2) : pass_ok(h->rad_n, h->N)) &&
                    (h->M == 1 || (h->W / 2 <= 8 * 1024 && pass_ok(h->rad_wh, h->W / 2) && pass_ok(h->rad_mwh, MW / 2)));
    if (ok) {
      h->generic_inplace = true;
    } else {
      if (must_inplace) h->use_big = true;
      h->rad_n = r_n; h->rad_nh = r_nh; h->rad_wh = r_wh; h->rad_mwh = r_mwh;
    }
  }
  // rows of which a CU holds one (two buffers beyond half the LDS) run with 1024 threads, 128 registers each: radix-16 passes there too
  h->generic_radix16 = h->generic_inplace;
  {
    static const int r16 = [] { const char* e = std::getenv("FDOCT_GENERIC_RADIX16"); return e ? std::atoi(e) : 1; }();  // measurement
    if (r16 && !h->generic_inplace && !h->use_big && !h->blu_m && generic_lds_bytes(h, 2) > (160 * 1024 - 1024) / 2) {
      if (!h->rad_n.empty()) factor_radices(h->N, h->rad_n, 4);
      if (!h->rad_nh.empty()) factor_radices(h->N / 2, h->rad_nh, 4);
      if (h->M > 1) {
        factor_radices(h->W / 2, h->rad_wh, 4);
        factor_radices(MW / 2, h->rad_mwh, 4);
      }
      h->generic_radix16 = true;
    }
  }
  {
    static const int force = [] { const char* e = std::getenv("FDOCT_FORCE_LONG_ROWS"); return e ? std::atoi(e) : 0; }();  // measurement
    if (force) h->use_big = true, h->generic_inplace = false;
  }
  if (h->use_big && (h->N > (1 << 24) || MW > (1 << 24)))
    return fail(h, FDOCT_ERR_UNSUPPORTED, "rows of more than 2^24 points");
  h->use_generic = true;
  return FDOCT_OK;
}

// Pick the compiled plan for the current (N, W, phase) and derive LDS geometry; configurations without a
// specialised kernel go to the generic path.
int select_plan(fdoct_ctx* h) {
  h->cplx = !h->phase.empty();
  h->use_generic = false;
  h->NC = h->cplx ? h->N : h->N / 2;
  const bool special_ok = is_pow2(h->N) && h->M == 1 && (h->W % 8) == 0 && (h->cplx || h->D <= h->N / 2) &&
                          h->plan_override != -2;
  bool found = false;
  // preference order for equal NC: the override, then the measured-fastest plan ids
  static const int pref[] = {5, 2, 3, 0, 1, 7, 6, 8, 4};  // per NC: fastest first; equal plans: smallest chunk count that holds W
  FusedPlan q{};
  if (special_ok && h->plan_override >= 0 && fused_plan_get(h->plan_override, &q) && q.nc == h->NC && h->W <= 8 * q.T * q.WCH) {
    h->plan = q;
    found = true;
  }
  for (int i = 0; special_ok && !found && i < (int)(sizeof pref / sizeof pref[0]); i++) {
    if (fused_plan_get(pref[i], &q) && q.nc == h->NC && h->W <= 8 * q.T * q.WCH) {
      h->plan = q;
      found = true;
    }
  }
  if (!found) return select_generic(h);
  const FusedPlan& p = h->plan;
  const int WC = 8 * p.T * p.WCH;
  const int LP = p.R1 == 32 ? 5 : p.R1 == 16 ? 4 : p.R1 == 8 ? 3 : 2;
  const int stg = 4 * (WC + 4);
  const int xch = p.kind == 1 ? 8 * (65 * 16 + 2) : p.kind == 2 ? 8 * (129 * 16 + 2) : 8 * (h->NC + (h->NC >> LP) + 2);
  h->scratch_bytes = ((stg > xch ? stg : xch) + 15) & ~15;
  const double sigma = (h->cplx ? 1.0 : 2.0) * (double)(h->W * h->M) / (double)h->N;
  h->split = (sigma >= 1.5 && sigma <= 3.0) ? 1 : 0;
  int tw = (p.R2 - 1) * p.R1 + (p.R3 > 1 ? (p.R3 - 1) * p.R1 * p.R2 : 0);
  if (p.kind == 1) tw = 48 + 15 * 64;
  if (p.kind == 2) tw = 96 + 128;  // step-5 twiddles are formed as powers of W_2048^(l') in the kernel
  h->tw_count = (tw + 1) & ~1;
  return FDOCT_OK;
}

// planes: the three constant planes are staged in LDS (kernels that do not keep them in registers); il_plane: so is the low
// word of the reciprocal background (FusedArgs::prec == 1)
// il_half: that plane holds half floats (the fast-path kernels with at most 32 samples per lane: fused_il_half)
size_t const_lds_bytes(const fdoct_ctx* h, bool planes, bool il_plane, bool il_half) {
  const int WC = 8 * h->plan.T * h->plan.WCH;
  return (planes ? (size_t)3 : 0) * WC * 4 + (il_plane ? (size_t)WC * (il_half ? 2 : 4) : 0) + (size_t)h->tw_count * 8 + (h->cplx ? (size_t)h->NC * 8 : 0) + (size_t)h->NC * 4;
}

// main:1132 divides by data_yb in double.  The kernels multiply by the reciprocal, held as an unevaluated sum of two floats
// ib + il = 1/yb to 2^-48: ib = fl32(1/yb) alone is off by up to 6e-8 of the quotient -- a fixed per-column pattern of the
// size of the DC level, which the chain turns into up to 4e-6 of the DC level per depth bin: more than the whole tolerance
// once the fringes are weaker than about 1 % of it.  With d = fma(v, ib, -c0) (rounded at the size of the deviation from the
// mean estimate c0) followed by d = fma(v, il, d), nothing is rounded at the size of the DC level.  x/0 -> 0 (OpenCV 3.x
// Mat division).
// float -> IEEE half bits, round to nearest even (values here are at most 2^14 in magnitude: no overflow handling needed beyond inf)
uint16_t half_bits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t e = (int32_t)((x >> 23) & 0xffu) - 127 + 15;
  uint32_t m = x & 0x7fffffu;
  if (((x >> 23) & 0xffu) == 0xffu) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0u));
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {  // subnormal half (or zero)
    if (e < -10) return (uint16_t)sign;
    m |= 0x800000u;
    const int shift = 14 - e;  // 13 + (1 - e)
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (r & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((uint32_t)e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;  // (a carry into the exponent is the right result)
  return (uint16_t)(sign | r);
}

// The second word as the fast-path kernels with at most 32 samples per lane apply it (fdoct_kernels.h: FDOCT_PREC16): what
// v * ib leaves out of v / yb is (v * ib) * rho, rho = (1/yb - ib) / ib, |rho| <= 2^-24; the kernel adds c0 * rho (c0: its
// estimate of the row mean of v / yb).  rho * 2^38 as half floats, in the order the lanes read them: the lane's 8-sample group
// of chunk c is 16 bytes at ((c T + lane) * 16), dword q = samples chunk_pair_offset(q), + 2 (the RawChunk pair order).
void half_pattern_row(const double* yb, int WC, int T, uint32_t* out) {
  auto rho_h = [&](int i) -> uint16_t {
    if (yb[i] == 0.0) return 0;
    const double q = 1.0 / yb[i];
    const float ib = (float)q;
    if (!std::isfinite(ib) || ib == 0.f) return 0;
    return half_bits((float)(std::ldexp((q - (double)ib) / (double)ib, kPrec16Shift)));
  };
  for (int i0 = 0; i0 < WC; i0 += 8) {
    const int grp = i0 / 8, ln = grp % T, c = grp / T;
    for (int q = 0; q < 4; q++) {
      const int off = (q & 1) * 4 + (q >> 1);
      out[(size_t)(c * T + ln) * 4 + q] = (uint32_t)rho_h(i0 + off) | ((uint32_t)rho_h(i0 + off + 2) << 16);
    }
  }
}

void reciprocal_words(const std::vector<double>& yb, std::vector<float>& ib, std::vector<float>& il) {
  ib.resize(yb.size());
  il.resize(yb.size());
  for (size_t i = 0; i < yb.size(); i++) {
    if (yb[i] != 0.0) {
      const double q = 1.0 / yb[i];
      ib[i] = (float)q;
      const double lo = q - (double)ib[i];
      il[i] = std::isfinite(lo) ? (float)lo : 0.f;  // (1/yb beyond the float range: ib is inf, as before)
    } else {
      ib[i] = il[i] = 0.f;
    }
  }
}

// smoothmovavg (main:247-304, 990-991) divides its 2n + 2 taps by 2 (n + 1) in double.  An f32 quotient would be a rounding at the
// size of the DC level unless n + 1 is a power of two (5 x the tolerance on fringes of 0.1 % of it with n = 2), so the pass in front
// of the chain hands on the tap SUMS -- exact in f32 for the camera's integer samples up to n = 126 -- and the factor K = 2 (n + 1)
// goes where the reference's arithmetic puts it: into the dark frame (subtracted from the samples themselves), and, unless a min-max
// normalisation follows (it is scale-invariant), into the pi frame and the background as well.
struct PlaneScales { double yb, yp, yd; };
PlaneScales plane_scales(const fdoct_ctx* h) {
  const double K = h->cfg.movavgn > 0 ? 2.0 * ((double)h->cfg.movavgn + 1.0) : 1.0;
  const bool norm_on = h->cfg.rowwisenormalize || (h->cfg.variant == FDOCT_VARIANT_SIM) || !h->cfg.donotnormalize;
  return {norm_on ? 1.0 : K, norm_on ? 1.0 : K, K};
}
std::vector<double> scaled_copy(const std::vector<double>& v, double s) {
  std::vector<double> t(v);
  if (s != 1.0)
    for (double& x : t) x *= s;
  return t;
}

// Recompute everything the kernel reads from the host-side state and upload it.
int rebuild_generic_state(fdoct_ctx* h);

int rebuild_device_state(fdoct_ctx* h) {
  int rc = select_plan(h);
  if (rc) return rc;
  h->generic_tables_ok = false;
  h->wave_tables_ok = false;
  if (h->use_generic) return rebuild_generic_state(h);
  const int W = h->W, H = h->H, N = h->N;
  const FusedPlan& p = h->plan;
  const int WC = 8 * p.T * p.WCH;
  DEVICE_SCOPE(h);

  // 1/background in double, as two floats (reciprocal_words)
  {
    std::vector<float> ib, il;
    const std::vector<double> ybs = h->yb.rows ? scaled_copy(h->yb.v, plane_scales(h).yb) : std::vector<double>();
    if (h->yb.rows) reciprocal_words(ybs, ib, il);
    {  // the half-float pattern of the second word (rows exactly one chunk width wide: the fast path's condition)
      std::vector<uint32_t> h16, h16_2d;
      if (W == WC && h->yb.rows == 1) {
        h16.resize((size_t)WC / 2);
        half_pattern_row(ybs.data(), WC, p.T, h16.data());
      } else if (W == WC && h->yb.rows > 1) {
        h16_2d.resize((size_t)H * WC / 2);
        for (int r = 0; r < H; r++) half_pattern_row(ybs.data() + (size_t)r * W, WC, p.T, h16_2d.data() + (size_t)r * WC / 2);
      }
      if ((rc = upload(h, &h->d_il16, h16))) return rc;
      if ((rc = upload(h, &h->d_il16_2d, h16_2d))) return rc;
    }
    if (h->yb.rows == 1) {
      if ((rc = upload(h, &h->d_ib, ib))) return rc;
      if ((rc = upload(h, &h->d_il, il))) return rc;
      {  // the same plane in the slot order of the kernels' LDS planes (sample 8 (ln + T c) + e -> c 8T + (e & 1) 4T + 4 ln + (e >> 1))
        std::vector<float> ilp((size_t)WC, 0.f);
        for (int i = 0; i < W; i++) {
          const int e = i & 7, ln = (i >> 3) & (p.T - 1), c = i / (8 * p.T);
          ilp[(size_t)c * 8 * p.T + (e & 1) * 4 * p.T + 4 * ln + (e >> 1)] = il[i];
        }
        if ((rc = upload(h, &h->d_il_p, ilp))) return rc;
      }
      if ((rc = dev_alloc(h, &h->d_ib2d_f, 0))) return rc;
      if ((rc = dev_alloc(h, &h->d_il2d_f, 0))) return rc;
    } else {
      // the fused kernels read a 2-D background with every 8-sample group stored evens first, then odds (the
      // order their sample pairs are held in), rows padded to the plan's chunk width; the generic kernel keeps
      // its own natural-order copy (d_ib2d)
      std::vector<float> perm((size_t)H * WC, 0.f);
      auto permute = [&](const std::vector<float>& src) {
        for (int r = 0; r < H; r++)
          for (int i = 0; i < W; i++) perm[(size_t)r * WC + (i & ~7) + ((i & 1) * 4 + ((i & 7) >> 1))] = src[(size_t)r * W + i];
      };
      permute(ib);
      if ((rc = upload(h, &h->d_ib2d_f, perm))) return rc;
      permute(il);
      if ((rc = upload(h, &h->d_il2d_f, perm))) return rc;
      if ((rc = dev_alloc(h, &h->d_ib, 0))) return rc;
      if ((rc = dev_alloc(h, &h->d_il, 0))) return rc;
    }
  }
  auto up_ref = [&](const RefFrame& f, double scale, float** d) -> int {
    std::vector<float> t(f.v.size());
    for (size_t i = 0; i < t.size(); i++) t[i] = (float)(f.v[i] * scale);
    return upload(h, d, t);
  };
  if ((rc = up_ref(h->yp, plane_scales(h).yp, &h->d_yp))) return rc;
  if ((rc = up_ref(h->yd, plane_scales(h).yd, &h->d_yd))) return rc;
  {
    // Window (main:1142) and slope step (main:1153-1173) folded into two per-sample planes: with t = x - mean and
    // y = t * w, s_i = y_i + g_i (y_i - y_(i-1)) = a_i t_i + b_i t_(i-1), a_i = (1 + g_i) w_i, b_i = -g_i w_(i-1).
    // Sample 0 has slopes[0] = slopes[1] (main:1161): s_0 = (1 - g_0) w_0 t_0 + g_0 w_1 t_1; the kernel feeds t_1 there.
    // g_i = fractionalk[i]: the reference indexes fractionalk (N entries) by nearestkindex[q], a SAMPLE index; past N
    // it is out of bounds there and defined as 0 here.  Real path: the 1/2 of the real-input untangle is folded into
    // the window (exact: power of two).  Products in double, rounded once.
    std::vector<float> pa(W), pb(W);
    const double half = h->cplx ? 1.0 : 0.5;
    auto gg = [&](int i) { return i < N ? h->frac[i] : 0.0; };
    for (int i = 1; i < W; i++) {
      pa[i] = (float)((1.0 + gg(i)) * half * h->win[i]);
      pb[i] = (float)(-gg(i) * half * h->win[i - 1]);
    }
    pa[0] = (float)((1.0 - gg(0)) * half * h->win[0]);
    pb[0] = (float)(gg(0) * half * h->win[1]);
    if ((rc = upload(h, &h->d_win, pa))) return rc;
    if ((rc = upload(h, &h->d_g, pb))) return rc;
  }
  {
    // gather sources: data_ylin[q] = s[nearestkindex[q]] for q = 1..N-2, else 0 (main:1164)
    std::vector<uint32_t> gi(h->NC);
    auto off = [&](int q) -> uint32_t {
      if (q <= 0 || q >= N - 1) return (uint32_t)(4 * WC);
      return (uint32_t)staging_offset_bytes(h->idx[q], WC, h->split);
    };
    for (int n = 0; n < h->NC; n++) gi[n] = h->cplx ? off(n) : (off(2 * n) | (off(2 * n + 1) << 16));
    if ((rc = upload(h, &h->d_gidx, gi))) return rc;
  }
  {
    std::vector<float2> tw(h->tw_count, make_float2(0.f, 0.f));
    size_t o = 0;
    if (p.kind == 1 || p.kind == 2) {
      // row-swap plans: tw2[(3c + i-1)*4 + j] = W_(4Q)^(i*(4c+j)), c < Q/4; tw3[(b-1)*L + l] = W_NC^(b*l), l < L = NC/16
      // (Q = first radix: 16 for fft1024_rowswap, 32 for fft2048_rowswap)
      const int Q = p.R1, L = h->NC / 16;
      for (int c = 0; c < Q / 4; c++)
        for (int i = 1; i < 4; i++)
          for (int j = 0; j < 4; j++) {
            const double a = 2.0 * kPi * (double)(i * (4 * c + j)) / (double)(4 * Q);
            tw[(3 * c + i - 1) * 4 + j] = make_float2((float)std::cos(a), (float)std::sin(a));
          }
      for (int b = 1; b < (p.kind == 1 ? 16 : 2); b++)  // kind 2 keeps only the b = 1 row
        for (int l = 0; l < L; l++) {
          const double a = 2.0 * kPi * (double)(b * l) / (double)h->NC;
          tw[3 * Q + (b - 1) * L + l] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    } else
    for (int r = 1; r < p.R2; r++)
      for (int k = 0; k < p.R1; k++) {
        const double a = 2.0 * kPi * (double)r * (double)k / (double)(p.R1 * p.R2);
        tw[o++] = make_float2((float)std::cos(a), (float)std::sin(a));
      }
    if (p.kind == 0 && p.R3 > 1)
      for (int r = 1; r < p.R3; r++)
        for (int k = 0; k < p.R1 * p.R2; k++) {
          const double a = 2.0 * kPi * (double)r * (double)k / (double)h->NC;
          tw[o++] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    if ((rc = upload(h, &h->d_tw, tw))) return rc;
    std::vector<float2> utw(p.T);
    for (int l = 0; l < p.T; l++) {
      const double a = 2.0 * kPi * (double)l / (double)N;
      utw[l] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    if ((rc = upload(h, &h->d_utw, utw))) return rc;
  }
  {
    std::vector<float2> ph(h->phase.size() / 2);
    for (size_t i = 0; i < ph.size(); i++) ph[i] = make_float2(h->phase[2 * i], h->phase[2 * i + 1]);
    if ((rc = upload(h, &h->d_phase, ph))) return rc;
  }
  (void)H;
  h->dirty = false;
  return FDOCT_OK;
}

// Bluestein tables for the +i transform of length n: X[k] = c[k] * sum_m (x[m] c[m]) conj(c[k-m]), c[m] = e^(+i pi m^2/n)
// (m^2 taken mod 2n in integers, so the angle stays exact); bhat = forward DFT of the wrapped conj(c), scaled by 1/Mb for
// the unscaled inverse transform that follows it in the kernels.  Computed in double.
void build_bluestein_tables(int n, int Mb, std::vector<float2>& chirp, std::vector<float2>& bhat) {
  std::vector<double> cr(n), ci(n);
  chirp.resize(n);
  for (long long m = 0; m < n; m++) {
    const double ang = kPi * (double)((m * m) % (2LL * n)) / (double)n;
    cr[m] = std::cos(ang);
    ci[m] = std::sin(ang);
    chirp[m] = make_float2((float)cr[m], (float)ci[m]);
  }
  std::vector<double> br(Mb, 0.0), bi(Mb, 0.0);
  for (int m = 0; m < n; m++) {
    br[m] = cr[m];
    bi[m] = -ci[m];
    if (m) {
      br[Mb - m] = cr[m];
      bi[Mb - m] = -ci[m];
    }
  }
  // forward DFT of length Mb (power of two) in double: iterative radix-2
  for (int i = 1, j = 0; i < Mb; i++) {
    int bit = Mb >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      std::swap(br[i], br[j]);
      std::swap(bi[i], bi[j]);
    }
  }
  for (int len = 2; len <= Mb; len <<= 1) {
    const double ang = -2.0 * kPi / (double)len;
    for (int i = 0; i < Mb; i += len)
      for (int k = 0; k < len / 2; k++) {
        const double wr = std::cos(ang * k), wi = std::sin(ang * k);
        const double ur = br[i + k], ui = bi[i + k];
        const double vr = br[i + k + len / 2] * wr - bi[i + k + len / 2] * wi, vi = br[i + k + len / 2] * wi + bi[i + k + len / 2] * wr;
        br[i + k] = ur + vr;
        bi[i + k] = ui + vi;
        br[i + k + len / 2] = ur - vr;
        bi[i + k + len / 2] = ui - vi;
      }
  }
  bhat.resize(Mb);
  for (int m = 0; m < Mb; m++) bhat[m] = make_float2((float)(br[m] / Mb), (float)(bi[m] / Mb));
}

// Device tables of the generic path.
int rebuild_generic_state(fdoct_ctx* h) {
  int rc;
  const int W = h->W, N = h->N, MW = h->W * h->M;
  DEVICE_SCOPE(h);
  {
    std::vector<float> ib, il;
    reciprocal_words(scaled_copy(h->yb.v, plane_scales(h).yb), ib, il);
    if (h->yb.rows == 1) {
      if ((rc = upload(h, &h->d_ib, ib))) return rc;
      if ((rc = upload(h, &h->d_il, il))) return rc;
      if ((rc = dev_alloc(h, &h->d_ib2d, 0))) return rc;
      if ((rc = dev_alloc(h, &h->d_il2d, 0))) return rc;
    } else {
      if ((rc = upload(h, &h->d_ib2d, ib))) return rc;
      if ((rc = upload(h, &h->d_il2d, il))) return rc;
      if ((rc = dev_alloc(h, &h->d_ib, 0))) return rc;
      if ((rc = dev_alloc(h, &h->d_il, 0))) return rc;
    }
  }
  auto up_ref = [&](const RefFrame& f, double scale, float** d) -> int {
    std::vector<float> t(f.v.size());
    for (size_t i = 0; i < t.size(); i++) t[i] = (float)(f.v[i] * scale);
    return upload(h, d, t);
  };
  if ((rc = up_ref(h->yp, plane_scales(h).yp, &h->d_yp))) return rc;
  if ((rc = up_ref(h->yd, plane_scales(h).yd, &h->d_yd))) return rc;
  std::vector<float> w(W), g(MW);
  for (int i = 0; i < W; i++) w[i] = (float)h->win[i];
  for (int i = 0; i < MW; i++) g[i] = (i < N) ? (float)h->frac[i] : 0.f;  // fractionalk[nearestkindex[q]], 0 past its end
  if ((rc = upload(h, &h->d_win_g, w))) return rc;
  if ((rc = upload(h, &h->d_g_g, g))) return rc;
  if ((rc = upload(h, &h->d_idx_g, h->idx))) return rc;
  auto up_tw = [&](int n, float2** d) -> int {
    std::vector<float2> t(n);
    for (int j = 0; j < n; j++) {
      const double a = 2.0 * kPi * (double)j / (double)n;
      t[j] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    return upload(h, d, t);
  };
  if ((rc = up_tw(N, &h->d_twg_n))) return rc;
  if ((N % 2) == 0 && (rc = up_tw(N / 2, &h->d_twg_nh))) return rc;
  if (h->blu_m) {
    const int n = generic_real_half(h) ? N / 2 : N, Mb = h->blu_m;
    std::vector<float2> chirp, bhat;
    build_bluestein_tables(n, Mb, chirp, bhat);
    if ((rc = upload(h, &h->d_blu_chirp, chirp))) return rc;
    if ((rc = upload(h, &h->d_blu_bhat, bhat))) return rc;
    if ((rc = up_tw(Mb, &h->d_twg_blu))) return rc;
  }
  if (h->M > 1) {
    if ((rc = up_tw(W, &h->d_twg_w))) return rc;     // untangle factors of the half-length transforms
    if ((rc = up_tw(MW, &h->d_twg_mw))) return rc;
    if ((rc = up_tw(W / 2, &h->d_twg_wh))) return rc;
    if ((rc = up_tw(MW / 2, &h->d_twg_mwh))) return rc;
  }
  {
    std::vector<float2> ph(h->phase.size() / 2);
    for (size_t i = 0; i < ph.size(); i++) ph[i] = make_float2(h->phase[2 * i], h->phase[2 * i + 1]);
    if ((rc = upload(h, &h->d_phase, ph))) return rc;
  }
  h->dirty = false;
  h->generic_tables_ok = true;
  return FDOCT_OK;
}

// Tables of the wave-per-row kernels: packed gather sources and the twiddle blob
// [N/2 passes][M W/2 passes][W/2 passes][e^(2 pi i k/W), k < W/2][e^(2 pi i k/(M W)), k < W/2][e^(2 pi i k/N), k < D].
int rebuild_wave_state(fdoct_ctx* h) {
  // complex rows (dispersion phase): the final transform runs over the whole row, one gather source per point
  const bool cplx = !h->phase.empty();
  const int W = h->W, M = h->M, N = h->N, MW = W * M, NC = cplx ? N : N / 2, D = h->D;
  std::vector<uint32_t> gi(NC);
  auto src = [&](int q) -> uint32_t { return (q <= 0 || q >= N - 1) ? (uint32_t)MW : (uint32_t)h->idx[q]; };  // main:1164
  for (int n = 0; n < NC; n++) gi[n] = cplx ? src(n) : (src(2 * n) | (src(2 * n + 1) << 16));
  std::vector<float2> tw;
  auto unit = [&](double num, double den) {
    const double ang = 2.0 * kPi * num / den;
    return make_float2((float)std::cos(ang), (float)std::sin(ang));
  };
  auto pass_tables = [&](int n) {
    const WavePlan p = wave_plan(n);
    for (int i = 0; i < p.npass; i++)
      if (p.Ns[i] > 1)
        for (int k = 0; k < p.Ns[i]; k++) tw.push_back(unit((double)k, (double)p.Ns[i] * p.R[i]));
  };
  h->wave_off[0] = (int)tw.size();
  pass_tables(NC);
  h->wave_off[1] = (int)tw.size();
  if (M > 1) pass_tables(MW / 2);
  h->wave_off[2] = (int)tw.size();
  if (M > 1) pass_tables(W / 2);
  h->wave_off[3] = (int)tw.size();
  if (M > 1)
    for (int k = 0; k < W / 2; k++) tw.push_back(unit((double)k, (double)W));
  h->wave_off[4] = (int)tw.size();
  if (M > 1)
    for (int k = 0; k < W / 2; k++) tw.push_back(unit((double)k, (double)MW));
  h->wave_off[5] = (int)tw.size();
  // untangle factors of the real rows: bins below numdisplaypoints, or (displayed beyond N/2: the upper bins mirror) up to N/2
  if (!cplx)
    for (int k = 0; k < (D > N / 2 ? N / 2 + 1 : D); k++) tw.push_back(unit((double)k, (double)N));
  h->wave_tw_count = (int)tw.size();
  int rc;
  if ((rc = upload(h, &h->d_wave_gidx, gi))) return rc;
  if ((rc = upload(h, &h->d_wave_tw, tw))) return rc;
  h->wave_tables_ok = true;
  return FDOCT_OK;
}

int kernel_dtype(int dt) {
  switch (dt) {
    case FDOCT_U8: return FDOCT_K_U8;
    case FDOCT_U16: return FDOCT_K_U16;
    case FDOCT_F32: return FDOCT_K_F32;
    default: return -1;
  }
}

// medianBlur + INTER_AREA binning of device-resident raw frames into a packed, 16-byte-pitched buffer.
// Returns the binned frames in *out / *out_pitch (library workspace).
int run_frontend(fdoct_ctx* h, const void* d_raw, int kdt, int nframes, int raw_w, int raw_h, size_t raw_pitch, int mediann,
                 int binx, int biny, void** out, size_t* out_pitch) {
  if (kdt != FDOCT_K_U8 && kdt != FDOCT_K_U16) return fail(h, FDOCT_ERR_UNSUPPORTED, "the front end takes 8- or 16-bit camera frames");
  if (binx < 1 || biny < 1 || raw_w % binx || raw_h % biny) return fail(h, FDOCT_ERR_INVALID, "frame size must be a multiple of the bin factors");
  if (mediann != 0 && mediann != 3 && mediann != 5 && mediann != 7) return fail(h, FDOCT_ERR_INVALID, "mediann must be 0, 3, 5 or 7");
  // cv::medianBlur takes ksize 3 or 5 only for CV_16U (main:955 would throw): there is no reference behaviour to match
  if (mediann == 7 && kdt == FDOCT_K_U16) return fail(h, FDOCT_ERR_INVALID, "a 7x7 median exists for 8-bit frames only (cv::medianBlur)");
  const size_t es = kdt == FDOCT_K_U8 ? 1 : 2;
  int rc;
  hipStream_t st = h->stream;
  const void* src = d_raw;
  size_t src_pitch = raw_pitch;
  if (mediann > 0) {
    const size_t mp = ((size_t)raw_w * es + 15) & ~(size_t)15;
    if ((rc = dev_reserve(h, &h->ws_med, &h->ws_med_cap, mp * (size_t)raw_h * nframes))) return rc;
    HIP_TRY(h, launch_median(src, (long long)src_pitch, h->ws_med, (long long)mp, kdt, raw_w, raw_h, mediann, nframes, st));
    src = h->ws_med;
    src_pitch = mp;
  }
  const int ow = raw_w / binx, oh = raw_h / biny;
  const size_t op = ((size_t)ow * es + 15) & ~(size_t)15;
  if ((rc = dev_reserve(h, &h->ws_front, &h->ws_front_cap, op * (size_t)oh * nframes))) return rc;
  HIP_TRY(h, launch_bin(src, (long long)src_pitch, h->ws_front, (long long)op, kdt, ow, oh, binx, biny, nframes, st));
  *out = h->ws_front;
  *out_pitch = op;
  return FDOCT_OK;
}

// ---- long-row path (fdoct_big.hip) ----------------------------------------------------------------------------------
// The passes of an n-point transform (n = 2^a 3^b 5^c) as a few groups, each one launch with its data in LDS: the prime
// factors are dealt to G groups so that the groups' lengths come out as equal as they can (16384 = 128 x 128, 4096 = 64 x 64),
// G the smallest count that keeps every length within what a workgroup's tile holds.
bool big_plan_groups(int n, std::vector<fdoct_ctx::BigGroupPlan>& groups) {
  groups.clear();
  std::vector<int> primes;
  int m = n;
  for (int p : {5, 3, 2})
    while (m % p == 0) { primes.push_back(p); m /= p; }
  if (m != 1 || n < 2) return false;
  constexpr int kQmax = BIG_GROUP_TILE_VALUES / 8;   // 8 sub-problems of this many points fill the tile (64 contiguous bytes per element index)
  int G = 1;
  for (double cap = kQmax; cap < (double)n; cap *= kQmax) G++;
  for (; G <= 4; G++) {
    std::vector<long long> prod(G, 1);
    std::vector<std::vector<int>> fac(G);
    for (int p : primes) {  // largest factors first, each to the group that is shortest so far
      int best = 0;
      for (int g = 1; g < G; g++)
        if (prod[g] < prod[best]) best = g;
      prod[best] *= p;
      fac[best].push_back(p);
    }
    bool ok = true;
    for (int g = 0; g < G; g++) ok = ok && prod[g] <= BIG_GROUP_TILE_VALUES / 4;
    if (!ok) continue;
    long long P = 1;
    for (int g = 0; g < G; g++) {
      fdoct_ctx::BigGroupPlan gp;
      gp.P = (int)P;
      gp.Q = (int)prod[g];
      gp.F = (int)(n / (P * prod[g]));
      int twos = 0;
      for (int p : fac[g]) {
        if (p == 2) twos++;
        else gp.rad.push_back(p);
      }
      for (; twos >= 3; twos -= 3) gp.rad.push_back(8);
      if (twos == 2) gp.rad.push_back(4);
      if (twos == 1) gp.rad.push_back(2);
      if ((int)gp.rad.size() > BIG_GROUP_MAX_PASSES || gp.rad.empty()) { ok = false; break; }
      const long long S = (long long)gp.P * gp.F;
      int l2 = 4;
      while (l2 > 0 && (((long long)gp.Q << l2) > BIG_GROUP_TILE_VALUES || (1LL << l2) > S)) l2--;
      gp.log2ts = l2;
      groups.push_back(gp);
      P *= prod[g];
    }
    if (ok) return true;
    groups.clear();
  }
  return false;
}

// DFT plan of one length: Stockham radices when it factors into 2, 3, 5, else Bluestein around a power of two >= 2n - 1.
int big_plan_get(fdoct_ctx* h, int n, fdoct_ctx::BigPlan** out) {
  auto it = h->big_plans.find(n);
  if (it != h->big_plans.end()) {
    *out = &it->second;
    return FDOCT_OK;
  }
  fdoct_ctx::BigPlan p;
  auto radices = [](int len, std::vector<int>& rad) {  // 5s and 3s first, then 8s, then what is left of the power of two
    rad.clear();
    while (len % 5 == 0) { rad.push_back(5); len /= 5; }
    while (len % 3 == 0) { rad.push_back(3); len /= 3; }
    while (len % 8 == 0) { rad.push_back(8); len /= 8; }
    if (len % 4 == 0) { rad.push_back(4); len /= 4; }
    if (len % 2 == 0) { rad.push_back(2); len /= 2; }
    return len == 1;
  };
  int tn = n;
  if (!radices(n, p.rad)) {
    int mb = 1;
    while (mb < 2 * n - 1) mb <<= 1;
    p.mb = mb;
    radices(mb, p.rad);
    tn = mb;
    std::vector<float2> chirp, bhat;
    build_bluestein_tables(n, mb, chirp, bhat);
    int rc;
    if ((rc = upload(h, &p.d_chirp, chirp))) return rc;
    if ((rc = upload(h, &p.d_bhat, bhat))) return rc;
  }
  static const bool per_pass = [] { const char* e = std::getenv("FDOCT_BIG_PER_PASS"); return e && std::atoi(e) != 0; }();  // measurement: round 3's form
  if (!per_pass) big_plan_groups(tn, p.groups);
  std::vector<float2> tw(tn);
  for (int j = 0; j < tn; j++) {
    const double a = 2.0 * kPi * (double)j / (double)tn;
    tw[j] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  int rc;
  if ((rc = upload(h, &p.d_tw, tw))) return rc;
  *out = &h->big_plans.emplace(n, p).first->second;
  return FDOCT_OK;
}

void big_plans_free(fdoct_ctx* h) {
  for (auto& kv : h->big_plans)
    for (float2* p : {kv.second.d_tw, kv.second.d_chirp, kv.second.d_bhat})
      if (p) (void)hipFree(p);
  h->big_plans.clear();
}

// What stands in front of a transform, fused into the loads of its first launch (or run as a kernel of its own where the
// transform is not one of grouped launches): the row of floats read as complex (A4's first transform), the W-point spectrum
// re-packed into the M W-point one (A4), the slope step and lambda -> k gather with the phase (A5 / A6 / A6').
struct BigLoader {
  int load = BIG_LOAD_CPLX;
  const float* yr = nullptr;
  const float2* yc = nullptr;
  const float2* spec = nullptr;
  int ylen = 0, W = 0, bandpass = 0;
  const int32_t* idx = nullptr;
  const float* g = nullptr;
  const float2* phase = nullptr;
};

// X = IDFT_n (+i exponent, unscaled) of `rows` rows; x holds them (ld == null) or is free and the loader supplies them;
// `other` is the second buffer (both hold rows * max(n, mb) values).  *result = the buffer that holds the rows * n result;
// out_limit > 0: only the first out_limit values of each result row are needed (and, with grouped launches, written).
int big_idft(fdoct_ctx* h, float2* x, float2* other, long long rows, int n, float2** result, hipStream_t st, const BigLoader* ld = nullptr,
             int out_limit = 0) {
  fdoct_ctx::BigPlan* p = nullptr;
  int rc;
  if ((rc = big_plan_get(h, n, &p))) return rc;
  auto materialise = [&]() -> int {  // the loader as a kernel of its own, into x
    if (!ld) return FDOCT_OK;
    switch (ld->load) {
      case BIG_LOAD_REAL: HIP_TRY(h, big_launch_real_to_complex(ld->yr, rows * n, x, st)); break;
      case BIG_LOAD_PAD: HIP_TRY(h, big_launch_pad(ld->spec, rows, ld->W, n, ld->bandpass, x, st)); break;
      case BIG_LOAD_RESAMPLE: HIP_TRY(h, big_launch_resample(ld->yr, ld->yc, rows, ld->ylen, n, ld->idx, ld->g, ld->phase, x, st)); break;
      default: break;
    }
    return FDOCT_OK;
  };
  // the passes of one len-point transform over src -> ... -> *last (ping-pong between the two buffers)
  auto passes = [&](float2*& src, float2*& dst, int len, const BigLoader* first_ld, int limit) -> int {
    if (!p->groups.empty()) {
      for (size_t gi = 0; gi < p->groups.size(); gi++) {
        const auto& gp = p->groups[gi];
        BigGroup g{};
        g.src = src; g.dst = dst; g.rows = rows; g.n = len;
        g.P = gp.P; g.Q = gp.Q; g.F = gp.F; g.log2ts = gp.log2ts;
        g.npass = (int)gp.rad.size();
        for (int i = 0; i < g.npass; i++) g.rad[i] = gp.rad[i];
        g.out_limit = (gi + 1 == p->groups.size() && limit > 0) ? limit : len;
        g.tw = p->d_tw;
        g.load = BIG_LOAD_CPLX;
        if (gi == 0 && first_ld) {
          g.load = first_ld->load;
          g.yr = first_ld->yr; g.yc = first_ld->yc; g.ylen = first_ld->ylen;
          g.idx = first_ld->idx; g.g = first_ld->g; g.phase = first_ld->phase;
          g.W = first_ld->W; g.bandpass = first_ld->bandpass;
          if (first_ld->load == BIG_LOAD_PAD) g.src = first_ld->spec;
        }
        HIP_TRY(h, big_launch_fft_group(g, st));
        std::swap(src, dst);
      }
      return FDOCT_OK;
    }
    int Ns = 1;
    for (int R : p->rad) {
      HIP_TRY(h, big_launch_fft_pass(src, dst, rows, len, R, Ns, p->d_tw, st));
      std::swap(src, dst);
      Ns *= R;
    }
    return FDOCT_OK;
  };
  float2 *src = x, *dst = other;
  if (!p->mb) {
    const bool fused = ld && !p->groups.empty();
    if (!fused && (rc = materialise())) return rc;
    if ((rc = passes(src, dst, n, fused ? ld : nullptr, out_limit))) return rc;
    *result = src;
    return FDOCT_OK;
  }
  // Bluestein: u = conj(x c) zero-padded; conj(IDFT u) = DFT(x c); times bhat; IDFT; times c
  if ((rc = materialise())) return rc;
  HIP_TRY(h, big_launch_chirp_in(x, rows, n, p->mb, p->d_chirp, other, st));
  src = other;
  dst = x;
  if ((rc = passes(src, dst, p->mb, nullptr, 0))) return rc;
  HIP_TRY(h, big_launch_conj_mul(src, rows, p->mb, p->d_bhat, st));
  if ((rc = passes(src, dst, p->mb, nullptr, 0))) return rc;
  HIP_TRY(h, big_launch_chirp_out(src, rows, n, p->mb, p->d_chirp, dst, st));
  *result = dst;
  return FDOCT_OK;
}

// The whole chain for device-resident frames on the long-row path, chunk by chunk of whole averaging groups.
int run_big(fdoct_ctx* h, const void* kframes, const float* kframes_lo, int kdt, size_t kpitch, int nframes, bool need_minmax, float* k_mag, float* k_db,
            hipStream_t st) {
  const int W = h->W, H = h->H, N = h->N, D = h->D, M = h->M, A = h->A;
  // the padded spectrum / upsampled row: W + 2 floor((M W - W) / 2) points (main:229) -- M W, or M W - 1 for an odd width under an
  // even multiplier
  const int MW = W + 2 * ((W * M - W) / 2);
  int rc;
  size_t lmax = (size_t)std::max(N, M > 1 ? MW : 0);
  for (int n : {N, M > 1 ? W : 0, M > 1 ? MW : 0}) {
    if (!n) continue;
    fdoct_ctx::BigPlan* p = nullptr;
    if ((rc = big_plan_get(h, n, &p))) return rc;
    lmax = std::max(lmax, (size_t)std::max(n, p->mb));
  }
  const size_t per_group = (size_t)A * H * ((size_t)W * 4 + 2 * lmax * sizeof(float2));
  long long cg = (long long)(((size_t)2 << 30) / per_group);
  const int G = nframes / A;
  if (cg < 1) cg = 1;
  if (cg > G) cg = G;
  const size_t crow = (size_t)cg * A * H;
  if ((rc = dev_reserve(h, &h->ws_big_y, &h->ws_big_y_cap, crow * W * 4))) return rc;
  if ((rc = dev_reserve(h, &h->ws_big_a, &h->ws_big_a_cap, crow * lmax * sizeof(float2)))) return rc;
  if ((rc = dev_reserve(h, &h->ws_big_b, &h->ws_big_b_cap, crow * lmax * sizeof(float2)))) return rc;
  for (long long g0 = 0; g0 < G; g0 += cg) {
    const long long ng = std::min<long long>(cg, G - g0);
    BigArgs a{};
    a.frames = static_cast<const unsigned char*>(kframes) + (size_t)g0 * A * H * kpitch;
    a.frames_lo = kframes_lo ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(kframes_lo) + (size_t)g0 * A * H * kpitch) : nullptr;
    a.pitch_bytes = (long long)kpitch;
    a.in_rows = ng * A * H;
    a.out_rows = ng * H;
    a.dtype = kdt;
    a.W = W; a.H = H; a.N = N; a.D = D; a.M = M; a.A = A;
    a.ib = h->yb.rows == 1 ? h->d_ib : h->d_ib2d;
    a.il = h->yb.rows == 1 ? h->d_il : h->d_il2d;
    a.ib_2d = h->yb.rows > 1;
    a.yp = h->d_yp; a.yp_2d = h->yp.rows > 1;
    a.yd = h->d_yd; a.yd_2d = h->yd.rows > 1;
    a.win = h->d_win_g;
    a.minmax = need_minmax ? h->d_minmax + (size_t)g0 * A : nullptr;
    a.rowwisenormalize = h->cfg.rowwisenormalize;
    a.dcmask = h->cfg.dc_mask;
    a.inv_A = (float)(1.0 / (double)A);
    a.eps = (h->cfg.variant == FDOCT_VARIANT_SIM) ? 1e-6f : 1e-5f;
    a.db_scale = (float)(20.0 / 2.303 * 0.6931471805599453);
    HIP_TRY(h, big_launch_pre(a, h->ws_big_y, st));
    // Buffer discipline of big_idft with a loader: the first launch reads the loader's source and writes `other`, the next one
    // writes `x`, and so on; so the source may live in x (it is dead once the first launch is through) but never in `other`.
    // A transform that cannot fuse its loader (one launch per pass, Bluestein) materialises the rows into x first: there the
    // source must not live in x.
    float2 *bufa = h->ws_big_a, *bufb = h->ws_big_b, *res = nullptr;
    auto fuses = [&](int n, bool* yes) -> int {
      fdoct_ctx::BigPlan* p = nullptr;
      if (int rc2 = big_plan_get(h, n, &p)) return rc2;
      *yes = !p->mb && !p->groups.empty();
      return FDOCT_OK;
    };
    BigLoader rs;                 // A5 / A6 / A6': what the final transform reads
    rs.load = BIG_LOAD_RESAMPLE;
    rs.yr = h->ws_big_y;
    rs.ylen = W;
    rs.idx = h->d_idx_g;
    rs.g = h->d_g_g;
    rs.phase = h->d_phase;
    float2* held = nullptr;       // the buffer the final transform's source rows live in (null: the float rows)
    if (M > 1) {  // A4
      BigLoader l1;
      l1.load = BIG_LOAD_REAL;
      l1.yr = h->ws_big_y;
      if ((rc = big_idft(h, bufa, bufb, a.in_rows, W, &res, st, &l1))) return rc;
      BigLoader l2;
      l2.load = BIG_LOAD_PAD;
      l2.spec = res;
      l2.W = W;
      l2.bandpass = h->bandpass ? 1 : 0;
      float2* spare = (res == bufa) ? bufb : bufa;
      bool f = false;
      if ((rc = fuses(MW, &f))) return rc;
      if ((rc = f ? big_idft(h, res, spare, a.in_rows, MW, &res, st, &l2) : big_idft(h, spare, res, a.in_rows, MW, &res, st, &l2))) return rc;
      rs.yr = nullptr;
      rs.yc = res;
      rs.ylen = MW;
      held = res;
    }
    {  // A5 / A6 / A7; only the first numdisplaypoints bins of the result are needed
      float2* spare = held ? (held == bufa ? bufb : bufa) : bufb;
      float2* mine = held ? held : bufa;
      bool f = false;
      if ((rc = fuses(N, &f))) return rc;
      if ((rc = f ? big_idft(h, mine, spare, a.in_rows, N, &res, st, &rs, D) : big_idft(h, held ? spare : mine, held ? mine : spare, a.in_rows, N, &res, st, &rs, D)))
        return rc;
    }
    HIP_TRY(h, big_launch_post(res, a, k_mag ? k_mag + (size_t)g0 * H * D : nullptr, k_db ? k_db + (size_t)g0 * H * D : nullptr, st));
  }
  return FDOCT_OK;
}

// Can the chain write the reference's D x H layout itself (fused_kernel's TRO instantiations)?  The acquisition
// configurations on the 1024-point row-swap plan: 8/16-bit frames that go to the kernel as they are, 1-row or full-frame
// background, none or the whole-frame normalisation, rows in fours and depth bins in whole write-out steps, 16-byte
// aligned outputs.
bool fused_transposed_store_applies(const fdoct_ctx* h, fdoct_dtype dtype, const void* d_frames, size_t pitch_bytes,
                                    const float* d_out_bscan, const float* d_out_db, int nframes) {
  if (!h->tro_enabled || h->use_generic || h->staged || h->force_general || h->cplx) return false;
  const FusedPlan& p = h->plan;
  if (!fused_tro_compiled(p.kind, p.T, p.WCH)) return false;
  if (dtype != FDOCT_U8 && dtype != FDOCT_U16) return false;
  if (h->fe_median > 0 || h->fe_binx > 1 || h->fe_biny > 1 || h->cfg.movavgn > 0) return false;
  if (h->W != 8 * p.T * p.WCH || !h->yb.rows || h->yp.rows || h->yd.rows || h->cfg.rowwisenormalize) return false;
  const bool normalize = (h->cfg.variant == FDOCT_VARIANT_SIM) || !h->cfg.donotnormalize;
  if ((normalize || h->yb.rows > 1) && h->A != 1) return false;  // (those instantiations exist for one frame per B-scan)
  const size_t es = dtype == FDOCT_U8 ? 1 : 2, valign = dtype == FDOCT_U8 ? 8 : 16;
  const size_t pitch = pitch_bytes ? pitch_bytes : es * (size_t)h->W;
  if (((uintptr_t)d_frames % valign) || (pitch % valign)) return false;
  if ((h->H % 4) || (h->D % fused_tro_step_bins()) || h->D > h->NC) return false;
  // (both words: a full-frame background brings its second word along with the prefetched row -- no LDS plane; a 1-row one needs
  // the plane next to the ring, which then holds one computing wave less)
  if (h->precise_div && h->yb.rows > 1 && !fused_il_half(true, p.WCH)) return false;
  if (const_lds_bytes(h, false, h->precise_div && h->yb.rows == 1, fused_il_half(true, p.WCH)) + (size_t)h->scratch_bytes + fused_tro_ring_bytes(h->D) > 160 * 1024 - 64) return false;
  if (((uintptr_t)d_out_bscan % 16) || ((uintptr_t)d_out_db % 16)) return false;
  if ((long long)(nframes / h->A) * h->H >= 0x7fffffffLL) return false;
  // the write-out addresses one B-scan with 32-bit byte offsets inside a buffer descriptor of 0x7ffffff0 bytes
  if ((size_t)h->D * (size_t)h->H * 4 >= 0x7ffffff0u) return false;
  return true;
}

// ---- dispatch ----------------------------------------------------------------------------------------------------------
// Everything a call decides before it launches anything: which passes run in front of the chain, which kernel family takes it
// and with what.  A function of the handle's state and of the call's geometry only (pointers enter through their alignment), so
// that fdoct_prepare makes the same decisions -- and pays for a run-time compile -- without frames.
struct Route {
  int family = FDOCT_KERNEL_NONE;   // fdoct_kernel: who runs the chain
  bool frontend = false;            // medianBlur + binning pass over the raw frames first (main:953-958)
  bool narrow_f64 = false;          // data_y doubles narrowed once to float (main:987)
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

// Decides the route of a call.  frames_addr / pitch_bytes / out addresses: as the caller gave them (fdoct_prepare: an aligned,
// packed set-up).  May rebuild device tables and compile (hipRTC) -- never launches.
int choose_route(fdoct_ctx* h, fdoct_dtype dtype, uintptr_t frames_addr, size_t pitch_bytes, uintptr_t out_bscan_addr,
                 uintptr_t out_db_addr, fdoct_layout layout, int nframes, Route* r) {
  int rc;
  if (h->dirty && (rc = rebuild_device_state(h))) return rc;
  if (h->D > h->N) return fail(h, FDOCT_ERR_INVALID, "numdisplaypoints > numfftpoints");
  const int W = h->W, H = h->H, D = h->D, A = h->A;
  const long long out_rows = (long long)(nframes / A) * H;
  const size_t es = dtype_size(dtype);
  *r = Route{};
  int kdt = kernel_dtype(dtype);
  const bool normalize = (h->cfg.variant == FDOCT_VARIANT_SIM) || !h->cfg.donotnormalize;
  // with row-wise normalisation on, every non-degenerate row already spans [0,1] and the whole-frame pass (main:1128) is the identity
  r->need_minmax = normalize && !h->cfg.rowwisenormalize;
  // pi / dark frames, the band-pass and the normalisations are compile-time options of the wave-per-row kernel: the library's own
  // instantiations are the plain set-up, a handle that uses one of them gets its kernel from the run-time compiler
  // ... and so are the dispersion phase (complex rows, a full-length final transform) and a display beyond numfftpoints / 2
  const int wave_opt = (h->yp.rows ? FDOCT_WAVE_OPT_PI : 0) | (h->yd.rows ? FDOCT_WAVE_OPT_DARK : 0) |
                       (h->bandpass && h->M > 1 ? FDOCT_WAVE_OPT_BANDPASS : 0) | (h->cfg.rowwisenormalize ? FDOCT_WAVE_OPT_ROWNORM : 0) |
                       (r->need_minmax ? FDOCT_WAVE_OPT_FRAMENORM : 0) | (!h->phase.empty() ? FDOCT_WAVE_OPT_CPLX : 0) |
                       (h->phase.empty() && D > h->N / 2 ? FDOCT_WAVE_OPT_DEEP : 0);
  auto wave_tables = [&]() -> int {  // (the wave tables read the resample table's device copies)
    if (!h->generic_tables_ok) {
      const bool keep = h->use_generic;
      int rc2 = select_generic(h);
      h->use_generic = keep;
      if (rc2) return rc2;
      if ((rc2 = rebuild_generic_state(h))) return rc2;
    }
    return h->wave_tables_ok ? FDOCT_OK : rebuild_wave_state(h);
  };
  // 2 x 2 binning with nothing else in front of the chain, on a configuration the wave-per-row kernel takes: the kernel compiled
  // for the handle does the binning in its own loads (FDOCT_WAVE_OPT_BIN2) and the pass over the raw frames is skipped.
  // (Measured on the shipped shapes, tools/bench_generic.py with and without FDOCT_JIT=0: + 4.5 % on 160-sample 8-bit rows, + 5 % on
  // 640-sample 16-bit rows, - 1 % on 640-sample 8-bit rows -- twenty 2-byte loads per lane cost what the pass saves: those keep the pass.)
  if (h->fe_median == 0 && h->fe_binx == 2 && h->fe_biny == 2 && (dtype == FDOCT_U16 || (dtype == FDOCT_U8 && W <= 320)) && h->cfg.movavgn == 0 &&
      h->jit && h->use_generic && !h->use_big && h->plan_override != -2 && h->phase.empty() && D <= h->N / 2 && !r->need_minmax &&
      (frames_addr % 4 == 0) && (pitch_bytes % 4 == 0) && pitch_bytes >= es * 2 * (size_t)W && out_rows < 0x7fffffffLL &&
      wave_jit_shape_ok(W, h->M, h->N, D)) {
    if ((rc = wave_tables())) return rc;
    const size_t shared = wave_shared_lds_bytes(h->wave_tw_count, W, h->M, h->N, h->yb.rows > 1);
    std::string why;
    hipFunction_t fn = nullptr;
    if (shared + wave_private_lds_bytes(W, h->M, h->N) > 160 * 1024 - 64) {
      // no room for even one wave: the ordinary path (binning pass, then whichever kernel fits)
    } else if (wave_jit_get(W, h->M, h->N, kdt, (D + 63) / 64, wave_opt | FDOCT_WAVE_OPT_BIN2, h->device, &fn, &why) == hipSuccess) {
      r->bin2_in_kernel = true;
      r->jit_fn = fn;
      r->wave_opt = wave_opt | FDOCT_WAVE_OPT_BIN2;
    }
    h->jit_note = why;
  }
  // ---- passes in front of the chain, and what they leave for its kernel to read
  uintptr_t kaddr = frames_addr;
  size_t kpitch = pitch_bytes;
  if (!r->bin2_in_kernel && (h->fe_median > 0 || h->fe_binx > 1 || h->fe_biny > 1)) {
    if (dtype != FDOCT_U8 && dtype != FDOCT_U16)
      return fail(h, FDOCT_ERR_UNSUPPORTED, "the front end (median / binning) takes the camera's 8- or 16-bit frames");
    if (pitch_bytes < es * (size_t)W * h->fe_binx) return fail(h, FDOCT_ERR_INVALID, "pitch smaller than a raw camera row");
    r->frontend = true;
    kaddr = 0;                                                  // a library workspace: aligned
    kpitch = ((size_t)W * es + 15) & ~(size_t)15;
  }
  if (dtype == FDOCT_F64) {
    if (pitch_bytes % 8) return fail(h, FDOCT_ERR_INVALID, "f64 pitch must be a multiple of 8");
    r->narrow_f64 = true;
    kaddr = 0;
    kpitch = (size_t)W * 4;
    kdt = FDOCT_K_F32;
  }
  if (h->cfg.movavgn > 0) {
    r->movavg = true;
    kaddr = 0;
    kpitch = (size_t)W * 4;
    kdt = FDOCT_K_F32;
  }
  r->kdt = kdt;
  r->kpitch = kpitch;
  // the specialised kernels read 16-byte vectors; anything else goes through the generic kernel
  const size_t valign = (kdt == FDOCT_K_U8) ? 8 : 16;
  const bool misaligned = (kaddr % valign) || (kpitch % valign);
  const bool run_generic = h->use_generic || misaligned;
  if (run_generic && !h->generic_tables_ok) {
    const bool keep = h->use_generic;
    rc = select_generic(h);
    h->use_generic = keep;
    if (rc) return rc;
    if ((rc = rebuild_generic_state(h))) return rc;
  }
  if (run_generic && h->staged) return fail(h, FDOCT_ERR_UNSUPPORTED, "staged mode needs a specialised kernel for this configuration");

  const bool transposed = layout == FDOCT_LAYOUT_TRANSPOSED_DxH;
  r->tro = transposed && !run_generic && !r->frontend && !r->narrow_f64 && !r->movavg &&
           fused_transposed_store_applies(h, dtype, reinterpret_cast<const void*>(frames_addr), pitch_bytes,
                                          reinterpret_cast<const float*>(out_bscan_addr), reinterpret_cast<const float*>(out_db_addr), nframes);
  r->transpose_pass = transposed && !r->tro;

  // the acquisition configurations the reference ships: one wave per A-scan (fdoct_wave.hip) instead of one workgroup
  // (frames handed over as doubles carry a low word per sample: the fused any-option, workgroup-per-row and long-row kernels take it)
  const bool wave_scope = run_generic && !h->use_big && h->plan_override != -2 && kdt >= 0 && !r->narrow_f64 &&
                          (kaddr % 4 == 0) && (kpitch % 4 == 0) && out_rows < 0x7fffffffLL;
  if (r->bin2_in_kernel && !wave_scope) return fail(h, FDOCT_ERR_DEVICE, "internal: binning left to a kernel that does not run");
  bool run_wave = r->bin2_in_kernel;
  bool wave_builtin = false;
  if (wave_scope && !run_wave) {
    wave_builtin = wave_opt == 0 && wave_kernel_available(W, h->M, h->N, kdt, D);
    // any other shape the template can take: compiled for this handle's geometry at run time (fdoct_set_jit); the first call
    // (or fdoct_prepare) pays the compile, a refusal falls back to the workgroup-per-row kernel
    if (!wave_builtin && h->jit && wave_jit_shape_ok(W, h->M, h->N, D, wave_opt)) {
      std::string why;
      hipFunction_t fn = nullptr;
      if (wave_jit_get(W, h->M, h->N, kdt, (D + 63) / 64, wave_opt, h->device, &fn, &why) == hipSuccess) {
        r->jit_fn = fn;
        r->wave_opt = wave_opt;
      }
      h->jit_note = why;
    }
    run_wave = wave_builtin || r->jit_fn;
  }
  if (run_wave) {
    if ((rc = wave_tables())) return rc;
    const size_t shared = wave_shared_lds_bytes(h->wave_tw_count, W, h->M, h->N, h->yb.rows > 1, r->wave_opt);
    if (shared + wave_private_lds_bytes(W, h->M, h->N, r->wave_opt) > 160 * 1024 - 64) {  // not even one wave's buffer next to the tables
      if (r->bin2_in_kernel) return fail(h, FDOCT_ERR_DEVICE, "internal: binning left to a kernel that did not launch");
      run_wave = false;
      r->jit_fn = nullptr;
    }
  }
  if (run_wave)
    r->family = r->jit_fn ? FDOCT_KERNEL_WAVE_JIT : FDOCT_KERNEL_WAVE;
  else if (run_generic)
    r->family = h->use_big ? FDOCT_KERNEL_LONG_ROWS : FDOCT_KERNEL_GENERIC;
  else
    r->family = h->staged ? FDOCT_KERNEL_FUSED_STAGED : (r->tro ? FDOCT_KERNEL_FUSED_TRANSPOSED : FDOCT_KERNEL_FUSED);
  return FDOCT_OK;
}

// After the chain's kernel(s) of any family: end-of-kernel event, the transpose pass where the chain did not write D x H itself,
// end-of-call event, and the call's figures for fdoct_get_timing.
int finish_launch(fdoct_ctx* h, const Route& r, const Call& c, bool staged_timing) {
  hipStream_t st = c.st;
  if (h->record_now && h->rec_last) HIP_TRY(h, hipEventRecord(h->ev[2], st));
  if (r.transpose_pass) {
    if (c.d_out_bscan) HIP_TRY(h, launch_transpose(c.k_mag, c.d_out_bscan, h->H, h->D, c.G, st));
    if (c.d_out_db) HIP_TRY(h, launch_transpose(c.k_db, c.d_out_db, h->H, h->D, c.G, st));
  }
  if (h->record_now && h->rec_last) HIP_TRY(h, hipEventRecord(h->ev[3], st));
  h->last_kernel = r.family;
  h->timing.ascans = (uint64_t)c.in_rows;
  h->timing.bytes_in = (uint64_t)c.in_rows * h->W * c.es;
  h->timing.bytes_out = (uint64_t)c.out_rows * h->D * 4 * ((c.d_out_bscan ? 1 : 0) + (c.d_out_db ? 1 : 0));
  h->timing_pending = h->record_now;
  h->timing_staged = staged_timing;
  return FDOCT_OK;
}

float chain_eps(const fdoct_ctx* h) { return (h->cfg.variant == FDOCT_VARIANT_SIM) ? 1e-6f : 1e-5f; }  // sim:949 / main:1222
constexpr float kDbScale = (float)(20.0 / 2.303 * 0.6931471805599453);                                // main:1236, times ln 2 (the kernels use log2)

int launch_family_wave(fdoct_ctx* h, const Route& r, const Call& c) {
  const int W = h->W, H = h->H, D = h->D, A = h->A;
  WaveArgs wa{};
  wa.frames = c.kframes;
  wa.pitch_bytes = (long long)r.kpitch;
  wa.total_out_rows = c.out_rows;
  wa.dtype = r.kdt;
  wa.H = H; wa.D = D; wa.A = A;
  wa.ib = h->yb.rows == 1 ? h->d_ib : h->d_ib2d;
  wa.il = h->yb.rows == 1 ? h->d_il : h->d_il2d;
  wa.ib_2d = h->yb.rows > 1;
  wa.win = h->d_win_g;
  wa.g = h->d_g_g;
  wa.gidx = h->d_wave_gidx;
  wa.tw = h->d_wave_tw;
  wa.tw_count = h->wave_tw_count;
  wa.off_nc = h->wave_off[0]; wa.off_lh = h->wave_off[1]; wa.off_wh = h->wave_off[2];
  wa.off_tww = h->wave_off[3]; wa.off_twmw = h->wave_off[4]; wa.off_twn = h->wave_off[5];
  wa.dcmask = h->cfg.dc_mask;
  wa.inv_A = (float)(1.0 / (double)A);
  wa.eps = chain_eps(h);
  wa.db_scale = kDbScale;
  wa.out_mag = c.k_mag;
  wa.out_db = c.k_db;
  wa.yp = h->d_yp; wa.yp_2d = h->yp.rows > 1;
  wa.yd = h->d_yd; wa.yd_2d = h->yd.rows > 1;
  wa.minmax = r.need_minmax ? h->d_minmax : nullptr;
  wa.phase = h->d_phase;
#ifdef FDOCT_WAVE_PROBE  // measurement build: per-phase cycles of the first workgroups' waves, printed every 50 calls
  {
    static unsigned long long* d_probe = nullptr;
    const size_t pbytes = 4 * 16 * 12 * 8;
    if (!d_probe) {
      (void)hipMalloc(reinterpret_cast<void**>(&d_probe), pbytes);
      (void)hipMemset(d_probe, 0, pbytes);
    }
    wa.probe = d_probe;
    static int calls = 0;
    if (++calls % 50 == 0) {
      std::vector<unsigned long long> v(4 * 16 * 12);
      (void)hipStreamSynchronize(c.st);
      (void)hipMemcpy(v.data(), d_probe, pbytes, hipMemcpyDeviceToHost);
      static const char* names[9] = {"load+A2/A3", "fwd W/2", "re-pack", "inv MW/2", "slope", "gather", "final N/2", "untangle", "epilogue"};
      double tot = 0;
      double sum[9] = {};
      int nw = 0;
      for (int w = 0; w < 64; w++) {
        if (!v[w * 12 + 6]) continue;
        nw++;
        for (int i = 0; i < 9; i++) sum[i] += (double)v[w * 12 + i];
      }
      for (int i = 0; i < 9; i++) tot += sum[i];
      std::fprintf(stderr, "[wave probe] %d waves:", nw);
      for (int i = 0; i < 9; i++) std::fprintf(stderr, " %s %.1f%%", names[i], 100.0 * sum[i] / (tot > 0 ? tot : 1));
      std::fprintf(stderr, "\n");
    }
  }
#endif
  const size_t shared = wave_shared_lds_bytes(wa.tw_count, W, h->M, h->N, wa.ib_2d != 0, r.wave_opt);
  const size_t priv = wave_private_lds_bytes(W, h->M, h->N, r.wave_opt);
  int waves = (int)((160 * 1024 - 64 - shared) / priv);  // >= 1: choose_route
  if (waves > wave_max_waves(W, h->M, h->N, r.wave_opt)) waves = wave_max_waves(W, h->M, h->N, r.wave_opt);
  if (h->block_override && h->block_override / 64 >= 1 && h->block_override / 64 <= waves) waves = h->block_override / 64;
  long long wgrid = h->num_cu;
  const long long need = (c.out_rows + waves - 1) / waves;
  if (h->grid_override > 0) wgrid = h->grid_override;
  if (wgrid > need) wgrid = need;
  if (h->record_now && h->rec_first) HIP_TRY(h, hipEventRecord(h->ev[1], c.st));
  if (r.jit_fn)
    HIP_TRY(h, wave_jit_launch(r.jit_fn, wa, (int)wgrid, waves, shared + (size_t)waves * priv, c.st));
  else
    HIP_TRY(h, launch_wave(W, h->M, h->N, wa, (int)wgrid, waves, shared + (size_t)waves * priv, c.st));
  return finish_launch(h, r, c, false);
}

int launch_family_long_rows(fdoct_ctx* h, const Route& r, const Call& c) {
  int rc;
  if (h->record_now && h->rec_first) HIP_TRY(h, hipEventRecord(h->ev[1], c.st));
  if ((rc = run_big(h, c.kframes, c.kframes_lo, r.kdt, r.kpitch, c.nframes, r.need_minmax, c.k_mag, c.k_db, c.st))) return rc;
  return finish_launch(h, r, c, false);
}

int launch_family_generic(fdoct_ctx* h, const Route& r, const Call& c) {
  const int W = h->W, H = h->H, D = h->D, A = h->A;
  GenericArgs ga{};
  ga.frames = c.kframes;
  ga.frames_lo = c.kframes_lo;
  ga.pitch_bytes = (long long)r.kpitch;
  ga.total_out_rows = c.out_rows;
  ga.dtype = r.kdt;
  ga.W = W; ga.H = H; ga.N = h->N; ga.D = D; ga.M = h->M; ga.A = A;
  ga.L = generic_buffer_len(h);
  ga.real_half = generic_real_half(h) ? 1 : 0;
  ga.ybuf_len = (W + 3) & ~3;
  ga.ib = h->yb.rows == 1 ? h->d_ib : h->d_ib2d;
  ga.il = h->yb.rows == 1 ? h->d_il : h->d_il2d;
  ga.ib_2d = h->yb.rows > 1;
  ga.yp = h->d_yp; ga.yp_2d = h->yp.rows > 1;
  ga.yd = h->d_yd; ga.yd_2d = h->yd.rows > 1;
  ga.win = h->d_win_g;
  ga.g = h->d_g_g;
  ga.idx = h->d_idx_g;
  ga.phase = h->d_phase;
  ga.minmax = r.need_minmax ? h->d_minmax : nullptr;
  ga.tw_n = h->d_twg_n; ga.tw_nh = h->d_twg_nh; ga.tw_w = h->d_twg_w; ga.tw_mw = h->d_twg_mw;
  ga.tw_wh = h->d_twg_wh; ga.tw_mwh = h->d_twg_mwh;
  auto put_plan = [](const std::vector<int>& rad, int* rr, unsigned* mag) {
    unsigned long long ns = 1;
    for (size_t i = 0; i < rad.size(); i++) {
      rr[i] = rad[i];
      mag[i] = (unsigned)(((1ull << 32) + ns - 1) / ns);  // ceil(2^32 / Ns); unused for Ns == 1
      ns *= (unsigned)rad[i];
    }
  };
  put_plan(h->rad_n, ga.rad_n, ga.mag_n);
  if (ga.real_half) put_plan(h->rad_nh, ga.rad_nh, ga.mag_nh);
  ga.npass_nh = (int)h->rad_nh.size();
  put_plan(h->rad_wh, ga.rad_wh, ga.mag_wh);
  put_plan(h->rad_mwh, ga.rad_mwh, ga.mag_mwh);
  ga.npass_n = (int)h->rad_n.size(); ga.npass_wh = (int)h->rad_wh.size(); ga.npass_mwh = (int)h->rad_mwh.size();
  ga.blu_m = h->blu_m;
  if (h->blu_m) {
    put_plan(h->rad_blu, ga.rad_blu, ga.mag_blu);
    ga.npass_blu = (int)h->rad_blu.size();
    ga.blu_chirp = h->d_blu_chirp;
    ga.blu_bhat = h->d_blu_bhat;
    ga.tw_blu = h->d_twg_blu;
  }
  ga.bandpass = h->bandpass ? 1 : 0;
  ga.inplace = h->generic_inplace ? 1 : 0;
  ga.radix16 = h->generic_radix16 ? 1 : 0;
  ga.rowwisenormalize = h->cfg.rowwisenormalize;
  ga.dcmask = h->cfg.dc_mask;
  ga.inv_A = (float)(1.0 / (double)A);
  ga.eps = chain_eps(h);
  ga.db_scale = kDbScale;
  ga.out_mag = c.k_mag;
  ga.out_db = c.k_db;
  const size_t glds = generic_lds_bytes(h);
  int per_cu = (int)((160 * 1024 - 1024) / glds);
  if (per_cu > 6) per_cu = 6;  // generic_kernel is compiled for 6 waves per SIMD = 6 workgroups of 4 waves per CU
  if (per_cu < 1) per_cu = 1;
  long long ggrid = (long long)h->num_cu * per_cu;
  if (ggrid > c.out_rows) ggrid = c.out_rows;
  if (h->record_now && h->rec_first) HIP_TRY(h, hipEventRecord(h->ev[1], c.st));
  HIP_TRY(h, launch_generic(ga, (int)ggrid, glds, c.st));
  return finish_launch(h, r, c, false);
}

#if defined(FDOCT_RUNTIME_ABLATE) || defined(FDOCT_CLOCKPROBE)
#define FDOCT_DEV_BUILD 1
// Measurement builds only (tools/ablate.sh, tools/mkvariant.sh probe): the stage-skipping mask and the in-kernel clock probes.
void dev_build_hooks(FusedArgs& a, hipStream_t st) {
#ifdef FDOCT_RUNTIME_ABLATE
  static const int ablate = [] { const char* ab = std::getenv("FDOCT_ABLATE"); return ab ? std::atoi(ab) : 0; }();
  a.ablate = ablate;
#endif
#ifdef FDOCT_CLOCKPROBE
  static unsigned long long* d_probe = nullptr;
  const size_t pbytes = (32 + 1024) * 8;
  if (!d_probe) {
    (void)hipMalloc(reinterpret_cast<void**>(&d_probe), pbytes);
    (void)hipMemset(d_probe, 0, pbytes);
  }
  a.probe = d_probe;
  static int calls = 0;
  if (++calls % 64 == 0) {
    std::vector<unsigned long long> v(32 + 1024);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(v.data(), d_probe, pbytes, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "[probe]");
    for (int w = 0; w < 16; w++)
      if (v[2 * w + 1]) std::fprintf(stderr, " w%d %.0fus@%.2fGHz", w, v[2 * w + 1] / 100.0, v[2 * w] / (v[2 * w + 1] * 10.0));
    unsigned long long t0 = ~0ull;
    std::vector<double> stv, en;
    for (int b = 0; b < 512; b++)
      if (v[32 + 2 * b]) t0 = std::min(t0, v[32 + 2 * b]);
    for (int b = 0; b < 512; b++)
      if (v[32 + 2 * b]) {
        stv.push_back((v[32 + 2 * b] - t0) / 100.0);
        en.push_back((v[32 + 2 * b + 1] - t0) / 100.0);
      }
    if (!stv.empty()) {
      std::sort(stv.begin(), stv.end());
      std::sort(en.begin(), en.end());
      std::fprintf(stderr, "\n[probe] %zu blocks: start max %.1f us; end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us", stv.size(),
                   stv.back(), en.front(), en[en.size() / 10], en[en.size() / 2], en[en.size() * 9 / 10], en.back());
    }
    std::fprintf(stderr, "\n");
  }
#else
  (void)st;
#endif
}
#endif

int launch_family_fused(fdoct_ctx* h, const Route& r, const Call& c) {
  const int W = h->W, H = h->H, D = h->D, A = h->A;
  int rc;
  hipStream_t st = c.st;
  FusedArgs a{};
  a.frames = c.kframes;
  a.frames_lo = c.kframes_lo;
  a.pitch_bytes = (long long)r.kpitch;
  a.total_out_rows = c.out_rows;
  a.W = W;
  a.H = H;
  a.D = D;
  a.A = A;
  a.split = h->split;
  a.scratch_bytes = h->scratch_bytes;
  a.tw_count = h->tw_count;
  a.ib = h->d_ib;
  a.ib2d = h->d_ib2d_f;
  a.il = h->d_il;
  a.il2d = h->d_il2d_f;
  a.ilp = h->d_il_p;
  a.il16 = h->d_il16;
  a.il16_2d = h->d_il16_2d;
  a.yp = h->d_yp;
  a.yp_2d = h->yp.rows > 1;
  a.yd = h->d_yd;
  a.yd_2d = h->yd.rows > 1;
  a.win = h->d_win;
  a.g = h->d_g;
  a.gidx = h->d_gidx;
  a.tw = h->d_tw;
  a.utw = h->d_utw;
  a.phase = h->d_phase;
  a.minmax = r.need_minmax ? h->d_minmax : nullptr;
  a.rowwisenormalize = h->cfg.rowwisenormalize;
  a.dcmask = h->cfg.dc_mask;
  a.need_rc = (a.ib2d || a.yp_2d || a.yd_2d || a.minmax || a.frames_lo) ? 1 : 0;
  a.inv_A = (float)(1.0 / (double)A);
  a.eps = chain_eps(h);
  a.db_scale = kDbScale;
  a.out_mag = c.k_mag;
  a.out_db = c.k_db;
#ifdef FDOCT_DEV_BUILD
  dev_build_hooks(a, st);
#endif

  const FusedPlan& p = h->plan;
  // the unpredicated fast-path kernel applies to the plain acquisition configuration
  // (a full-frame background keeps the fast path on the row-swap plan: its resident registers prefetch the frame row)
  const bool fast_opts = fused_resident_consts(p.kind, true, A > 1, p.WCH, 0) && c.out_rows < 0x7fffffffLL && !h->staged;
  // (a full-frame background with the two-word reciprocal -- fdoct_set_precise_division -- runs on the any-option
  // kernel: the fast path's prefetch registers hold one word per sample)
  const bool bg_ok = h->yb.rows == 1 || (fast_opts && (!h->precise_div || fused_il_half(true, p.WCH)));
  const bool norm_ok = !a.minmax || fast_opts;  // whole-frame normalisation has a fast-path variant there too
  const bool lean = (r.kdt == FDOCT_K_U16 || r.kdt == FDOCT_K_U8) && W == 8 * p.T * p.WCH && bg_ok && !a.yp && !a.yd &&
                    (!a.rowwisenormalize || fast_opts) && norm_ok && !h->force_general;
  // launch geometry: as many waves per workgroup as LDS and the register budget allow
  const int rpw = 64 / p.T;
  a.lds_planes = fused_resident_consts(p.kind, lean, A > 1, p.WCH, 0) ? 0 : 1;
  // 1/background as two floats (reciprocal_words): always on the any-option kernel, by fdoct_set_precise_division on the fast path
  a.prec = (lean && !h->precise_div) ? 0 : (h->yb.rows == 1 ? 1 : 2);
  // (the averaging fast-path kernels that keep their planes in LDS are bound by its capacity: a fourth 4 W-byte plane would cost
  // C4 a wave per CU, so they read the low words from global memory instead)
  static const bool il_global_ok = [] { const char* e = std::getenv("FDOCT_PREC_IL_LDS"); return !(e && std::atoi(e) != 0); }();  // measurement: 1 = always LDS
  // (only the kernels with more than 32 samples per lane have that form: the others read the row's low words at its top from
  // the LDS plane, resident constants or not -- fused_kernel's ILX)
  if (a.prec == 1 && lean && a.lds_planes && A > 1 && p.WCH > 4 && il_global_ok) a.prec = 3;
  const size_t lds_const = const_lds_bytes(h, a.lds_planes != 0, a.prec == 1, fused_il_half(lean, p.WCH));
  const size_t lds_max = 160 * 1024 - 64;  // the kernel's static row-ticket counter lives in LDS too
  const int max_block = fused_max_block(h->NC, p.T, lean, p.kind);
  int max_waves = max_block / 64;
  int waves = (int)((lds_max - lds_const) / ((size_t)h->scratch_bytes * rpw));
  if (waves > max_waves) waves = max_waves;
  if (h->block_override) {
    int w = h->block_override / 64;
    if (w >= 1 && w <= waves) waves = w;
  }
  if (waves < 1) return fail(h, FDOCT_ERR_UNSUPPORTED, "row does not fit in LDS");
  const size_t lds = lds_const + (size_t)waves * rpw * h->scratch_bytes;
  const int blocks_per_cu = (int)(lds_max / lds) > 0 ? (int)(lds_max / lds) : 1;
  const int wave_cap = (max_block / 64) / waves;  // register budget: max_block threads per CU
  int bpc = blocks_per_cu < wave_cap ? blocks_per_cu : wave_cap;
  if (bpc < 1) bpc = 1;
  long long need = (c.out_rows + (long long)waves * rpw - 1) / ((long long)waves * rpw);
  long long grid = (long long)h->num_cu * bpc;
  if (h->grid_override > 0) grid = h->grid_override;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;

  size_t lds_launch = lds;
  int block_launch = waves * 64;
  if (r.tro) {
    if (!lean) return fail(h, FDOCT_ERR_DEVICE, "internal: fused transposed store chosen for a configuration off the fast path");
    // computing waves + the write-out wave; LDS: constants, one row buffer per computing wave, the ring of finished rows
    const size_t ring = fused_tro_ring_bytes(D);
    const int ww = fused_tro_writer_waves();
    int cw = (int)((lds_max - lds_const - ring) / (size_t)h->scratch_bytes);
    if (cw > max_waves - ww) cw = max_waves - ww;
    if (h->block_override && h->block_override / 64 - ww >= 1 && h->block_override / 64 - ww < cw) cw = h->block_override / 64 - ww;
    if (cw < 1) return fail(h, FDOCT_ERR_DEVICE, "internal: no LDS left for the transposed store's ring");
    block_launch = (cw + ww) * 64;
    lds_launch = lds_const + (size_t)cw * h->scratch_bytes + ring;
    const unsigned tpf = (unsigned)((H + FUSED_TR_ROWS - 1) / FUSED_TR_ROWS);
    const long long tiles = (long long)c.G * tpf;
    grid = h->grid_override > 0 ? h->grid_override : h->num_cu;   // one workgroup per CU (the ring fills its LDS)
    if (grid > tiles) grid = tiles;
    if (!h->d_tro_fault) {
      // (coherent, mapped host memory; the kernels raise the word with a plain system-scope STORE of 1 -- a read-modify-write
      // atomic on host memory would need PCIe AtomicOps on the link and is dropped silently where they are missing: ADVICE r4)
      HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->d_tro_fault), sizeof(unsigned), hipHostMallocCoherent | hipHostMallocMapped));
      *h->d_tro_fault = 0u;
    }
    h->tro_used = true;
    a.tr_fault = h->d_tro_fault;
    a.tro = 1;
    a.tr_tpf = tpf;
    a.tr_tpf_magic = tpf > 1 ? (unsigned)((1ull << 32) / tpf) : 0xffffffffu;
    a.tr_total_tiles = (unsigned)tiles;
  }
  if (h->record_now && h->rec_first) HIP_TRY(h, hipEventRecord(h->ev[1], st));
  if (h->staged) {
    if (!lean || r.kdt != FDOCT_K_U16)
      return fail(h, FDOCT_ERR_UNSUPPORTED, "staged mode is built for the plain u16 acquisition configuration only");
    // one k-linear row per INPUT A-scan between the stages: the resample stage runs over the in_rows input rows as they
    // lie (A = 1), the FFT stage gathers the A rows of an output A-scan
    if ((rc = dev_reserve(h, &h->ws_ylin, &h->ws_ylin_cap, (size_t)c.in_rows * h->NC * sizeof(float2)))) return rc;
    a.ylin = h->ws_ylin;
    h->ylin_rows = c.in_rows;
    {
      FusedArgs a1 = a;
      a1.stage = 1;
      a1.A = 1;
      a1.inv_A = 1.f;
      a1.total_out_rows = c.in_rows;
      long long need1 = (c.in_rows + (long long)waves * rpw - 1) / ((long long)waves * rpw);
      long long grid1 = h->grid_override > 0 ? h->grid_override : (long long)h->num_cu * bpc;
      if (grid1 > need1) grid1 = need1;
      HIP_TRY(h, launch_fused(p, a1, r.kdt, h->cplx, lean, (int)grid1, waves * 64, lds, st));
    }
    if (h->record_now) HIP_TRY(h, hipEventRecord(h->ev[4], st));
    a.stage = 2;
    HIP_TRY(h, launch_fused(p, a, r.kdt, h->cplx, lean, (int)grid, waves * 64, lds, st));
  } else {
    h->ylin_rows = 0;
    HIP_TRY(h, launch_fused(p, a, r.kdt, h->cplx, lean, (int)grid, block_launch, lds_launch, st));
  }
  return finish_launch(h, r, c, h->staged);
}

// Enqueue the whole path for device-resident frames.  d_out_* are row-major or
// transposed per `layout`.
int enqueue_one(fdoct_ctx* h, const void* d_frames, fdoct_dtype dtype, int nframes, size_t pitch_bytes,
                float* d_out_bscan, float* d_out_db, fdoct_layout layout) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!d_frames || nframes <= 0) return fail(h, FDOCT_ERR_INVALID, "no frames");
  if (nframes % h->A) return fail(h, FDOCT_ERR_INVALID, "nframes must be a multiple of averages");
  if (!h->yb.rows) return fail(h, FDOCT_ERR_STATE, "no background set (fdoct_set_background)");
  if (!d_out_bscan && !d_out_db) return fail(h, FDOCT_ERR_INVALID, "no output requested");
  const size_t es = dtype_size(dtype);
  if (!es) return fail(h, FDOCT_ERR_INVALID, "bad dtype");
  if (pitch_bytes == 0) pitch_bytes = es * h->W * h->fe_binx;
  if (pitch_bytes < es * h->W) return fail(h, FDOCT_ERR_INVALID, "pitch smaller than a row");
  DEVICE_SCOPE(h);
  int rc;
  Route r;
  if ((rc = choose_route(h, dtype, (uintptr_t)d_frames, pitch_bytes, (uintptr_t)d_out_bscan, (uintptr_t)d_out_db, layout, nframes, &r))) return rc;
  const int W = h->W, H = h->H, D = h->D, A = h->A;
  Call c;
  c.st = h->stream;
  c.nframes = nframes;
  c.G = nframes / A;
  c.in_rows = (long long)nframes * H;
  c.out_rows = (long long)c.G * H;
  c.es = es;
  c.d_out_bscan = d_out_bscan;
  c.d_out_db = d_out_db;
  c.kframes = d_frames;
  hipStream_t st = c.st;
  if (h->record_now && h->rec_first) HIP_TRY(h, hipEventRecord(h->ev[0], st));
  // ---- passes in front of the chain
  int kdt_now = kernel_dtype(dtype);
  size_t pitch_now = pitch_bytes;
  if (r.frontend) {  // raw camera frames: medianBlur + binning first (main:953-958); the caller's pitch describes the RAW rows
    void* fo = nullptr;
    size_t fp = 0;
    if ((rc = run_frontend(h, d_frames, kdt_now, nframes, W * h->fe_binx, H * h->fe_biny, pitch_bytes, h->fe_median, h->fe_binx, h->fe_biny, &fo, &fp)))
      return rc;
    c.kframes = fo;
    pitch_now = fp;
  }
  if (r.narrow_f64) {  // data_y doubles (main:987): split once into two f32 planes on the device, x = hi + lo
    if ((rc = dev_reserve(h, &h->ws_f32, &h->ws_f32_cap, (size_t)c.in_rows * W * 4))) return rc;
    if ((rc = dev_reserve(h, &h->ws_f32_lo, &h->ws_f32_lo_cap, (size_t)c.in_rows * W * 4))) return rc;
    HIP_TRY(h, launch_f64_split(static_cast<const double*>(d_frames), (long long)(pitch_bytes / 8), h->ws_f32, h->ws_f32_lo, W, c.in_rows, st));
    c.kframes = h->ws_f32;
    c.kframes_lo = h->ws_f32_lo;
    pitch_now = (size_t)W * 4;
    kdt_now = FDOCT_K_F32;
  }
  if (r.movavg) {  // smoothmovavg (main:990-991) runs before everything else, on the raw samples
    if ((rc = dev_reserve(h, &h->ws_mov, &h->ws_mov_cap, (size_t)c.in_rows * W * 4))) return rc;
    HIP_TRY(h, launch_movavg(c.kframes, kdt_now, (long long)pitch_now, W, c.in_rows, h->cfg.movavgn, h->ws_mov, st));
    c.kframes = h->ws_mov;
    if (c.kframes_lo) {  // (the pass is linear: the low words of f64 frames get their own tap sums)
      if ((rc = dev_reserve(h, &h->ws_mov_lo, &h->ws_mov_lo_cap, (size_t)c.in_rows * W * 4))) return rc;
      HIP_TRY(h, launch_movavg(c.kframes_lo, FDOCT_K_F32, (long long)W * 4, W, c.in_rows, h->cfg.movavgn, h->ws_mov_lo, st));
      c.kframes_lo = h->ws_mov_lo;
    }
    pitch_now = (size_t)W * 4;
    kdt_now = FDOCT_K_F32;
  }
  if (kdt_now != r.kdt || pitch_now != r.kpitch) return fail(h, FDOCT_ERR_DEVICE, "internal: the route and the passes in front of the chain disagree");
  if (r.need_minmax) {
    // [nframes] results followed by the fast kernel's per-workgroup partials
    const size_t mm_elems = (size_t)nframes + (size_t)minmax_partial_count(nframes);
    if ((rc = dev_reserve(h, &h->d_minmax, &h->minmax_cap, mm_elems * sizeof(float2)))) return rc;
    HIP_TRY(h, launch_minmax(c.kframes, r.kdt, (long long)r.kpitch, W, H, nframes, h->d_yd, h->yd.rows > 1, h->d_minmax,
                             h->d_minmax + nframes, st));
  }
  // ---- where the chain writes: the caller's arrays, or the transpose pass's input
  c.k_mag = d_out_bscan;
  c.k_db = d_out_db;
  if (r.transpose_pass) {
    const size_t bytes = (size_t)c.out_rows * D * 4;
    if ((rc = dev_reserve(h, &h->ws_tr, &h->ws_tr_cap, bytes * 2))) return rc;
    if (d_out_bscan) c.k_mag = h->ws_tr;
    if (d_out_db) c.k_db = h->ws_tr + (size_t)c.out_rows * D;
  }
  switch (r.family) {
    case FDOCT_KERNEL_WAVE:
    case FDOCT_KERNEL_WAVE_JIT: return launch_family_wave(h, r, c);
    case FDOCT_KERNEL_LONG_ROWS: return launch_family_long_rows(h, r, c);
    case FDOCT_KERNEL_GENERIC: return launch_family_generic(h, r, c);
    default: return launch_family_fused(h, r, c);
  }
}

// The whole path for device-resident frames.  The reference's own layout (bscan is D x H, main:1220) is produced by the
// chain writing row-major B-scans into a library-owned intermediate and a transpose pass; a long batch is cut into chunks of
// whole B-scans whose intermediate (tr_chunk_bytes, reused by every chunk) is small enough to stay in the 256 MB Infinity
// Cache between the two kernels, so that per A-scan only the camera samples and the final image cross HBM.
int enqueue(fdoct_ctx* h, const void* d_frames, fdoct_dtype dtype, int nframes, size_t pitch_bytes,
            float* d_out_bscan, float* d_out_db, fdoct_layout layout) {
  if (!h) return FDOCT_ERR_INVALID;
  if (h->d_tro_fault && *static_cast<volatile unsigned*>(h->d_tro_fault)) {  // raised by an earlier asynchronous call
    *static_cast<volatile unsigned*>(h->d_tro_fault) = 0u;
    return fail(h, FDOCT_ERR_DEVICE, "transposed store: a wave of an earlier call timed out waiting for its tile buffer; that call's results are invalid");
  }
  h->rec_first = h->rec_last = true;
  if (layout != FDOCT_LAYOUT_TRANSPOSED_DxH || nframes <= 0 || (nframes % h->A) || !d_frames)
    return enqueue_one(h, d_frames, dtype, nframes, pitch_bytes, d_out_bscan, d_out_db, layout);
  const int G = nframes / h->A;
  const size_t per_group = (size_t)h->H * h->D * 4 * ((d_out_bscan ? 1 : 0) + (d_out_db ? 1 : 0));
  long long cg = per_group ? (long long)(h->tr_chunk_bytes / per_group) : G;
  if (cg < 1) cg = 1;
  if (G <= cg) return enqueue_one(h, d_frames, dtype, nframes, pitch_bytes, d_out_bscan, d_out_db, layout);
  if (h->yb.rows) {  // (without a background enqueue_one reports the error)
    int rc;
    if (h->dirty && (rc = rebuild_device_state(h))) return rc;
    if (fused_transposed_store_applies(h, dtype, d_frames, pitch_bytes, d_out_bscan, d_out_db, nframes))  // no intermediate at all
      return enqueue_one(h, d_frames, dtype, nframes, pitch_bytes, d_out_bscan, d_out_db, layout);
  }
  const size_t es = dtype_size(dtype);
  if (!es) return fail(h, FDOCT_ERR_INVALID, "bad dtype");
  const size_t pitch = pitch_bytes ? pitch_bytes : es * (size_t)h->W * h->fe_binx;
  const size_t frame_stride = pitch * (size_t)h->H * h->fe_biny;  // raw camera rows when a front end is set
  const size_t out_group = (size_t)h->H * h->D;
  uint64_t ascans = 0, bin = 0, bout = 0;
  for (long long g0 = 0; g0 < G; g0 += cg) {
    const int ng = (int)std::min<long long>(cg, G - g0);
    h->rec_first = g0 == 0;
    h->rec_last = g0 + ng >= G;
    const unsigned char* fr = static_cast<const unsigned char*>(d_frames) + (size_t)g0 * h->A * frame_stride;
    int rc = enqueue_one(h, fr, dtype, ng * h->A, pitch_bytes, d_out_bscan ? d_out_bscan + (size_t)g0 * out_group : nullptr,
                         d_out_db ? d_out_db + (size_t)g0 * out_group : nullptr, layout);
    if (rc) {
      h->rec_first = h->rec_last = true;
      return rc;
    }
    ascans += h->timing.ascans;
    bin += h->timing.bytes_in;
    bout += h->timing.bytes_out;
  }
  h->rec_first = h->rec_last = true;
  h->timing.ascans = ascans;
  h->timing.bytes_in = bin;
  h->timing.bytes_out = bout;
  return FDOCT_OK;
}

}  // namespace

// ------------------------------------------------------------------ C ABI --
extern "C" {

#define FDOCT_STR_(x) #x
#define FDOCT_STR(x) FDOCT_STR_(x)
const char* fdoct_version(void) { return "fdoct-amd " FDOCT_STR(FDOCT_VERSION_MAJOR) "." FDOCT_STR(FDOCT_VERSION_MINOR) " (gfx950)"; }

int fdoct_build_resample_table(int width, int multiplier, int numfftpoints, double lambdamin, double lambdamax,
                               int32_t* nearestkindex, double* fractionalk) {
  if (width < 2 || multiplier < 1 || numfftpoints < 1 || !nearestkindex || !fractionalk) return FDOCT_ERR_INVALID;
  std::vector<int32_t> idx;
  std::vector<double> frac;
  build_resample_table(width, multiplier, numfftpoints, lambdamin, lambdamax, idx, frac);
  std::memcpy(nearestkindex, idx.data(), sizeof(int32_t) * idx.size());
  std::memcpy(fractionalk, frac.data(), sizeof(double) * frac.size());
  return FDOCT_OK;
}

int fdoct_build_window(int width, double* win) {
  if (width < 2 || !win) return FDOCT_ERR_INVALID;
  std::vector<double> w;
  build_barthann(width, w);
  std::memcpy(win, w.data(), sizeof(double) * w.size());
  return FDOCT_OK;
}

const char* fdoct_last_error(fdoct_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int fdoct_create(const fdoct_config* cfg, fdoct_handle* out) {
  if (!cfg || !out) return fail(nullptr, FDOCT_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(fdoct_config)) return fail(nullptr, FDOCT_ERR_INVALID, "fdoct_config.struct_size mismatch");
  if (cfg->width < 8 || cfg->height < 1 || cfg->numfftpoints < 8)
    return fail(nullptr, FDOCT_ERR_INVALID, "width/height/numfftpoints out of range");
  if (cfg->numdisplaypoints < 1 || cfg->numdisplaypoints > cfg->numfftpoints)
    return fail(nullptr, FDOCT_ERR_INVALID, "numdisplaypoints out of range");
  if (!(cfg->lambdamax > cfg->lambdamin) || !(cfg->lambdamin > 0))
    return fail(nullptr, FDOCT_ERR_INVALID, "lambda range");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, FDOCT_ERR_DEVICE, "no HIP device: this library has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, FDOCT_ERR_INVALID, "device ordinal out of range");

  fdoct_ctx* h = new (std::nothrow) fdoct_ctx();
  if (!h) return fail(nullptr, FDOCT_ERR_NOMEM, "out of memory");
  h->cfg = *cfg;
  h->W = cfg->width;
  h->H = cfg->height;
  h->N = cfg->numfftpoints;
  h->D = cfg->numdisplaypoints;
  h->M = cfg->increasefftpointsmultiplier > 0 ? cfg->increasefftpointsmultiplier : 1;
  h->A = cfg->averages > 0 ? cfg->averages : 1;
  if (cfg->variant == FDOCT_VARIANT_SIM) {  // sim:936-947 copies, it never accumulates: the chain runs one frame per B-scan
    h->sim_group = h->A;
    h->A = 1;
  }
  h->device = cfg->device;
  auto bail = [&](int code, const std::string& m) {
    g_create_error = m;
    fdoct_destroy(h);
    return code;
  };
  DeviceScope scope(h->device);  // the caller's current device is restored on every return path
  if (scope.err != hipSuccess) return bail(FDOCT_ERR_DEVICE, "hipSetDevice failed");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) return bail(FDOCT_ERR_DEVICE, "hipGetDeviceProperties failed");
  h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return bail(FDOCT_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess)
    return bail(FDOCT_ERR_DEVICE, "hipStreamCreate failed");
  h->stream = h->own_stream;
  for (auto& ev : h->ev)
    if (hipEventCreate(&ev) != hipSuccess) return bail(FDOCT_ERR_DEVICE, "hipEventCreate failed");

  build_resample_table(h->W, h->M, h->N, cfg->lambdamin, cfg->lambdamax, h->idx, h->frac);
  build_barthann(h->W, h->win);
  int rc = select_plan(h);
  if (rc) return bail(rc, h->err);
  builtin_jet(h->lut);
  if (const char* e = std::getenv("FDOCT_NO_TRO")) h->tro_enabled = std::atoi(e) == 0;
  if (const char* e = std::getenv("FDOCT_JIT")) h->jit = std::atoi(e) != 0;
  if (const char* e = std::getenv("FDOCT_PRECISE_DIVISION")) h->precise_div = std::atoi(e) != 0;
  if (const char* e = std::getenv("FDOCT_TR_CHUNK_MB")) {  // tuning aid (tools/layout_bench.py): 0 = one chunk
    const long long mb = std::atoll(e);
    h->tr_chunk_bytes = mb > 0 ? (size_t)mb << 20 : ~(size_t)0 >> 1;
  }
  *out = h;
  return FDOCT_OK;
}

int fdoct_destroy(fdoct_handle h) {
  if (!h) return FDOCT_OK;
  DeviceScope scope(h->device);
  if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
  if (h->stream && h->stream != h->own_stream) (void)hipStreamSynchronize(h->stream);  // work we enqueued on the caller's stream
  if (h->s_in) (void)hipStreamSynchronize(h->s_in);
  if (h->s_out) (void)hipStreamSynchronize(h->s_out);
  void* ptrs[] = {h->d_ib, h->d_ib2d, h->d_ib2d_f, h->d_il, h->d_il2d, h->d_il2d_f, h->d_il_p, h->d_il16, h->d_il16_2d, h->d_yp, h->d_yd, h->d_win, h->d_g, h->d_gidx, h->d_tw, h->d_utw,
                  h->d_phase, h->d_minmax, h->ws_in, h->ws_f32, h->ws_f32_lo, h->ws_mov_lo, h->ws_out0, h->ws_out1, h->ws_tr, h->ws_ylin,
                  h->d_win_g, h->d_g_g, h->d_idx_g, h->d_wave_gidx, h->d_wave_tw, h->d_blu_chirp, h->d_blu_bhat, h->d_twg_blu, h->d_twg_n, h->d_twg_nh, h->d_twg_w, h->d_twg_mw, h->d_twg_wh, h->d_twg_mwh, h->ws_mov, h->ws_front, h->ws_med, h->ws_raw, h->ws_sim,
                  h->d_lut, h->d_disp_part, h->ws_disp_in, h->ws_disp_in2, h->ws_disp_out};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (void* p : {(void*)h->ws_big_y, (void*)h->ws_big_a, (void*)h->ws_big_b})
    if (p) (void)hipFree(p);
  big_plans_free(h);
  if (h->d_tro_fault) (void)hipHostFree(h->d_tro_fault);
  for (auto& ev : h->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (int b = 0; b < 2; b++) {
    for (hipEvent_t e : {h->pe_in[b], h->pe_k[b], h->pe_out[b]})
      if (e) (void)hipEventDestroy(e);
    for (void* p : {h->pl_in[b], (void*)h->pl_mag[b], (void*)h->pl_db[b]})
      if (p) (void)hipFree(p);
  }
  if (h->s_in) (void)hipStreamDestroy(h->s_in);
  if (h->s_out) (void)hipStreamDestroy(h->s_out);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return FDOCT_OK;
}

int fdoct_set_stream(fdoct_handle h, void* hip_stream) {
  if (!h) return FDOCT_ERR_INVALID;
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return FDOCT_OK;
}

int fdoct_set_background(fdoct_handle h, const void* data, fdoct_dtype dtype, int rows, size_t pitch_bytes) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!data) return fail(h, FDOCT_ERR_INVALID, "background data is null");
  return copy_ref_frame(h, h->yb, data, dtype, rows, pitch_bytes);
}
int fdoct_set_pi_frame(fdoct_handle h, const void* data, fdoct_dtype dtype, int rows, size_t pitch_bytes) {
  if (!h) return FDOCT_ERR_INVALID;
  return copy_ref_frame(h, h->yp, data, dtype, rows, pitch_bytes);
}
int fdoct_set_dark(fdoct_handle h, const void* data, fdoct_dtype dtype, int rows, size_t pitch_bytes) {
  if (!h) return FDOCT_ERR_INVALID;
  return copy_ref_frame(h, h->yd, data, dtype, rows, pitch_bytes);
}

int fdoct_set_window(fdoct_handle h, const double* win, int n) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!win) {
    build_barthann(h->W, h->win);
    h->custom_win = false;
  } else {
    // W entries whatever the zero-pad multiplier is: the window is applied before the upsampling (main:1142, 1146)
    if (n != h->W) return fail(h, FDOCT_ERR_INVALID, "window length must equal width");
    h->win.assign(win, win + n);
    h->custom_win = true;
  }
  h->dirty = true;
  return FDOCT_OK;
}

int fdoct_set_resample_table(fdoct_handle h, const int32_t* nearestkindex, const double* fractionalk, int n) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!nearestkindex || !fractionalk || n != h->N) return fail(h, FDOCT_ERR_INVALID, "table length must equal numfftpoints");
  for (int i = 0; i < n; i++)
    if (nearestkindex[i] < 0 || nearestkindex[i] >= h->W * h->M)
      return fail(h, FDOCT_ERR_INVALID, "nearestkindex entry outside the row");
  h->idx.assign(nearestkindex, nearestkindex + n);
  h->frac.assign(fractionalk, fractionalk + n);
  h->custom_table = true;
  h->dirty = true;
  return FDOCT_OK;
}

int fdoct_set_lambda_range(fdoct_handle h, double lambdamin, double lambdamax) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!(lambdamax > lambdamin) || !(lambdamin > 0)) return fail(h, FDOCT_ERR_INVALID, "lambda range");
  h->cfg.lambdamin = lambdamin;
  h->cfg.lambdamax = lambdamax;
  build_resample_table(h->W, h->M, h->N, lambdamin, lambdamax, h->idx, h->frac);
  h->custom_table = false;
  h->dirty = true;
  return FDOCT_OK;
}

int fdoct_set_dispersion_phase(fdoct_handle h, const float* cos_sin_pairs, int n) {
  if (!h) return FDOCT_ERR_INVALID;
  std::vector<float> old = h->phase;
  if (!cos_sin_pairs) {
    h->phase.clear();
  } else {
    if (n != h->N) return fail(h, FDOCT_ERR_INVALID, "phase length must equal numfftpoints");
    h->phase.assign(cos_sin_pairs, cos_sin_pairs + 2 * (size_t)n);
  }
  h->dirty = true;
  int rc = select_plan(h);
  if (rc) {  // no kernel for the complex path at this size: keep the previous state usable
    const std::string msg = h->err;
    h->phase.swap(old);
    (void)select_plan(h);
    h->err = msg;
  }
  return rc;
}

int fdoct_get_resample_table(fdoct_handle h, int32_t* nearestkindex, double* fractionalk, int n) {
  if (!h) return FDOCT_ERR_INVALID;
  if (n != h->N) return fail(h, FDOCT_ERR_INVALID, "table length must equal numfftpoints");
  if (nearestkindex) std::memcpy(nearestkindex, h->idx.data(), sizeof(int32_t) * n);
  if (fractionalk) std::memcpy(fractionalk, h->frac.data(), sizeof(double) * n);
  return FDOCT_OK;
}

int fdoct_get_window(fdoct_handle h, double* win, int n) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!win || n != (int)h->win.size()) return fail(h, FDOCT_ERR_INVALID, "window length must equal width");
  std::memcpy(win, h->win.data(), sizeof(double) * n);
  return FDOCT_OK;
}

// The sim variant with averages = S > 1 (sim:936-947): of every S frames the reference keeps the LAST one's magnitudes (copyTo,
// no accumulate, no division).  Gathers those frames -- frame g S + S - 1 for every group g -- into a packed device buffer
// with the caller's row pitch (one strided copy on the handle's stream, from host or device memory) and re-points the call
// at it: nframes becomes the number of groups, the frames device-resident.  (The frame on which the reference EMITS, the
// (S + 1)-th of its loop, is computed and dropped there, sim:944-947: it never reaches an output, so it is the caller's to
// skip.)  A no-op for S = 1 and for the main variant.
static int sim_last_frames(fdoct_ctx* h, const void** frames, fdoct_memspace* space, fdoct_dtype dtype, int* nframes, size_t pitch_bytes) {
  const int S = h->sim_group;
  if (S <= 1) return FDOCT_OK;
  if (!*frames || *nframes <= 0) return fail(h, FDOCT_ERR_INVALID, "no frames");
  if (*nframes % S) return fail(h, FDOCT_ERR_INVALID, "nframes must be a multiple of averages");
  const size_t es = dtype_size(dtype);
  if (!es) return fail(h, FDOCT_ERR_INVALID, "bad dtype");
  const size_t pitch = pitch_bytes ? pitch_bytes : es * (size_t)h->W * h->fe_binx;
  const size_t frame_bytes = pitch * (size_t)h->H * h->fe_biny;  // raw camera rows when a front end is set
  const int G = *nframes / S;
  DEVICE_SCOPE(h);
  int rc;
  if ((rc = dev_reserve(h, &h->ws_sim, &h->ws_sim_cap, frame_bytes * (size_t)G))) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(h->ws_sim, frame_bytes, static_cast<const unsigned char*>(*frames) + (size_t)(S - 1) * frame_bytes, (size_t)S * frame_bytes,
                              frame_bytes, (size_t)G, *space == FDOCT_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->stream));
  *frames = h->ws_sim;
  *space = FDOCT_MEM_DEVICE;
  *nframes = G;
  return FDOCT_OK;
}

int fdoct_process_async(fdoct_handle h, const void* d_frames, fdoct_dtype dtype, int nframes, size_t pitch_bytes,
                        float* d_out_bscan, float* d_out_db, fdoct_layout layout) {
  if (!h) return FDOCT_ERR_INVALID;
  fdoct_memspace space = FDOCT_MEM_DEVICE;
  if (int rc = sim_last_frames(h, &d_frames, &space, dtype, &nframes, pitch_bytes)) return rc;
  h->record_now = h->async_timing;
  return enqueue(h, d_frames, dtype, nframes, pitch_bytes, d_out_bscan, d_out_db, layout);
}

int fdoct_set_timing(fdoct_handle h, int on) {
  if (!h) return FDOCT_ERR_INVALID;
  h->async_timing = on != 0;
  return FDOCT_OK;
}

// Did a wave of a transposed-store launch give up waiting (FusedArgs::tr_fault)?  The word lives in pinned host memory, so
// this is a plain read: after a synchronisation point it is final for the work synchronised on, anywhere else (the next
// enqueue, fdoct_get_timing -- callers of the async API who wait on their own stream or event) it reports what has been
// raised so far.  The fault is reported once and cleared.
static int check_tro_fault(fdoct_ctx* h) {
  if (!h->d_tro_fault) return FDOCT_OK;
  volatile unsigned* w = h->d_tro_fault;
  if (*w) {
    *w = 0u;
    return fail(h, FDOCT_ERR_DEVICE, "transposed store: a wave timed out waiting for its tile buffer; results of the transposed-layout calls since the last check are invalid");
  }
  return FDOCT_OK;
}

int fdoct_synchronize(fdoct_handle h) {
  if (!h) return FDOCT_ERR_INVALID;
  DEVICE_SCOPE(h);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return check_tro_fault(h);
}

// Host buffers in, host buffers out, more than one chunk of work: the batch is cut into chunks of whole averaging
// groups and pipelined over three streams -- chunk c+1 uploads while chunk c computes and chunk c-1 downloads (the
// two PCIe directions and the kernels overlap when the caller's buffers are pinned, e.g. from fdoct_host_alloc;
// pageable buffers still work, the runtime then stages them and the host thread serialises the copies).
static int process_pipelined_impl(fdoct_ctx* h, const unsigned char* frames, fdoct_dtype dtype, int nframes, size_t src_pitch,
                                  size_t row_bytes, long long rows_per_frame, float* out_bscan, float* out_db, fdoct_layout layout,
                                  int frames_per_chunk);

static int process_pipelined(fdoct_ctx* h, const unsigned char* frames, fdoct_dtype dtype, int nframes, size_t src_pitch,
                             size_t row_bytes, long long rows_per_frame, float* out_bscan, float* out_db, fdoct_layout layout,
                             int frames_per_chunk) {
  const int rc = process_pipelined_impl(h, frames, dtype, nframes, src_pitch, row_bytes, rows_per_frame, out_bscan, out_db, layout,
                                        frames_per_chunk);
  if (rc != FDOCT_OK) {  // leave nothing in flight that still points at the caller's buffers or the chunk slots
    if (h->s_in) (void)hipStreamSynchronize(h->s_in);
    (void)hipStreamSynchronize(h->stream);
    if (h->s_out) (void)hipStreamSynchronize(h->s_out);
  }
  return rc;
}

static int process_pipelined_impl(fdoct_ctx* h, const unsigned char* frames, fdoct_dtype dtype, int nframes, size_t src_pitch,
                                  size_t row_bytes, long long rows_per_frame, float* out_bscan, float* out_db, fdoct_layout layout,
                                  int frames_per_chunk) {
  int rc;
  if (!h->s_in) {
    HIP_TRY(h, hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
    HIP_TRY(h, hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++) {
      HIP_TRY(h, hipEventCreateWithFlags(&h->pe_in[b], hipEventDisableTiming));
      HIP_TRY(h, hipEventCreateWithFlags(&h->pe_k[b], hipEventDisableTiming));
      HIP_TRY(h, hipEventCreateWithFlags(&h->pe_out[b], hipEventDisableTiming));
    }
  }
  const size_t packed = (row_bytes + 15) & ~(size_t)15;
  const size_t out_per_group = (size_t)h->H * h->D;  // output floats per averaging group: chunks are whole groups (H D / A per input
                                                     // frame is not an integer in general -- 251 lines, 18 bins, 16 averages)
  const hipStream_t s_k = h->stream;
  h->record_now = false;
  uint64_t sum_in = 0, sum_out = 0;  // fdoct_get_timing reports the whole batch, not the last chunk
  for (int f0 = 0, c = 0; f0 < nframes; f0 += frames_per_chunk, c++) {
    const int b = c & 1;
    const int nf = std::min(frames_per_chunk, nframes - f0);
    const size_t in_rows = (size_t)nf * rows_per_frame;
    const size_t out_elems = (size_t)(nf / h->A) * out_per_group;
    if ((rc = dev_reserve(h, &h->pl_in[b], &h->pl_in_cap[b], packed * in_rows))) return rc;
    if (out_bscan && (rc = dev_reserve(h, &h->pl_mag[b], &h->pl_mag_cap[b], out_elems * 4))) return rc;
    if (out_db && (rc = dev_reserve(h, &h->pl_db[b], &h->pl_db_cap[b], out_elems * 4))) return rc;
    if (c >= 2) HIP_TRY(h, hipStreamWaitEvent(h->s_in, h->pe_k[b], 0));   // chunk c-2 has consumed this input slot
    HIP_TRY(h, hipMemcpy2DAsync(h->pl_in[b], packed, frames + (size_t)f0 * rows_per_frame * src_pitch, src_pitch, row_bytes, in_rows,
                                hipMemcpyHostToDevice, h->s_in));
    HIP_TRY(h, hipEventRecord(h->pe_in[b], h->s_in));
    HIP_TRY(h, hipStreamWaitEvent(s_k, h->pe_in[b], 0));
    if (c >= 2) HIP_TRY(h, hipStreamWaitEvent(s_k, h->pe_out[b], 0));     // chunk c-2 has left this output slot
    if ((rc = enqueue(h, h->pl_in[b], dtype, nf, packed, out_bscan ? h->pl_mag[b] : nullptr, out_db ? h->pl_db[b] : nullptr, layout)))
      return rc;
    sum_in += h->timing.bytes_in;
    sum_out += h->timing.bytes_out;
    HIP_TRY(h, hipEventRecord(h->pe_k[b], s_k));
    HIP_TRY(h, hipStreamWaitEvent(h->s_out, h->pe_k[b], 0));
    const size_t o0 = (size_t)(f0 / h->A) * out_per_group;
    if (out_bscan) HIP_TRY(h, hipMemcpyAsync(out_bscan + o0, h->pl_mag[b], out_elems * 4, hipMemcpyDeviceToHost, h->s_out));
    if (out_db) HIP_TRY(h, hipMemcpyAsync(out_db + o0, h->pl_db[b], out_elems * 4, hipMemcpyDeviceToHost, h->s_out));
    HIP_TRY(h, hipEventRecord(h->pe_out[b], h->s_out));
  }
  HIP_TRY(h, hipStreamSynchronize(h->s_out));
  HIP_TRY(h, hipStreamSynchronize(s_k));
  h->timing.bytes_in = sum_in;
  h->timing.bytes_out = sum_out;
  return FDOCT_OK;
}

void* fdoct_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void fdoct_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int fdoct_process(fdoct_handle h, const void* frames, fdoct_dtype dtype, fdoct_memspace space, int nframes,
                  size_t pitch_bytes, float* out_bscan, float* out_db, fdoct_memspace out_space,
                  fdoct_layout layout) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!frames || nframes <= 0) return fail(h, FDOCT_ERR_INVALID, "no frames");
  const size_t es = dtype_size(dtype);
  if (!es) return fail(h, FDOCT_ERR_INVALID, "bad dtype");
  if (int rc0 = sim_last_frames(h, &frames, &space, dtype, &nframes, pitch_bytes)) return rc0;
  if (nframes % h->A) return fail(h, FDOCT_ERR_INVALID, "nframes must be a multiple of averages");
  DEVICE_SCOPE(h);
  int rc;
  const long long in_rows = (long long)nframes * h->H * h->fe_biny;   // raw camera rows when a front end is set
  const size_t row_samples = (size_t)h->W * h->fe_binx;
  const size_t out_elems = (size_t)(nframes / h->A) * h->H * h->D;
  const void* d_frames = frames;
  size_t d_pitch = pitch_bytes ? pitch_bytes : es * row_samples;
  if (space == FDOCT_MEM_HOST && out_space == FDOCT_MEM_HOST) {
    // chunks of ~32 MB of input, whole averaging groups; two chunks or more are worth pipelining
    const size_t frame_bytes = es * row_samples * (size_t)h->H * h->fe_biny;
    long long fpc = (long long)((32u << 20) / (frame_bytes ? frame_bytes : 1));
    fpc = std::max<long long>(fpc / h->A, 1) * h->A;
    if (nframes >= 2 * fpc) {
      const auto t0 = std::chrono::steady_clock::now();
      rc = process_pipelined(h, static_cast<const unsigned char*>(frames), dtype, nframes, d_pitch, es * row_samples,
                             (long long)h->H * h->fe_biny, out_bscan, out_db, layout, (int)fpc);
      if (rc) return rc;
      if ((rc = check_tro_fault(h))) return rc;
      h->timing_pending = false;  // no per-call device events here: report the wall time of the whole pipeline
      h->timing.last_process_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      h->timing.last_kernel_ms = h->timing.resample_stage_ms = h->timing.fft_stage_ms = 0.0;
      h->timing.ascans = (uint64_t)in_rows;
      h->record_now = true;
      return FDOCT_OK;
    }
  }
  if (space == FDOCT_MEM_HOST) {
    // stage into an aligned, packed device buffer (PCIe-inclusive path)
    const size_t packed = (es * row_samples + 15) & ~(size_t)15;
    if ((rc = dev_reserve(h, &h->ws_in, &h->ws_in_cap, packed * (size_t)in_rows))) return rc;
    HIP_TRY(h, hipMemcpy2DAsync(h->ws_in, packed, frames, d_pitch, es * row_samples, (size_t)in_rows, hipMemcpyHostToDevice,
                                h->stream));
    d_frames = h->ws_in;
    d_pitch = packed;
  }
  float* d_mag = out_bscan;
  float* d_db = out_db;
  if (out_space == FDOCT_MEM_HOST) {
    if (out_bscan) {
      if ((rc = dev_reserve(h, &h->ws_out0, &h->ws_out0_cap, out_elems * 4))) return rc;
      d_mag = h->ws_out0;
    }
    if (out_db) {
      if ((rc = dev_reserve(h, &h->ws_out1, &h->ws_out1_cap, out_elems * 4))) return rc;
      d_db = h->ws_out1;
    }
  }
  h->record_now = true;
  if ((rc = enqueue(h, d_frames, dtype, nframes, d_pitch, d_mag, d_db, layout))) return rc;
  if (out_space == FDOCT_MEM_HOST) {
    if (out_bscan) HIP_TRY(h, hipMemcpyAsync(out_bscan, d_mag, out_elems * 4, hipMemcpyDeviceToHost, h->stream));
    if (out_db) HIP_TRY(h, hipMemcpyAsync(out_db, d_db, out_elems * 4, hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return check_tro_fault(h);
}

int fdoct_get_timing(fdoct_handle h, fdoct_timing* t) {
  if (!h || !t) return FDOCT_ERR_INVALID;
  if (h->timing_pending) {
    DEVICE_SCOPE(h);
    HIP_TRY(h, hipEventSynchronize(h->ev[3]));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[3]));
    h->timing.last_process_ms = ms;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[1], h->ev[2]));
    h->timing.last_kernel_ms = ms;
    h->timing.resample_stage_ms = h->timing.fft_stage_ms = 0.0;
    if (h->timing_staged) {
      HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[1], h->ev[4]));
      h->timing.resample_stage_ms = ms;
      HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[4], h->ev[2]));
      h->timing.fft_stage_ms = ms;
    }
    h->timing_pending = false;
    if (int frc = check_tro_fault(h)) return frc;
  } else if (!h->record_now) {
    h->timing.last_process_ms = h->timing.last_kernel_ms = h->timing.resample_stage_ms = h->timing.fft_stage_ms = 0.0;
  }
  *t = h->timing;
  return FDOCT_OK;
}

int fdoct_set_launch(fdoct_handle h, int threads_per_block, int blocks) {
  if (!h) return FDOCT_ERR_INVALID;
  if (threads_per_block < 0 || threads_per_block % 64 || threads_per_block > FDOCT_MAX_BLOCK || blocks < 0)
    return fail(h, FDOCT_ERR_INVALID, "threads_per_block must be a multiple of 64 up to the build's block limit");
  h->block_override = threads_per_block;
  h->grid_override = blocks;
  return FDOCT_OK;
}

int fdoct_set_plan(fdoct_handle h, int plan_id, int force_general_kernel) {
  if (!h) return FDOCT_ERR_INVALID;
  FusedPlan q{};
  if (plan_id >= 0 && !fused_plan_get(plan_id, &q)) return fail(h, FDOCT_ERR_INVALID, "unknown plan id");
  h->plan_override = plan_id;
  h->force_general = force_general_kernel != 0;
  h->dirty = true;
  return select_plan(h);
}

int fdoct_set_frontend(fdoct_handle h, int mediann, int binx, int biny) {
  if (!h) return FDOCT_ERR_INVALID;
  if (binx < 1 || biny < 1 || (mediann != 0 && mediann != 3 && mediann != 5 && mediann != 7))
    return fail(h, FDOCT_ERR_INVALID, "mediann must be 0/3/5/7 and the bin factors >= 1");
  h->fe_median = mediann;
  h->fe_binx = binx;
  h->fe_biny = biny;
  return FDOCT_OK;
}

int fdoct_frontend(fdoct_handle h, const void* raw, fdoct_dtype dtype, int nframes, int raw_w, int raw_h, size_t pitch_bytes,
                   int mediann, int binx, int biny, void* out) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!raw || !out || nframes <= 0 || raw_w <= 0 || raw_h <= 0) return fail(h, FDOCT_ERR_INVALID, "fdoct_frontend: bad arguments");
  if (dtype != FDOCT_U8 && dtype != FDOCT_U16)
    return fail(h, FDOCT_ERR_UNSUPPORTED, "the front end takes 8- or 16-bit camera frames");
  const int kdt = kernel_dtype(dtype);
  const size_t es = dtype_size(dtype);
  if (pitch_bytes == 0) pitch_bytes = es * raw_w;
  if (pitch_bytes < es * raw_w) return fail(h, FDOCT_ERR_INVALID, "pitch smaller than a raw camera row");
  DEVICE_SCOPE(h);
  int rc;
  const size_t packed = (es * raw_w + 15) & ~(size_t)15;
  if ((rc = dev_reserve(h, &h->ws_raw, &h->ws_raw_cap, packed * (size_t)raw_h * nframes))) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(h->ws_raw, packed, raw, pitch_bytes, es * raw_w, (size_t)raw_h * nframes, hipMemcpyHostToDevice,
                              h->stream));
  void* fo = nullptr;
  size_t fp = 0;
  if ((rc = run_frontend(h, h->ws_raw, kdt, nframes, raw_w, raw_h, packed, mediann, binx, biny, &fo, &fp))) return rc;
  const int ow = raw_w / binx, oh = raw_h / biny;
  HIP_TRY(h, hipMemcpy2DAsync(out, es * ow, fo, fp, es * ow, (size_t)oh * nframes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FDOCT_OK;
}

int fdoct_set_colormap(fdoct_handle h, const unsigned char* bgr256) {
  if (!h) return FDOCT_ERR_INVALID;
  if (bgr256)
    std::memcpy(h->lut, bgr256, 768);
  else
    builtin_jet(h->lut);
  h->lut_dirty = true;
  return FDOCT_OK;
}

int fdoct_get_colormap(fdoct_handle h, unsigned char* bgr256) {
  if (!h || !bgr256) return FDOCT_ERR_INVALID;
  std::memcpy(bgr256, h->lut, 768);
  return FDOCT_OK;
}

int fdoct_display(fdoct_handle h, const float* bscandb, fdoct_memspace in_mem, int nbscans, int rows, int cols,
                  double bscanthreshold, int clampupper, unsigned char* out_gray, unsigned char* out_bgr,
                  fdoct_memspace out_mem) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!bscandb || nbscans <= 0 || rows <= 0 || cols <= 0) return fail(h, FDOCT_ERR_INVALID, "fdoct_display: bad arguments");
  if (!out_gray && !out_bgr) return fail(h, FDOCT_ERR_INVALID, "fdoct_display: no output requested");
  if (clampupper && (rows <= 5 || cols <= 5)) return fail(h, FDOCT_ERR_INVALID, "clampupper needs a B-scan larger than 5x5");
  DEVICE_SCOPE(h);
  int rc;
  const long long count = (long long)rows * cols;
  const size_t total = (size_t)count * nbscans;
  const float* d_in = bscandb;
  if (in_mem == FDOCT_MEM_HOST) {
    if ((rc = dev_reserve(h, &h->ws_disp_in, &h->ws_disp_in_cap, total * sizeof(float)))) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->ws_disp_in, bscandb, total * sizeof(float), hipMemcpyHostToDevice, h->stream));
    d_in = static_cast<const float*>(h->ws_disp_in);
  }
  if (h->lut_dirty || !h->d_lut) {
    if (!h->d_lut && (rc = dev_alloc(h, &h->d_lut, 768))) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_lut, h->lut, 768, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // h->lut may change right after we return
    h->lut_dirty = false;
  }
  if ((rc = dev_reserve(h, &h->d_disp_part, &h->disp_part_cap, (size_t)nbscans * display_parts(count) * 2 * sizeof(double))))
    return rc;
  unsigned char *d_gray = out_gray, *d_bgr = out_bgr;
  if (out_mem == FDOCT_MEM_HOST) {
    const size_t need = (out_gray ? total : 0) + (out_bgr ? 3 * total : 0);
    if ((rc = dev_reserve(h, &h->ws_disp_out, &h->ws_disp_out_cap, need))) return rc;
    unsigned char* w = static_cast<unsigned char*>(h->ws_disp_out);
    d_bgr = out_bgr ? w : nullptr;  // colour first: its 12-byte groups stay 4-byte aligned
    d_gray = out_gray ? w + (out_bgr ? 3 * total : 0) : nullptr;
  }
  const long long clamp_at = clampupper ? 5LL * cols + 5 : -1;  // bscandisp.at<double>(5, 5), main:1252
  HIP_TRY(h, launch_display(d_in, count, nbscans, bscanthreshold, clamp_at, h->d_disp_part, h->d_lut, d_gray, d_bgr, h->stream));
  if (out_mem == FDOCT_MEM_HOST) {
    if (out_gray) HIP_TRY(h, hipMemcpyAsync(out_gray, d_gray, total, hipMemcpyDeviceToHost, h->stream));
    if (out_bgr) HIP_TRY(h, hipMemcpyAsync(out_bgr, d_bgr, 3 * total, hipMemcpyDeviceToHost, h->stream));
  }
  if (in_mem == FDOCT_MEM_HOST || out_mem == FDOCT_MEM_HOST) HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FDOCT_OK;
}

int fdoct_lockin_db(fdoct_handle h, const float* bscan, const float* jscan, fdoct_memspace mem, int nbscans, size_t count,
                    float* out_db) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!bscan || !jscan || !out_db || nbscans <= 0 || count == 0) return fail(h, FDOCT_ERR_INVALID, "fdoct_lockin_db: bad arguments");
  DEVICE_SCOPE(h);
  const size_t total = count * (size_t)nbscans;
  if (mem == FDOCT_MEM_DEVICE) {
    HIP_TRY(h, launch_lockin_db(bscan, jscan, (long long)total, (long long)count, out_db, h->stream));
    return FDOCT_OK;
  }
  int rc;
  if ((rc = dev_reserve(h, &h->ws_disp_in, &h->ws_disp_in_cap, total * sizeof(float)))) return rc;
  if ((rc = dev_reserve(h, &h->ws_disp_in2, &h->ws_disp_in2_cap, count * sizeof(float)))) return rc;
  if ((rc = dev_reserve(h, &h->ws_disp_out, &h->ws_disp_out_cap, total * sizeof(float)))) return rc;
  HIP_TRY(h, hipMemcpyAsync(h->ws_disp_in, bscan, total * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->ws_disp_in2, jscan, count * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, launch_lockin_db(static_cast<const float*>(h->ws_disp_in), static_cast<const float*>(h->ws_disp_in2), (long long)total,
                              (long long)count, static_cast<float*>(h->ws_disp_out), h->stream));
  HIP_TRY(h, hipMemcpyAsync(out_db, h->ws_disp_out, total * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FDOCT_OK;
}

int fdoct_set_averages(fdoct_handle h, int averages) {
  if (!h) return FDOCT_ERR_INVALID;
  if (averages < 1) return fail(h, FDOCT_ERR_INVALID, "averages must be >= 1");
  if (h->cfg.variant == FDOCT_VARIANT_SIM)
    h->sim_group = averages;  // (sim_last_frames: the last frame of every group is what the reference emits)
  else
    h->A = averages;  // a launch parameter only: no table depends on it
  h->cfg.averages = averages;
  return FDOCT_OK;
}

int fdoct_set_bandpass(fdoct_handle h, int on) {
  if (!h) return FDOCT_ERR_INVALID;
  h->bandpass = on != 0;  // takes effect inside the zero-pad stage (increasefftpointsmultiplier > 1), as in the reference
  return FDOCT_OK;
}

int fdoct_set_staged(fdoct_handle h, int on) {
  if (!h) return FDOCT_ERR_INVALID;
  h->staged = on != 0;
  return FDOCT_OK;
}

int fdoct_prepare(fdoct_handle h, fdoct_dtype dtype, fdoct_layout layout) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!h->yb.rows) return fail(h, FDOCT_ERR_STATE, "no background set (fdoct_set_background)");
  const size_t es = dtype_size(dtype);
  if (!es) return fail(h, FDOCT_ERR_INVALID, "bad dtype");
  DEVICE_SCOPE(h);
  // the call fdoct_process* will see: aligned device frames, packed rows, one averaging group, both images asked for
  Route r;
  const int rc = choose_route(h, dtype, 0, es * (size_t)h->W * h->fe_binx, 0, 0, layout, h->A, &r);
  if (rc) return rc;
  return r.family;
}

int fdoct_set_precise_division(fdoct_handle h, int on) {
  if (!h) return FDOCT_ERR_INVALID;
  h->precise_div = on != 0;
  return FDOCT_OK;
}

int fdoct_set_jit(fdoct_handle h, int on) {
  if (!h) return FDOCT_ERR_INVALID;
  h->jit = on != 0;
  return FDOCT_OK;
}

int fdoct_last_kernel(fdoct_handle h) { return h ? h->last_kernel : FDOCT_KERNEL_NONE; }

const char* fdoct_jit_note(fdoct_handle h) { return h ? h->jit_note.c_str() : ""; }

long long fdoct_jit_compile_check(int width, int multiplier, int numfftpoints, int numdisplaypoints, fdoct_dtype dtype, const char* gcn_arch,
                                  char* why, int why_len) {
  std::string reason;
  long long n = -1;
  const int kdt = kernel_dtype(dtype);
  if (!gcn_arch || !*gcn_arch)
    reason = "no architecture named";
  else if (!wave_jit_shape_ok(width, multiplier, numfftpoints, numdisplaypoints))
    reason = "the wave-per-row kernel cannot take this shape";
  else
    n = wave_jit_compile_only(width, multiplier, numfftpoints, kdt, (numdisplaypoints + 63) / 64, 0, gcn_arch, &reason);
  if (why && why_len > 0) std::snprintf(why, (size_t)why_len, "%s", reason.c_str());
  return n;
}

int fdoct_device_count(void) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) return 0;
  return ndev > 0 ? ndev : 0;
}

int fdoct_shard_frames(int nframes_total, int averages, int part, int nparts, int* first, int* count) {
  if (nframes_total < 0 || averages < 1 || nparts < 1 || part < 0 || part >= nparts || !first || !count) return FDOCT_ERR_INVALID;
  const int groups = nframes_total / averages, base = groups / nparts, extra = groups % nparts;
  const int g0 = part * base + (part < extra ? part : extra);
  const int g1 = g0 + base + (part < extra ? 1 : 0);
  *first = g0 * averages;
  *count = (g1 - g0) * averages;
  return FDOCT_OK;
}

int fdoct_clone_to_device(fdoct_handle h, int device, fdoct_handle* out) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!out) return fail(h, FDOCT_ERR_INVALID, "fdoct_clone_to_device: null output");
  *out = nullptr;
  fdoct_config cfg = h->cfg;
  cfg.device = device;
  cfg.averages = h->cfg.variant == FDOCT_VARIANT_SIM ? h->sim_group : h->A;
  fdoct_handle c = nullptr;
  int rc = fdoct_create(&cfg, &c);
  if (rc) return fail(h, rc, std::string("fdoct_clone_to_device: ") + fdoct_last_error(nullptr));
  // constant state: the same blob the multi-process set-up broadcasts (host memory; the device tables of the clone are
  // built on ITS device at the first call, like any handle's)
  size_t used = 0;
  rc = fdoct_export_state(h, nullptr, 0, &used);
  std::vector<unsigned char> blob(used);
  if (!rc) rc = fdoct_export_state(h, blob.data(), blob.size(), &used);
  if (!rc) rc = fdoct_import_state(c, blob.data(), blob.size());
  if (rc) {
    const std::string msg = rc == FDOCT_ERR_INVALID ? c->err : h->err;
    fdoct_destroy(c);
    return fail(h, rc, "fdoct_clone_to_device: " + msg);
  }
  // run-time settings
  c->fe_median = h->fe_median;
  c->fe_binx = h->fe_binx;
  c->fe_biny = h->fe_biny;
  c->bandpass = h->bandpass;
  c->jit = h->jit;
  c->precise_div = h->precise_div;
  c->staged = h->staged;
  c->async_timing = h->async_timing;
  c->force_general = h->force_general;
  c->plan_override = h->plan_override;
  c->block_override = h->block_override;
  c->grid_override = h->grid_override;
  std::memcpy(c->lut, h->lut, sizeof c->lut);
  c->lut_dirty = true;
  c->dirty = true;
  rc = select_plan(c);
  if (rc) {
    const std::string msg = c->err;
    fdoct_destroy(c);
    return fail(h, rc, "fdoct_clone_to_device: " + msg);
  }
  *out = c;
  return FDOCT_OK;
}

int fdoct_get_ylin(fdoct_handle h, long long row0, int nrows, double* out) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!out || nrows <= 0 || row0 < 0) return fail(h, FDOCT_ERR_INVALID, "fdoct_get_ylin: bad arguments");
  if (!h->ylin_rows || !h->ws_ylin) return fail(h, FDOCT_ERR_STATE, "fdoct_get_ylin: the last run was not a staged one (fdoct_set_staged)");
  if (row0 + nrows > h->ylin_rows) return fail(h, FDOCT_ERR_INVALID, "fdoct_get_ylin: rows past the end of the last batch");
  DEVICE_SCOPE(h);
  const int NC = h->NC, N = h->N;
  std::vector<float2> z((size_t)nrows * NC);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(z.data(), h->ws_ylin + (size_t)row0 * NC, z.size() * sizeof(float2), hipMemcpyDeviceToHost));
  for (int r = 0; r < nrows; r++) {
    const float2* zr = z.data() + (size_t)r * NC;
    double* o = out + (size_t)r * N;
    if (h->cplx) {
      // complex path: the stage stores data_ylin[q] * (cos, sin)[q]; the phasors have unit modulus
      for (int q = 0; q < N; q++) o[q] = (double)zr[q].x * h->phase[2 * q] + (double)zr[q].y * h->phase[2 * q + 1];
    } else {
      // real path: FFT point n packs (data_ylin[2n], data_ylin[2n+1]), with the untangle's 1/2 folded into the window
      for (int n = 0; n < NC; n++) {
        o[2 * n] = 2.0 * (double)zr[n].x;
        o[2 * n + 1] = 2.0 * (double)zr[n].y;
      }
    }
  }
  return FDOCT_OK;
}

// ---- state blob (format 2): 12 x int32 header {magic, version, W, H, N, M, yb_rows, yp_rows, yd_rows, nphase floats,
// window length, flags (bit 0 custom window, bit 1 custom resample table)}, then yb, yp, yd, window, fractionalk as
// doubles, nearestkindex as int32, the phase as floats.  The window has W entries whatever the zero-pad multiplier is:
// it is applied before the upsampling (main:1142 precedes main:1146).
static const int32_t kStateMagic = 0x46444f43;  // 'FDOC'
static const int32_t kStateVersion = 2;
static const size_t kStateHeader = 12 * sizeof(int32_t);

// The set-up broadcast of the multi-GPU arrangement with one process per GPU, for a C / C++ host that has an RCCL
// communicator: rank `root` exports its constant state, the blob's size and bytes travel as two ncclBroadcast calls over
// device buffers on the handle's stream, the other ranks import it.  librccl is looked up at run time (dlopen: a host that
// never calls this does not need it); the communicator and its lifetime are the caller's.  A one-rank communicator is a plain
// export / import round trip.
int fdoct_broadcast_state_rccl(fdoct_handle h, void* nccl_comm, int root) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!nccl_comm) return fail(h, FDOCT_ERR_INVALID, "fdoct_broadcast_state_rccl: null communicator");
  // (the entry points used, by their documented C signatures: ncclResult_t is an int with 0 = success, ncclUint8 = 1,
  // ncclUint64 = 5 in every NCCL / RCCL 2.x header -- ncclGetVersion is asked before those values are relied on)
  typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
  typedef int (*rank_fn)(const void*, int*);
  typedef int (*version_fn)(int*);
  typedef const char* (*err_fn)(int);
  static void* lib = nullptr;
  static bcast_fn nccl_broadcast = nullptr;
  static rank_fn nccl_rank = nullptr, nccl_count = nullptr;
  static err_fn nccl_err = nullptr;
  static int nccl_major = 0;
  static std::once_flag once;  // (handles on several devices may call this from several host threads)
  std::call_once(once, [] {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "libnccl.so.2"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) return;
    nccl_broadcast = reinterpret_cast<bcast_fn>(dlsym(lib, "ncclBroadcast"));
    nccl_rank = reinterpret_cast<rank_fn>(dlsym(lib, "ncclCommUserRank"));
    nccl_count = reinterpret_cast<rank_fn>(dlsym(lib, "ncclCommCount"));
    nccl_err = reinterpret_cast<err_fn>(dlsym(lib, "ncclGetErrorString"));
    if (version_fn getv = reinterpret_cast<version_fn>(dlsym(lib, "ncclGetVersion"))) {
      int v = 0;   // 2.x.y: 2000 + 100 x + y up to 2.8, 20000 + 100 x + y from 2.9
      if (getv(&v) == 0) nccl_major = v >= 10000 ? v / 10000 : v / 1000;
    }
  });
  if (!lib) return fail(h, FDOCT_ERR_UNSUPPORTED, "fdoct_broadcast_state_rccl: librccl.so not found");
  if (!nccl_broadcast || !nccl_rank || !nccl_count) return fail(h, FDOCT_ERR_UNSUPPORTED, "fdoct_broadcast_state_rccl: librccl.so lacks ncclBroadcast / ncclCommUserRank / ncclCommCount");
  if (nccl_major != 2) return fail(h, FDOCT_ERR_UNSUPPORTED, "fdoct_broadcast_state_rccl: the collective library does not report a 2.x version (ncclGetVersion); its datatype codes are not known here");
  auto nccl_try = [&](int r, const char* what) -> int {
    if (r == 0) return FDOCT_OK;
    return fail(h, FDOCT_ERR_DEVICE, std::string(what) + ": " + (nccl_err ? nccl_err(r) : "RCCL error"));
  };
  DEVICE_SCOPE(h);
  int rank = -1, count = 0, rc;
  if ((rc = nccl_try(nccl_rank(nccl_comm, &rank), "ncclCommUserRank"))) return rc;
  if ((rc = nccl_try(nccl_count(nccl_comm, &count), "ncclCommCount"))) return rc;
  if (root < 0 || root >= count) return fail(h, FDOCT_ERR_INVALID, "fdoct_broadcast_state_rccl: root outside the communicator");
  // Everything that can fail on ONE rank alone happens before the first collective: the export on the root, the
  // device buffers -- a size word and one fixed-size chunk the blob travels through, so that nothing is allocated between the
  // collectives.  A root that cannot export still takes part in the size broadcast, with 0, and every rank returns an error; a
  // rank that fails HERE returns without having entered a collective -- the others then wait in theirs, and the caller must
  // ncclCommAbort the communicator (include/fdoct.h says so).
  const size_t kChunk = (size_t)4 << 20;
  std::vector<unsigned char> blob;
  unsigned long long nbytes = 0;
  int root_rc = FDOCT_OK;
  if (rank == root) {
    size_t used = 0;
    root_rc = fdoct_export_state(h, nullptr, 0, &used);
    if (!root_rc) {
      blob.resize(used);
      root_rc = fdoct_export_state(h, blob.data(), blob.size(), &used);
    }
    nbytes = root_rc ? 0 : used;
  }
  hipStream_t st = h->stream;
  unsigned long long* d_n = nullptr;
  unsigned char* d_chunk = nullptr;
  auto cleanup = [&]() {
    if (d_n) (void)hipFree(d_n);
    if (d_chunk) (void)hipFree(d_chunk);
  };
  if (hipMalloc(reinterpret_cast<void**>(&d_n), sizeof nbytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d_chunk), kChunk) != hipSuccess) {
    cleanup();
    return fail(h, FDOCT_ERR_NOMEM, "fdoct_broadcast_state_rccl: no device memory for the staging buffers (no collective was entered: abort the communicator)");
  }
  if (hipMemcpyAsync(d_n, &nbytes, sizeof nbytes, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    cleanup();
    return fail(h, FDOCT_ERR_DEVICE, "fdoct_broadcast_state_rccl: copy of the blob size failed (no collective was entered: abort the communicator)");
  }
  // ---- collective 1: the size
  if ((rc = nccl_try(nccl_broadcast(d_n, d_n, 1, /*ncclUint64*/ 5, root, nccl_comm, st), "ncclBroadcast (size)"))) {
    cleanup();
    return rc;
  }
  if (hipMemcpyAsync(&nbytes, d_n, sizeof nbytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
      nbytes > (1ull << 34)) {
    cleanup();
    return fail(h, FDOCT_ERR_DEVICE, "fdoct_broadcast_state_rccl: implausible blob size from the root");
  }
  if (nbytes == 0) {  // the root had nothing to send: every rank returns an error, nobody is left in a later broadcast
    cleanup();
    return root_rc ? root_rc : fail(h, FDOCT_ERR_INVALID, "fdoct_broadcast_state_rccl: the root rank could not export its state");
  }
  // ---- collectives 2 ...: the blob, chunk by chunk through the staging buffer.  A copy that fails from here on does not
  // take this rank out of the remaining broadcasts (the others would wait in them): the error is returned at the end.
  if (rank != root) {
    try {
      blob.resize(nbytes);
    } catch (...) {
      blob.clear();   // (no host memory: keep taking part, report afterwards)
    }
  }
  int late = FDOCT_OK;
  for (unsigned long long off = 0; off < nbytes; off += kChunk) {
    const size_t n = (size_t)std::min<unsigned long long>(kChunk, nbytes - off);
    if (rank == root && hipMemcpyAsync(d_chunk, blob.data() + off, n, hipMemcpyHostToDevice, st) != hipSuccess && !late)
      late = fail(h, FDOCT_ERR_DEVICE, "fdoct_broadcast_state_rccl: upload of the blob failed");
    if ((rc = nccl_try(nccl_broadcast(d_chunk, d_chunk, n, /*ncclUint8*/ 1, root, nccl_comm, st), "ncclBroadcast (blob)"))) {
      cleanup();
      return rc;   // (the collective itself failed: the communicator is in error for every rank)
    }
    if (rank != root && !blob.empty()) {
      if ((hipMemcpyAsync(blob.data() + off, d_chunk, n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) && !late)
        late = fail(h, FDOCT_ERR_DEVICE, "fdoct_broadcast_state_rccl: download of the blob failed");
    } else if (hipStreamSynchronize(st) != hipSuccess && !late) {  // the chunk buffer is reused by the next round
      late = fail(h, FDOCT_ERR_DEVICE, "fdoct_broadcast_state_rccl: stream error during the blob broadcast");
    }
  }
  cleanup();
  if (late) return late;
  if (blob.size() != nbytes) return fail(h, FDOCT_ERR_NOMEM, "fdoct_broadcast_state_rccl: no host memory for the blob");
  // (the root imports its own blob too: every rank ends in the state the blob describes, validated the same way)
  return fdoct_import_state(h, blob.data(), blob.size());
}

int fdoct_export_state(fdoct_handle h, void* buf, size_t cap, size_t* used) {
  if (!h || !used) return FDOCT_ERR_INVALID;
  const size_t need = kStateHeader + (h->yb.v.size() + h->yp.v.size() + h->yd.v.size() + h->win.size() + h->frac.size()) * 8 +
                      h->idx.size() * 4 + h->phase.size() * 4;
  *used = need;
  if (!buf) return FDOCT_OK;
  if (cap < need) return fail(h, FDOCT_ERR_INVALID, "state buffer too small");
  unsigned char* p = static_cast<unsigned char*>(buf);
  const int32_t hdr[12] = {kStateMagic, kStateVersion, h->W, h->H, h->N, h->M, h->yb.rows, h->yp.rows, h->yd.rows,
                           (int32_t)h->phase.size(), (int32_t)h->win.size(), (h->custom_win ? 1 : 0) | (h->custom_table ? 2 : 0)};
  std::memcpy(p, hdr, sizeof hdr);
  p += sizeof hdr;
  auto put = [&](const void* src, size_t bytes) {
    if (bytes) std::memcpy(p, src, bytes);
    p += bytes;
  };
  put(h->yb.v.data(), h->yb.v.size() * 8);
  put(h->yp.v.data(), h->yp.v.size() * 8);
  put(h->yd.v.data(), h->yd.v.size() * 8);
  put(h->win.data(), h->win.size() * 8);
  put(h->frac.data(), h->frac.size() * 8);
  put(h->idx.data(), h->idx.size() * 4);
  put(h->phase.data(), h->phase.size() * 4);
  return FDOCT_OK;
}

// Everything is parsed and checked into temporaries first: a blob that fails any check leaves the handle untouched.
int fdoct_import_state(fdoct_handle h, const void* buf, size_t len) {
  if (!h) return FDOCT_ERR_INVALID;
  if (!buf || len < kStateHeader) return fail(h, FDOCT_ERR_INVALID, "state blob too short");
  const unsigned char* p = static_cast<const unsigned char*>(buf);
  int32_t hdr[12];
  std::memcpy(hdr, p, sizeof hdr);
  p += sizeof hdr;
  if (hdr[0] != kStateMagic) return fail(h, FDOCT_ERR_INVALID, "not a state blob");
  if (hdr[1] != kStateVersion) return fail(h, FDOCT_ERR_INVALID, "state blob of another format version");
  if (hdr[2] != h->W || hdr[3] != h->H || hdr[4] != h->N || hdr[5] != h->M)
    return fail(h, FDOCT_ERR_INVALID, "state blob does not match this handle's geometry (width, height, numfftpoints, multiplier)");
  auto rows_ok = [&](int r) { return r == 0 || r == 1 || r == h->H; };
  if (!rows_ok(hdr[6]) || !rows_ok(hdr[7]) || !rows_ok(hdr[8])) return fail(h, FDOCT_ERR_INVALID, "corrupt state blob: reference frame rows");
  if (hdr[9] != 0 && hdr[9] != 2 * h->N) return fail(h, FDOCT_ERR_INVALID, "corrupt state blob: phase length must be 0 or 2 x numfftpoints");
  if (hdr[10] != h->W) return fail(h, FDOCT_ERR_INVALID, "corrupt state blob: window length must equal width");
  const size_t nyb = (size_t)hdr[6] * h->W, nyp = (size_t)hdr[7] * h->W, nyd = (size_t)hdr[8] * h->W;
  const size_t nwin = (size_t)hdr[10], nph = (size_t)hdr[9];
  const size_t need = kStateHeader + (nyb + nyp + nyd + nwin + (size_t)h->N) * 8 + (size_t)h->N * 4 + nph * 4;
  if (len < need) return fail(h, FDOCT_ERR_INVALID, "state blob truncated");
  auto get = [&](void* dst, size_t bytes) {
    if (bytes) std::memcpy(dst, p, bytes);
    p += bytes;
  };
  RefFrame yb, yp, yd;
  std::vector<double> win(nwin), frac(h->N);
  std::vector<int32_t> idx(h->N);
  std::vector<float> phase(nph);
  yb.v.resize(nyb); yb.rows = hdr[6]; get(yb.v.data(), nyb * 8);
  yp.v.resize(nyp); yp.rows = hdr[7]; get(yp.v.data(), nyp * 8);
  yd.v.resize(nyd); yd.rows = hdr[8]; get(yd.v.data(), nyd * 8);
  get(win.data(), nwin * 8);
  get(frac.data(), (size_t)h->N * 8);
  get(idx.data(), (size_t)h->N * 4);
  get(phase.data(), nph * 4);
  for (int32_t v : idx)
    if (v < 0 || v >= h->W * h->M)  // the kernels index LDS with these
      return fail(h, FDOCT_ERR_INVALID, "corrupt state blob: nearestkindex entry outside the row");
  // the complex path must have a kernel at this size before anything is committed (as fdoct_set_dispersion_phase checks)
  std::vector<float> old_phase = h->phase;
  h->phase.swap(phase);
  int rc = select_plan(h);
  if (rc) {
    const std::string msg = h->err;
    h->phase.swap(old_phase);
    (void)select_plan(h);
    h->err = msg;
    return rc;
  }
  h->yb = std::move(yb);
  h->yp = std::move(yp);
  h->yd = std::move(yd);
  h->win.swap(win);
  h->frac.swap(frac);
  h->idx.swap(idx);
  h->custom_win = (hdr[11] & 1) != 0;
  h->custom_table = (hdr[11] & 2) != 0;
  h->dirty = true;
  return FDOCT_OK;
}

}  // extern "C"
