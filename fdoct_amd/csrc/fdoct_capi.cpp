// fdoct_capi.cpp -- the extern "C" boundary (include/fdoct.h) over the HIP kernels.
//
// Replaces the processing block of the reference's main() loop
// (BscanFFT.cpp:1123-1240 / BscanFFTsim.cpp:842-955) plus its one-time set-up
// (BscanFFT.cpp:544-698, 936-944).  No CPU compute path exists here: every
// fdoct_process* call runs the gfx950 kernels or fails.
#include "fdoct_ctx.h"

using namespace fdoct_impl;

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

int fdoct_build_colormap_jet(unsigned char* bgr256) {
  if (!bgr256) return FDOCT_ERR_INVALID;
  build_opencv_jet(bgr256);
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
  void* ptrs[] = {h->d_ib, h->d_ib2d, h->d_ib2d_f, h->d_il, h->d_il2d, h->d_il2d_f, h->d_il_p, h->d_il16, h->d_il16_2d, h->d_yp, h->d_yd, h->d_yp_lo, h->d_yd_lo, h->d_win, h->d_g, h->d_gidx, h->d_tw, h->d_utw,
                  h->d_phase, h->d_minmax, h->ws_in, h->ws_f32, h->ws_f32_lo, h->ws_mov_lo, h->ws_out0, h->ws_out1, h->ws_tr, h->ws_ylin,
                  h->d_win_g, h->d_win_lo_g, h->d_g_g, h->d_idx_g, h->d_wave_gidx, h->d_wave_tw, h->d_blu_chirp, h->d_blu_bhat, h->d_twg_blu, h->d_twg_n, h->d_twg_nh, h->d_twg_w, h->d_twg_mw, h->d_twg_wh, h->d_twg_mwh, h->ws_mov, h->ws_front, h->ws_med, h->ws_raw, h->ws_sim,
                  h->d_lut, h->d_disp_part, h->ws_disp_in, h->ws_disp_in2, h->ws_disp_out, h->d_gen_tickets,
                  h->gzf.d_tw, h->gzf.d_chirp, h->gzf.d_bhat, h->gzi.d_tw, h->gzi.d_chirp, h->gzi.d_bhat};
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
    for (void* p : {h->pin_in[b], (void*)h->pin_mag[b], (void*)h->pin_db[b]})
      if (p) (void)hipHostFree(p);
  }
  delete h->copy_pool;
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
// skip.)  A no-op for S = 1 and for the main variant.  Host batches worth chunking do not come here: fdoct_process hands the
// pipeline a frame stride instead (no batch-sized buffer).
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
                                  int frames_per_chunk, size_t frame_stride);

// frame_stride: bytes from one frame of the batch to the next in the caller's memory -- rows_per_frame * src_pitch for a packed
// batch; the sim variant with averages = S reads every S-th frame (the last of each group), S times that.
static int process_pipelined(fdoct_ctx* h, const unsigned char* frames, fdoct_dtype dtype, int nframes, size_t src_pitch,
                             size_t row_bytes, long long rows_per_frame, float* out_bscan, float* out_db, fdoct_layout layout,
                             int frames_per_chunk, size_t frame_stride) {
  const int rc = process_pipelined_impl(h, frames, dtype, nframes, src_pitch, row_bytes, rows_per_frame, out_bscan, out_db, layout,
                                        frames_per_chunk, frame_stride);
  if (rc != FDOCT_OK) {  // leave nothing in flight that still points at the caller's buffers or the chunk slots
    if (h->s_in) (void)hipStreamSynchronize(h->s_in);
    (void)hipStreamSynchronize(h->stream);
    if (h->s_out) (void)hipStreamSynchronize(h->s_out);
  }
  return rc;
}

static bool host_staging_enabled(const fdoct_ctx* h) {
  bool on = h->host_staging != 0;
  if (const char* e = std::getenv("FDOCT_HOST_STAGING")) on = on && std::atoi(e) != 0;
  return on;
}

// Is this host pointer pinned (hipHostMalloc / hipHostRegister), i.e. can a DMA engine reach it without the runtime's bounce
// buffer?  Pageable memory is "unregistered" to the runtime (an error from hipPointerGetAttributes on older runtimes).
static bool host_pointer_is_pinned(const void* p) {
  hipPointerAttribute_t a;
  std::memset(&a, 0, sizeof a);
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// The copy threads of a handle (fdoct_hostcopy.h), started with the first batch that needs them.  An explicit count
// (fdoct_set_host_staging(h, n) or FDOCT_HOST_COPY_THREADS) is taken as given.  Left to the library: half of the hardware
// threads this process may use, eight at most, the caller's thread among them -- and NO staging below four, because one or two threads
// copy more slowly than the runtime's own bounce path (MI355X host, 64 frames of 2048 x 1000 u16 per call, result array
// reused: 3.2-3.4 / 5.9-6.0 / 8.4-8.7 / 8.4-9.4 M A-scans/s with 1 / 2 / 4 / 8 threads against 6.0-6.4 M from the runtime and 10.5 M from pinned
// buffers; profiles/r06_pcie_rate.txt).
static int copy_thread_count(const fdoct_ctx* h) {  // 0: pageable buffers are not staged
  if (!host_staging_enabled(h)) return 0;
  int n = h->host_staging > 0 ? h->host_staging : 0;
  if (!n)
    if (const char* e = std::getenv("FDOCT_HOST_COPY_THREADS")) n = std::atoi(e);
  if (n <= 0) {
    n = std::min(8, (int)std::thread::hardware_concurrency() / 2);
    if (n < 4) return 0;
  }
  return std::min(n, 64);
}

static fdoct_impl::HostCopyPool* copy_pool(fdoct_ctx* h) {
  const int n = copy_thread_count(h);
  if (!n) return nullptr;
  if (!h->copy_pool) h->copy_pool = new (std::nothrow) fdoct_impl::HostCopyPool(n);
  return h->copy_pool;
}

static int process_pipelined_impl(fdoct_ctx* h, const unsigned char* frames, fdoct_dtype dtype, int nframes, size_t src_pitch,
                                  size_t row_bytes, long long rows_per_frame, float* out_bscan, float* out_db, fdoct_layout layout,
                                  int frames_per_chunk, size_t frame_stride) {
  int rc;
  const bool packed_batch = frame_stride == (size_t)rows_per_frame * src_pitch;  // one 2-D copy moves a whole chunk
  // Pageable buffers go through the handle's pinned slots (fdoct_hostcopy.h); pinned ones are the DMA engines' to read and write.
  fdoct_impl::HostCopyPool* pool = copy_pool(h);
  bool stage_in = pool && !host_pointer_is_pinned(frames);
  bool stage_mag = pool && out_bscan && !host_pointer_is_pinned(out_bscan);
  bool stage_db = pool && out_db && !host_pointer_is_pinned(out_db);
  struct Landed {  // a chunk whose downloads go to (or sit in) the pinned slots and still have to reach the caller's buffers
    size_t o0 = 0, elems = 0;
    bool live = false;
  } landed[2];
  auto hand_over = [&](int b) -> int {
    if (!landed[b].live) return FDOCT_OK;
    HIP_TRY(h, hipEventSynchronize(h->pe_out[b]));
    if (stage_mag) pool->copy(out_bscan + landed[b].o0, h->pin_mag[b], landed[b].elems * 4);
    if (stage_db) pool->copy(out_db + landed[b].o0, h->pin_db[b], landed[b].elems * 4);
    landed[b].live = false;
    return FDOCT_OK;
  };
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
  {
    // The pinned slots, sized for the first (the largest) chunk, before anything is enqueued: a host that will not pin that
    // much memory (a locked-memory limit) gets the runtime's own bounce copies for that buffer, not an error.
    const int nf0 = std::min(frames_per_chunk, nframes);
    const size_t in0 = packed * (size_t)nf0 * (size_t)rows_per_frame, out0 = (size_t)(nf0 / h->A) * out_per_group * 4;
    const std::string err_before = h->err;
    for (int b = 0; b < 2; b++) {
      if (stage_in && host_reserve(h, &h->pin_in[b], &h->pin_in_cap[b], in0)) stage_in = false;
      if (stage_mag && host_reserve(h, &h->pin_mag[b], &h->pin_mag_cap[b], out0)) stage_mag = false;
      if (stage_db && host_reserve(h, &h->pin_db[b], &h->pin_db_cap[b], out0)) stage_db = false;
    }
    h->err = err_before;
  }
  uint64_t sum_in = 0, sum_out = 0;  // fdoct_get_timing reports the whole batch, not the last chunk
  for (int f0 = 0, c = 0; f0 < nframes; f0 += frames_per_chunk, c++) {
    const int b = c & 1;
    const int nf = std::min(frames_per_chunk, nframes - f0);
    const size_t in_rows = (size_t)nf * rows_per_frame;
    const size_t out_elems = (size_t)(nf / h->A) * out_per_group;
    if ((rc = dev_reserve(h, &h->pl_in[b], &h->pl_in_cap[b], packed * in_rows))) return rc;
    if (out_bscan && (rc = dev_reserve(h, &h->pl_mag[b], &h->pl_mag_cap[b], out_elems * 4))) return rc;
    if (out_db && (rc = dev_reserve(h, &h->pl_db[b], &h->pl_db_cap[b], out_elems * 4))) return rc;
    const unsigned char* src = frames + (size_t)f0 * frame_stride;
    // a packed batch moves as one 2-D copy of the chunk's rows, a strided one frame by frame
    const int pieces = packed_batch ? 1 : nf;
    const size_t piece_rows = packed_batch ? in_rows : (size_t)rows_per_frame;
    if (stage_in) {
      if (c >= 2) HIP_TRY(h, hipEventSynchronize(h->pe_in[b]));           // chunk c-2's upload has left this pinned slot
      for (int q = 0; q < pieces; q++)
        pool->copy2d(static_cast<unsigned char*>(h->pin_in[b]) + (size_t)q * piece_rows * packed, packed, src + (size_t)q * frame_stride, src_pitch,
                     row_bytes, piece_rows);
    }
    if (c >= 2) HIP_TRY(h, hipStreamWaitEvent(h->s_in, h->pe_k[b], 0));   // chunk c-2 has consumed this input slot
    if (stage_in) {
      HIP_TRY(h, hipMemcpyAsync(h->pl_in[b], h->pin_in[b], packed * in_rows, hipMemcpyHostToDevice, h->s_in));
    } else {
      for (int q = 0; q < pieces; q++)
        HIP_TRY(h, hipMemcpy2DAsync(static_cast<unsigned char*>(h->pl_in[b]) + (size_t)q * piece_rows * packed, packed, src + (size_t)q * frame_stride,
                                    src_pitch, row_bytes, piece_rows, hipMemcpyHostToDevice, h->s_in));
    }
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
    // chunk c-2's images leave the pinned slots (while chunk c uploads and computes) before chunk c's download may land there
    if ((rc = hand_over(b))) return rc;
    if (out_bscan) HIP_TRY(h, hipMemcpyAsync(stage_mag ? h->pin_mag[b] : out_bscan + o0, h->pl_mag[b], out_elems * 4, hipMemcpyDeviceToHost, h->s_out));
    if (out_db) HIP_TRY(h, hipMemcpyAsync(stage_db ? h->pin_db[b] : out_db + o0, h->pl_db[b], out_elems * 4, hipMemcpyDeviceToHost, h->s_out));
    HIP_TRY(h, hipEventRecord(h->pe_out[b], h->s_out));
    landed[b].o0 = o0;
    landed[b].elems = out_elems;
    landed[b].live = stage_mag || stage_db;
  }
  HIP_TRY(h, hipStreamSynchronize(h->s_out));
  HIP_TRY(h, hipStreamSynchronize(s_k));
  for (int b = 0; b < 2; b++)
    if ((rc = hand_over(b))) return rc;
  h->timing.bytes_in = sum_in;
  h->timing.bytes_out = sum_out;
  return FDOCT_OK;
}

int fdoct_get_host_staging(fdoct_handle h) {
  if (!h) return FDOCT_ERR_INVALID;
  return copy_thread_count(h);
}

int fdoct_set_host_staging(fdoct_handle h, int threads) {
  if (!h) return FDOCT_ERR_INVALID;
  const int want = threads < 0 ? -1 : std::min(threads, 64);
  if (want != h->host_staging) {  // the pool is sized when it starts: a new count means a new pool (no batch is in flight here)
    delete h->copy_pool;
    h->copy_pool = nullptr;
  }
  h->host_staging = want;
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
  const size_t row_samples = (size_t)h->W * h->fe_binx;
  size_t d_pitch = pitch_bytes ? pitch_bytes : es * row_samples;
  const long long rows_per_frame = (long long)h->H * h->fe_biny;      // raw camera rows when a front end is set
  size_t frame_stride = (size_t)rows_per_frame * d_pitch;
  DEVICE_SCOPE(h);
  // Host buffers on both sides: the batch is cut into chunks of whole averaging groups and pipelined (process_pipelined).  Chunk
  // size and the batch that is worth chunking, from tools/pcie_chunk.py (profiles/r06_pcie_chunk.txt, M A-scans/s on C2's frames;
  // round 5 had 32 MB chunks and two of them as the threshold).  Pinned buffers: 16 MB chunks for batches of ~100 MB and more,
  // 8 MB below, two chunks are worth it (15 / 31 / 62 / 125 MB in: 7.3 / 8.6 / 9.5 / 10.1 against 6.3 / 6.6 / 8.3 / 9.6).
  // Pageable buffers (staged by the copy threads): 16 MB chunks, four of them or the single shot (62 / 125 / 250 MB in:
  // 6.9-7.1 / 8.3 / 8.5-9.1 against 6.2-6.5 / 6.6-7.4 / 8.5-8.7; 8 MB chunks lose to the single shot at 31 MB).
  const size_t frame_bytes = es * row_samples * (size_t)rows_per_frame;
  const int S0 = h->sim_group > 1 ? h->sim_group : 1;
  const bool both_host = space == FDOCT_MEM_HOST && out_space == FDOCT_MEM_HOST;
  const bool pageable = both_host && (!host_pointer_is_pinned(frames) || (out_bscan && !host_pointer_is_pinned(out_bscan)) ||
                                      (out_db && !host_pointer_is_pinned(out_db)));
  size_t chunk_bytes = pageable || frame_bytes * (size_t)(nframes / S0) >= ((size_t)96 << 20) ? (size_t)16 << 20 : (size_t)8 << 20;
  if (const char* e = std::getenv("FDOCT_HOST_CHUNK_MB"))  // tuning aid (tools/pcie_chunk.py)
    if (std::atoll(e) > 0) chunk_bytes = (size_t)std::atoll(e) << 20;
  long long fpc = (long long)(chunk_bytes / (frame_bytes ? frame_bytes : 1));
  fpc = std::max<long long>(fpc / h->A, 1) * h->A;
  const int min_chunks = pageable ? 4 : 2;
  const int S = h->sim_group;
  if (S > 1 && space == FDOCT_MEM_HOST && out_space == FDOCT_MEM_HOST && nframes % S == 0 && nframes / S >= min_chunks * fpc) {
    // sim variant, averages = S, a batch worth pipelining: the chunks read the last frame of every group where it lies
    // (no batch-sized gather buffer, sim_last_frames' fallback below)
    frames = static_cast<const unsigned char*>(frames) + (size_t)(S - 1) * frame_stride;
    frame_stride *= (size_t)S;
    nframes /= S;
  } else if (int rc0 = sim_last_frames(h, &frames, &space, dtype, &nframes, pitch_bytes)) {
    return rc0;
  }
  if (nframes % h->A) return fail(h, FDOCT_ERR_INVALID, "nframes must be a multiple of averages");
  int rc;
  const long long in_rows = (long long)nframes * rows_per_frame;
  const size_t out_elems = (size_t)(nframes / h->A) * h->H * h->D;
  const void* d_frames = frames;
  if (space == FDOCT_MEM_HOST && out_space == FDOCT_MEM_HOST) {
    if (nframes >= min_chunks * fpc) {
      const auto t0 = std::chrono::steady_clock::now();
      rc = process_pipelined(h, static_cast<const unsigned char*>(frames), dtype, nframes, d_pitch, es * row_samples,
                             rows_per_frame, out_bscan, out_db, layout, (int)fpc, frame_stride);
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
  if (plan_id < -3) return fail(h, FDOCT_ERR_INVALID, "plan id: -1 automatic, -2 the workgroup-per-row kernel, -3 the long-row path");
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
  {
    // (FDOCT_JIT_CHECK_OPT: the kernel's option mask -- fdoct_wave.h, FDOCT_WAVE_OPT_* -- for build checks of the optional stages)
    const char* eo = std::getenv("FDOCT_JIT_CHECK_OPT");
    n = wave_jit_compile_only(width, multiplier, numfftpoints, kdt, numdisplaypoints, eo ? std::atoi(eo) : 0, gcn_arch, &reason);
  }
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
  c->host_staging = h->host_staging;  // (the setting: the clone starts its own copy threads)
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
      try {
        blob.resize(used);
      } catch (...) {   // (no exception crosses the C ABI: the root takes part in the size broadcast with 0, every rank returns an error)
        blob.clear();
        root_rc = fail(h, FDOCT_ERR_NOMEM, "fdoct_broadcast_state_rccl: no host memory for the state blob on the root");
      }
    }
    if (!root_rc) root_rc = fdoct_export_state(h, blob.data(), blob.size(), &used);
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
    // (this rank cannot know how many chunk broadcasts follow, so it cannot stay in step with the others: include/fdoct.h
    // documents this exit next to the pre-collective ones)
    return fail(h, FDOCT_ERR_DEVICE, "fdoct_broadcast_state_rccl: the broadcast blob size could not be read back or is implausible (this rank leaves "
                                     "before the chunk broadcasts: abort the communicator)");
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
