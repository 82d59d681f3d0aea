/*
 * fdoct.h -- C ABI of the MI355X-native FD-OCT A-scan reconstruction path.
 *
 * Drop-in boundary for the per-frame processing block of hn-88/FDOCT.  The
 * reference has no plugin/FFI interface: the block is inlined in main() at
 * BscanFFT.cpp:1123-1240 (BscanFFTsim.cpp:842-955) and its "interface" is the
 * set of live locals there.  Each entry point below names the reference lines
 * it replaces.  All citations are into the reference tree; "main" =
 * BscanFFT.cpp, "sim" = BscanFFTsim.cpp, "dark" = BscanDark.cpp.
 *
 * Conventions (mirroring the reference host code, main:729-925,1991-1993):
 *   - every call returns int, 0 = success, negative = error; nothing throws or
 *     exits across this boundary; fdoct_last_error() gives the text;
 *   - the caller owns every buffer it passes; setters COPY; the library never
 *     keeps a caller pointer after the call returns;
 *   - a handle is used from one thread at a time (the reference loop is single
 *     threaded, main:946); distinct handles (one per GPU / stream) are
 *     independent;
 *   - there is no CPU fallback: without a HIP device the create call fails;
 *   - every call runs on the handle's device and restores the calling thread's
 *     current HIP device before it returns.
 */
#ifndef FDOCT_H
#define FDOCT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDOCT_VERSION_MAJOR 0
#define FDOCT_VERSION_MINOR 4   /* 0.4: fdoct_set_host_staging, fdoct_get_host_staging (additions only) */

typedef struct fdoct_ctx* fdoct_handle;

/* error codes */
enum {
  FDOCT_OK = 0,
  FDOCT_ERR_INVALID = -1,     /* bad argument */
  FDOCT_ERR_UNSUPPORTED = -2, /* valid in the reference, not built here yet */
  FDOCT_ERR_DEVICE = -3,      /* HIP error (no device, launch failure, ...) */
  FDOCT_ERR_NOMEM = -4,
  FDOCT_ERR_STATE = -5        /* required state (e.g. background) missing */
};

/* sample types of caller buffers */
typedef enum {
  FDOCT_U8 = 0,  /* 8-bit camera frame (CV_8U `opm`, main:958) */
  FDOCT_U16 = 1, /* 16-bit camera frame (CV_16U) */
  FDOCT_F32 = 2,
  FDOCT_F64 = 3  /* `data_y` after convertTo(CV_64F), main:987 */
} fdoct_dtype;

/* where a caller pointer lives */
typedef enum { FDOCT_MEM_HOST = 0, FDOCT_MEM_DEVICE = 1 } fdoct_memspace;

/* output layout */
typedef enum {
  FDOCT_LAYOUT_ROWMAJOR_HxD = 0,  /* out[g][row][depth]  (fast path) */
  FDOCT_LAYOUT_TRANSPOSED_DxH = 1 /* out[g][depth][row] = the reference's `bscan`, main:1220 */
} fdoct_layout;

/* which program's constants to follow */
typedef enum {
  FDOCT_VARIANT_MAIN = 0, /* BscanFFT.cpp: accumulate + /averages, eps 1e-5 (main:1197-1222) */
  FDOCT_VARIANT_SIM = 1   /* BscanFFTsim.cpp: eps 1e-6 (sim:949); whole-frame normalise always (sim:845); no averaging --
                           * sim:936-947 copies each frame's magnitudes and emits the last one: with averages = A the last frame
                           * of every group of A is processed and emitted, undivided (the frame on which the reference's
                           * loop emits, its (A + 1)-th, is computed and dropped there: it is the caller's to skip) */
} fdoct_variant;

/* The locals of main() that the block reads (main:395-484 ini values,
 * main:544-545 derived sizes).  Zero-initialise, set struct_size, fill. */
typedef struct {
  uint32_t struct_size;
  int32_t width;                       /* W = opw: samples per row after binning (main:544) */
  int32_t height;                      /* H = oph: rows (A-scans) per frame (main:545) */
  int32_t numfftpoints;                /* N (ini; main:471) */
  int32_t numdisplaypoints;            /* D <= N */
  int32_t increasefftpointsmultiplier; /* M: zero-pad spectral upsampling (main:180-245, 1146-1147) */
  int32_t averages;                    /* A = averagestoggle, frames averaged per B-scan (main:481) */
  int32_t rowwisenormalize;            /* main:1126 */
  int32_t donotnormalize;              /* main:1128; ignored (treated as 0) for FDOCT_VARIANT_SIM */
  int32_t movavgn;                     /* smoothmovavg taps (main:990-991) */
  int32_t variant;                     /* fdoct_variant */
  int32_t dc_mask;                     /* 1 = copy dB depth-bin 4 over bins 0,1 (main:1237-1238) */
  int32_t device;                      /* HIP device ordinal */
  double lambdamin, lambdamax;         /* main:381-382,479-480 */
} fdoct_config;

typedef struct {
  double last_process_ms;  /* device time of the last fdoct_process* call (HIP events) */
  double last_kernel_ms;   /* device time of the fused kernel (staged mode: both stage kernels) in that call */
  double resample_stage_ms; /* staged mode only: the resample-stage kernel; 0 otherwise */
  double fft_stage_ms;      /* staged mode only: the FFT-stage kernel; 0 otherwise */
  uint64_t ascans;         /* input A-scans processed by that call */
  uint64_t bytes_in, bytes_out; /* algorithmic bytes of that call */
} fdoct_timing;

const char* fdoct_version(void);

/* Which kernels run: power-of-two numfftpoints with M = 1, width % 8 == 0 and D <= N/2 use the
 * specialised fused kernels (fdoct_kernels.hip); the acquisition shapes of the shipped ini files run one
 * wave per A-scan (fdoct_wave.hip) -- instantiations the library carries for those shapes and their neighbours,
 * and, for any other geometry whose lengths factor into 2, 3 and 5 (or with a pi / dark frame, the band-pass, a
 * normalisation, the dispersion phase or a display beyond numfftpoints / 2 set), an instantiation compiled for the handle at
 * run time (fdoct_set_jit below); every other configuration the reference accepts (any numfftpoints -- lengths with prime
 * factors above 5 run as Bluestein's algorithm --, any width, unaligned device frames) runs on the any-configuration
 * kernel (fdoct_generic.hip), rows beyond the LDS on the long-row path (fdoct_big.hip) -- slower, same
 * arithmetic.  fdoct_last_kernel tells which one a call took; what still returns FDOCT_ERR_UNSUPPORTED is
 * listed in DESIGN.md 7. */

/* Replaces the one-time setup main:544-698 + 936-944: allocates device state,
 * builds the k tables (A0) and the Bartlett-Hann window (A1) on the host in
 * double precision and uploads them.  A background must still be set. */
int fdoct_create(const fdoct_config* cfg, fdoct_handle* out);
int fdoct_destroy(fdoct_handle h);
const char* fdoct_last_error(fdoct_handle h); /* h may be NULL: last create error */

/* Use an existing HIP stream (hipStream_t) for all work of this handle; NULL =
 * the handle's own stream. */
int fdoct_set_stream(fdoct_handle h, void* hip_stream);

/* data_yb, the 'b' key (main:1000-1075, sim:803-813): rows = 1 (one spectrum
 * for every row) or H (full frame).  pitch_bytes = 0 means tightly packed.
 * Host pointer.  Zeros divide to 0 (OpenCV 3.x semantics). */
int fdoct_set_background(fdoct_handle h, const void* data, fdoct_dtype dtype, int rows, size_t pitch_bytes);
/* data_yp, the 'p' key (main:1077-1099); NULL clears (zeros, main:563). */
int fdoct_set_pi_frame(fdoct_handle h, const void* data, fdoct_dtype dtype, int rows, size_t pitch_bytes);
/* data_yd (dark:1269); NULL clears. */
int fdoct_set_dark(fdoct_handle h, const void* data, fdoct_dtype dtype, int rows, size_t pitch_bytes);
/* barthannwin (main:936-944); NULL restores the built-in window.  n must equal `width` whatever
 * increasefftpointsmultiplier is: the window multiplies the row before the zero-pad upsampling
 * (main:1142 precedes main:1146). */
int fdoct_set_window(fdoct_handle h, const double* win, int n);
/* nearestkindex / fractionalk (main:673-698) supplied by the caller instead of
 * derived from lambdamin/lambdamax. */
int fdoct_set_resample_table(fdoct_handle h, const int32_t* nearestkindex, const double* fractionalk, int n);
int fdoct_set_lambda_range(fdoct_handle h, double lambdamin, double lambdamax);
/* Extension (not in the reference C++; Octave prototype wangOCTrec4.m:130-169):
 * N (cos,sin) pairs multiplied into data_ylin before the IDFT.  NULL = off. */
int fdoct_set_dispersion_phase(fdoct_handle h, const float* cos_sin_pairs, int n);

/* Host-only helpers (no device needed): the reference's one-time tables.
 * fdoct_build_resample_table = main:615-698 (A0); fdoct_build_window =
 * main:936-944 (A1). */
int fdoct_build_resample_table(int width, int multiplier, int numfftpoints, double lambdamin, double lambdamax,
                               int32_t* nearestkindex, double* fractionalk);
int fdoct_build_window(int width, double* win);

/* Read back the tables the handle uses (for parity checks against main:615-698). */
int fdoct_get_resample_table(fdoct_handle h, int32_t* nearestkindex, double* fractionalk, int n);
int fdoct_get_window(fdoct_handle h, double* win, int n);

/* averagestoggle (main:481; the 'a' / 'A' style run-time changes of the reference's UI): frames averaged per output
 * B-scan from the next fdoct_process* call on.  nframes of a call must be a multiple of it. */
int fdoct_set_averages(fdoct_handle h, int averages);

/* BscanDark.cpp's `bandpassfilter` (BscanDark.cpp:218-236): inside the zero-pad upsampling the shifted row spectrum
 * is blanked except for a band next to DC (bins 3 <= k < floor(width/10) survive).  Only acts when
 * increasefftpointsmultiplier > 1, exactly as in the reference, where the filter sits inside zeropadrowwise.
 * What is displayed with the filter on is the little the window leaks into those few bins, so every float rounding in front
 * of the blanking counts at the size of the whole row (the reference's own float chain, main:209-211, sits up to tens of
 * tolerances from its mathematics on such rows): with the filter on, the library forms the row and evaluates the kept bins
 * in DOUBLE (rows that fit a compute unit's LDS; 0.3-0.8 of the rate without the filter). */
int fdoct_set_bandpass(fdoct_handle h, int on);

/* The frame-source tail that sits right before the block: cv::medianBlur(mraw, m,
 * mediann) when mediann > 0 (main:953-956; 3, 5 or 7, replicate border) and the software binning
 * cv::resize(m, opm, 1/binx, 1/biny, INTER_AREA) (main:958; BscanFFTspinjnt.cpp:1553 for binx != biny).
 * With a front end set, fdoct_process* take RAW camera frames of (height*biny) rows x (width*binx)
 * samples (u8/u16) and bin them on the GPU; 0/1/1 switches it off.  Integer-factor INTER_AREA rounds to
 * the sample type: (s+2)>>2 for 2x2, round-half-even of s/(binx*biny) otherwise. */
int fdoct_set_frontend(fdoct_handle h, int mediann, int binx, int biny);
/* The same front end on its own: raw frames in, binned frames out (same dtype, tightly packed), host
 * pointers.  raw_w / raw_h are the camera frame's size; they must be multiples of binx / biny. */
int fdoct_frontend(fdoct_handle h, const void* raw, fdoct_dtype dtype, int nframes, int raw_w, int raw_h,
                   size_t pitch_bytes, int mediann, int binx, int biny, void* out);

/* The display post-chain that consumes bscandb, main:1242-1255 and 1284:
 *   bscandisp = max(bscandb, bscanthreshold); if clampupper, element (5,5) <- 50 dB; min-max normalise to
 *   [0,1]; x255 -> u8 (round half to even, saturate); optional colour look-up (applyColorMap, main:1284).
 * bscandb: nbscans B-scans of rows x cols floats each (the D x H layout the reference displays).  The
 * arithmetic is done in double on the f32 input, as the reference does on its CV_64F Mats.
 * out_gray (nbscans*rows*cols bytes) and out_bgr (3x that, B,G,R order like cv::Mat CV_8UC3) may each be
 * NULL.  in_mem / out_mem say where the pointers live. */
int fdoct_display(fdoct_handle h, const float* bscandb, fdoct_memspace in_mem, int nbscans, int rows, int cols,
                  double bscanthreshold, int clampupper, unsigned char* out_gray, unsigned char* out_bgr,
                  fdoct_memspace out_mem);
/* The 256-entry B,G,R table fdoct_display applies (768 bytes, copied); NULL restores the built-in one.  The built-in table is
 * COLORMAP_JET (main:1284) BUILT THE WAY OPENCV BUILDS IT -- Octave's jet(256) as float literals, interp1 over
 * linspace(0.f, 1.f, 256) in float, convertTo(CV_8U, 255.): imgproc/src/colormap.cpp -- operation by operation, because every
 * table value lies half-way between two bytes and the float roundings decide each entry (fdoct_host.cpp::build_opencv_jet).
 * fdoct_build_colormap_jet returns that table without a handle (no GPU needed).  OpenCV itself is not available where this
 * library is built, so the table is pinned by the recipe and the published end points until a maintainer runs
 * `make -C oracle opencv-golden`; a caller who links OpenCV may still pass applyColorMap's own table here. */
int fdoct_build_colormap_jet(unsigned char* bgr256);
int fdoct_set_colormap(fdoct_handle h, const unsigned char* bgr256);
int fdoct_get_colormap(fdoct_handle h, unsigned char* bgr256);
/* J0 lock-in (main:1225-1230, 1260-1261): out_db = 20*ln(max(bscan - jscan, 0) + 0.001)/2.303 for nbscans
 * linear B-scans of `count` floats against ONE saved jscan of `count` floats ('j' key, main:1292-1296).
 * Feed out_db to fdoct_display for the "Bscan subtracted" image.  All pointers in `mem`. */
int fdoct_lockin_db(fdoct_handle h, const float* bscan, const float* jscan, fdoct_memspace mem, int nbscans,
                    size_t count, float* out_db);

/* Replaces main:1123-1240 for a batch of frames.
 *   frames   nframes*H rows of W samples, row pitch pitch_bytes (0 = packed)
 *   nframes  multiple of `averages`; G = nframes/averages outputs
 *            (FDOCT_VARIANT_SIM: of every group of `averages` frames the LAST one's magnitudes are what sim:936-947 emits,
 *            copied, undivided; the other frames of a group are not even read)
 *   out_bscan  G*H*D floats: bscan = mean over the group + epsilon (main:1220-1222), or NULL
 *   out_db     G*H*D floats: 20*ln(bscan)/2.303 with the DC mask (main:1235-1238), or NULL
 * Synchronous: results are complete when the call returns. */
int fdoct_process(fdoct_handle h, const void* frames, fdoct_dtype dtype, fdoct_memspace space, int nframes,
                  size_t pitch_bytes, float* out_bscan, float* out_db, fdoct_memspace out_space,
                  fdoct_layout layout);
/* Same, device pointers only, enqueued on the handle's stream without a host
 * sync (the batch / benchmark path).
 * Transposed layout: the chain's own D x H store (FDOCT_KERNEL_FUSED_TRANSPOSED) has a bounded wait inside the kernel; should a
 * wave ever give up (a broken hand-over protocol: never seen), one word of pinned host memory is raised and the NEXT entry point
 * that looks -- fdoct_synchronize, fdoct_get_timing, fdoct_process, or the next fdoct_process_async, which then returns
 * FDOCT_ERR_DEVICE WITHOUT enqueueing anything -- reports it once and clears it: the error refers to the EARLIER transposed-layout
 * calls since the last check, not to the call that returns it.
 * FDOCT_VARIANT_SIM with averages > 1: the last frame of every group is first GATHERED into a packed buffer the handle owns
 * (one strided device-to-device copy on the stream: one more read and write of those frames in HBM, G * frame bytes of device
 * memory), and the FIRST such call of a given batch size allocates or grows that buffer -- hipMalloc / hipFree, which
 * synchronise the device.  The sim variant is the reference's file-driven test harness, not an acquisition loop; the copy is
 * the price of keeping a group stride out of every kernel's row arithmetic.  With host frames and host results
 * (fdoct_process) a batch worth chunking needs no gather: the three-stream pipeline's uploads read every S-th frame where
 * it lies; a smaller batch is one strided upload of its last frames. */
int fdoct_process_async(fdoct_handle h, const void* d_frames, fdoct_dtype dtype, int nframes, size_t pitch_bytes,
                        float* d_out_bscan, float* d_out_db, fdoct_layout layout);
int fdoct_synchronize(fdoct_handle h);

/* Pinned host memory for frames / results handed to fdoct_process with FDOCT_MEM_HOST.  With host buffers on both
 * sides fdoct_process cuts a large batch into chunks and overlaps upload, kernels and download on three streams;
 * the DMA engines read and write pinned memory directly (both PCIe directions at once).  NULL on failure. */
void* fdoct_host_alloc(size_t bytes);
void fdoct_host_free(void* p);

/* Pageable buffers -- what cv::Mat owns at sim:842, so what the patch of INTEGRATION.md 1 hands over -- go through
 * pinned staging slots the handle owns (two chunks of each image and of the input, 8-16 MB each): a few host threads copy
 * chunk c + 1 in and chunk c - 1 out while chunk c is on the device, instead of the runtime's one-direction-at-a-time
 * bounce copies on the calling thread.  Decided per buffer and per call (a pinned buffer is never staged).
 *   threads < 0   the default: half of the host's hardware threads, eight at most, the caller's among them -- and no
 *                 staging on hosts where that is fewer than four (one or two threads copy more slowly than the runtime's
 *                 bounce path does).  Environment: FDOCT_HOST_COPY_THREADS sets the count, FDOCT_HOST_STAGING=0 turns
 *                 staging off.
 *   threads = 0   no staging: pageable buffers are handed to the runtime as they are (rounds 1-5)
 *   threads > 0   staging with that many copy threads (1 = the calling thread alone)
 * Takes effect with the next batch; the threads are started by the first batch that stages and end with the handle.
 * Measured (MI355X host, 64 frames of 2048 x 1000 u16 per call, dB image out, buffers reused from call to call; two boxes):
 * 6.0-6.4 M A-scans/s through the runtime's bounce copies, 3.2-3.4 / 5.9-6.0 / 8.4-8.7 / 8.4-9.4 M with one / two / four /
 * eight copy threads, 10.5-10.6 M from pinned buffers (profiles/r06_pcie_rate.txt; by batch size: r06_pcie_chunk.txt).  A
 * result buffer allocated afresh for every call pays its page faults first (1.4-1.6 M unstaged, 2.0-3.7 M staged): keep
 * the cv::Mat.  Single-chunk calls (one frame per call, the reference's own call
 * shape) are not affected. */
int fdoct_set_host_staging(fdoct_handle h, int threads);
/* The number of copy threads a pageable batch would be staged with under the current setting (0: not staged). */
int fdoct_get_host_staging(fdoct_handle h);

int fdoct_get_timing(fdoct_handle h, fdoct_timing* t);
/* fdoct_process always brackets its work with device events; fdoct_process_async records them only after
 * fdoct_set_timing(h, 1) -- each record costs a few microseconds of stream time between kernels, which a
 * back-to-back batch loop does not want to pay.  Without them fdoct_get_timing reports 0 ms. */
int fdoct_set_timing(fdoct_handle h, int on);

/* main:1132 divides by data_yb in double; the kernels multiply by 1/data_yb held as the unevaluated sum of two floats
 * (high word + low word = the quotient to 2^-48), so that nothing is rounded at the size of the DC level and the north-star
 * tolerance holds for fringes of any depth of modulation.  (A single f32 reciprocal leaves a fixed per-column pattern of up
 * to 6e-8 of the DC level per sample, up to 4e-6 of it per depth bin: more than the tolerance once the fringes are weaker
 * than ~1 % of the DC level -- 8 x the tolerance at 0.1 %.)  Every kernel does this, and since round 5 so does the fast path of
 * the fused kernel BY DEFAULT (there the second word is applied as c0 * (low / high) with the pattern held as half floats:
 * one more instruction per sample and a 2 W-byte plane; a full-frame background brings its pattern along
 * with the prefetched row).  fdoct_set_precise_division(h, 0) -- or FDOCT_PRECISE_DIVISION=0 in the environment at
 * fdoct_create -- is the OPT-OUT for callers who know their fringes exceed ~1 % of the DC level: the fast path then
 * multiplies by the high word alone (cost of the default and what it buys: INTEGRATION.md 4). */
int fdoct_set_precise_division(fdoct_handle h, int on);

/* Tuning knobs of the fused kernel (0 = automatic). */
int fdoct_set_launch(fdoct_handle h, int threads_per_block, int blocks);
/* Choose the compiled FFT plan (-1 = automatic, -2 = force the any-configuration kernel, -3 = force the long-row path: rows in
 * HBM, the route of geometries whose transforms no compute unit's LDS holds) and optionally force the general
 * (predicated) kernel where the fast-path one would apply.  Results do not depend
 * on either; they exist for tuning and for testing both kernels. */
int fdoct_set_plan(fdoct_handle h, int plan_id, int force_general_kernel);
/* Staged mode (results identical to the default fused chain -- bit for bit, except 4096-sample rows with averages > 1 and both
 * words of 1/background, where the two differ by < 0.1 of the parity tolerance: the low words travel as half floats in one kernel
 * and as floats in the other -- 3x the HBM traffic): run the path as two
 * kernels, "resample" (samples -> k-linear rows in a library-owned HBM buffer) and "FFT" (rows ->
 * magnitudes/dB), so that each stage can be timed against the HBM roofline on its own.  Built for the
 * plain acquisition configuration (u16 frames, 1-row background, no normalisation, any averaging: the
 * buffer holds one row per INPUT A-scan and the FFT stage averages); other configurations return
 * FDOCT_ERR_UNSUPPORTED while it is on. */
int fdoct_set_staged(fdoct_handle h, int on);
/* data_ylin of the last staged run (fdoct_set_staged on, then fdoct_process*): rows row0 .. row0+nrows-1 of the
 * k-linear spectra the resample stage left in HBM, as the reference holds them -- numfftpoints doubles per A-scan, the
 * window's internal 1/2 undone, columns 0 and numfftpoints-1 zero (the reference never writes them).  This is the
 * counterpart of the first-frame dump of BscanFFTsim.cpp:901-909 (savematasdata(.., "debugzpaddedlin", data_ylin)).
 * Rows count A-scans over the whole batch (frame * height + row).  `out` is host memory.  FDOCT_ERR_STATE when the
 * last run was not a staged one. */
int fdoct_get_ylin(fdoct_handle h, long long row0, int nrows, double* out);

/* Which kernel family the last fdoct_process* of this handle launched ("Which kernels run" above).  Results do not
 * depend on it; it exists so that a deployment (and the tests) can see what a configuration gets. */
typedef enum {
  FDOCT_KERNEL_NONE = 0,             /* nothing processed yet */
  FDOCT_KERNEL_FUSED = 1,            /* fused_kernel: power-of-two numfftpoints, no zero-pad upsampling */
  FDOCT_KERNEL_FUSED_TRANSPOSED = 2, /* the same, writing the D x H layout itself */
  FDOCT_KERNEL_FUSED_STAGED = 3,     /* fdoct_set_staged: resample kernel + FFT kernel */
  FDOCT_KERNEL_WAVE = 4,             /* wave_kernel, a shape compiled into the library */
  FDOCT_KERNEL_WAVE_JIT = 5,         /* wave_kernel, compiled for this handle's shape at run time (fdoct_set_jit) */
  FDOCT_KERNEL_GENERIC = 6,          /* generic_kernel: any configuration, one workgroup per A-scan */
  FDOCT_KERNEL_LONG_ROWS = 7         /* rows beyond the LDS (transforms of more than 16384 complex points): rows in HBM */
} fdoct_kernel;
int fdoct_last_kernel(fdoct_handle h);

/* Everything fdoct_process* would do for this handle before its first launch, without frames: the device tables are built
 * and uploaded, the kernel family is resolved and -- where the configuration gets its kernel from the run-time compiler
 * (fdoct_set_jit below) -- that kernel is compiled (0.3-0.9 s) or loaded from the disk cache.  An acquisition loop calls it once
 * after the setters so that its first frame does not stall.  `dtype` / `layout`: what the loop will pass (device frames,
 * 16-byte aligned, packed rows are assumed; a call that differs still works, it just resolves again).  Returns the
 * fdoct_kernel that fdoct_process* will take (> 0), or a negative error code.  Needs the background. */
int fdoct_prepare(fdoct_handle h, fdoct_dtype dtype, fdoct_layout layout);

/* Run-time specialisation (on by default; fdoct_set_jit(h, 0) or FDOCT_JIT=0 in the environment turns it off).  The
 * reference's instrument configurations use zero-pad upsampling and a numfftpoints that is not a power of two
 * (build/BscanFFT.ini:31-32, 51-52); the wave-per-row kernel that serves them is a template over (width, multiplier,
 * numfftpoints) and the library carries instantiations for the shipped shapes and their neighbours.  A handle whose
 * geometry is not among them (another ROI width, bin factor, multiplier or numfftpoints, BscanFFT.ini:9-12, 25-26) has that
 * template compiled for its own geometry by hipRTC -- libhiprtc.so is loaded then, not before -- instead of running the
 * 2.5-5x slower workgroup-per-row kernel; so has a handle of ANY such geometry, the shipped ones included, that uses a
 * pi-shifted frame (fdoct_set_pi_frame), a dark frame (fdoct_set_dark), the band-pass (fdoct_set_bandpass), the row-wise or
 * the whole-frame normalisation (rowwisenormalize, !donotnormalize, the sim variant), the dispersion phase
 * (fdoct_set_dispersion_phase) or a numdisplaypoints beyond numfftpoints / 2: these are compile-time options of the
 * template and the built-in instantiations are the plain set-up.  The compile happens inside fdoct_prepare, or else inside the first fdoct_process* call that needs it (under
 * a second on the build machine; one kernel per sample type, option set and ceil(numdisplaypoints / 64)), is kept for the life of the
 * process and written to $FDOCT_JIT_CACHE (else $XDG_CACHE_HOME/fdoct_amd, else $HOME/.cache/fdoct_amd; FDOCT_JIT_CACHE=""
 * disables the disk cache; a damaged or truncated file is detected and recompiled), so a later process loads it in
 * milliseconds.  Results are those of the built-in instantiations: same source, same compiler flags.  If the template cannot
 * take the shape (a length with a prime factor above 5, fewer than 128 upsampled samples, no room in the LDS), libhiprtc
 * is absent or the compile fails, the call proceeds on the workgroup-per-row kernel and fdoct_jit_note says why (empty
 * string: nothing was refused). */
int fdoct_set_jit(fdoct_handle h, int on);
const char* fdoct_jit_note(fdoct_handle h);
/* Build / deployment check, no GPU needed: compile the wave-per-row kernel for a geometry and an architecture name
 * ("gfx950") exactly as fdoct_set_jit would and return the size of the code object in bytes, or -1 with the reason in `why`. */
long long fdoct_jit_compile_check(int width, int multiplier, int numfftpoints, int numdisplaypoints, fdoct_dtype dtype,
                                  const char* gcn_arch, char* why, int why_len);

/* State exchange for multi-GPU setups: the constant state
 * (background, pi, dark, window, tables, phase) as one opaque blob that rank 0
 * exports and the other ranks import after an RCCL broadcast. */
int fdoct_export_state(fdoct_handle h, void* buf, size_t cap, size_t* used);
int fdoct_import_state(fdoct_handle h, const void* buf, size_t len);

/* The same exchange for a C / C++ host with one process per GPU that holds an RCCL communicator (ncclComm_t, passed as a
 * void* so that this header needs no RCCL header): rank `root` of the communicator exports, ncclBroadcast calls on the
 * handle's stream carry the blob's size and then its bytes (in 4 MB chunks through one staging buffer), every rank imports.
 * Collective: every rank of the communicator calls it with the same root.  librccl.so is looked up at run time on the first
 * call (FDOCT_ERR_UNSUPPORTED if it is absent or does not report a 2.x version).  Failures: everything that can fail on one
 * rank alone (the root's export, the staging buffers) happens before the first collective -- a root that cannot export sends
 * size 0 and EVERY rank returns an error; a rank that returns FDOCT_ERR_NOMEM / FDOCT_ERR_DEVICE with "no collective was
 * entered" in fdoct_last_error has left the others waiting in theirs, and the caller must ncclCommAbort the communicator;
 * once a rank KNOWS the size it never leaves early (a failed copy is reported after the last chunk).  The one exit in between:
 * a rank whose own download of the broadcast size word fails (a local hipMemcpy / stream error), or that reads an implausible
 * size (> 16 GiB), cannot know how many chunk broadcasts follow; it returns FDOCT_ERR_DEVICE with "abort the communicator" in
 * fdoct_last_error, and the caller must ncclCommAbort, as after a pre-collective failure. */
int fdoct_broadcast_state_rccl(fdoct_handle h, void* nccl_comm, int root);

/* One process, several GPUs ("one process per node with one handle+stream per GPU"): a second handle with
 * the same configuration, constant state and run-time settings on another device.  Handles are independent afterwards
 * (later setters apply to the handle they are called on); each may be driven from its own host thread. */
int fdoct_clone_to_device(fdoct_handle h, int device, fdoct_handle* out);
/* Number of gfx950 devices this process sees (0 without a GPU; never an error). */
int fdoct_device_count(void);
/* The frame-shard rule of the path (the same as fdoct_amd/dist.py::shard_frames): part `part` of `nparts` gets the
 * contiguous frames [*first, *first + *count); averaging groups of `averages` frames never straddle parts, sizes differ
 * by at most one group, frames past the last whole group are dropped.  Pure host arithmetic. */
int fdoct_shard_frames(int nframes_total, int averages, int part, int nparts, int* first, int* count);

#ifdef __cplusplus
}
#endif
#endif /* FDOCT_H */
