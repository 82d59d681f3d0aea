"""ctypes binding of include/fdoct.h (libfdoct_hip.so) and the host-side mirror
of the reference's processing block.

The reference (hn-88/FDOCT) has no operator/plugin API: the block is inlined in
``main()`` (BscanFFT.cpp:1123-1240) and configured by the ini values read at
BscanFFT.cpp:395-484.  ``Config`` therefore carries exactly those names
(``numfftpoints``, ``numdisplaypoints``, ``averages``, ``lambdamin`` ...) and
``Reconstructor`` exposes the state the key handlers capture
(``set_background`` = the 'b' key, main:1000-1075; ``set_pi_frame`` = 'p',
main:1077-1099) plus ``process`` = one pass of main:1123-1240 over a batch.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

DTYPE_U8, DTYPE_U16, DTYPE_F32, DTYPE_F64 = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1
LAYOUT_ROWMAJOR, LAYOUT_TRANSPOSED = 0, 1
VARIANT_MAIN, VARIANT_SIM = 0, 1

_NP2DT = {np.dtype(np.uint8): DTYPE_U8, np.dtype(np.uint16): DTYPE_U16, np.dtype(np.float32): DTYPE_F32,
          np.dtype(np.float64): DTYPE_F64}


class FdoctError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fdoct error %d: %s" % (code, msg))
        self.code = code


class _CConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32),
                ("numfftpoints", C.c_int32), ("numdisplaypoints", C.c_int32),
                ("increasefftpointsmultiplier", C.c_int32), ("averages", C.c_int32),
                ("rowwisenormalize", C.c_int32), ("donotnormalize", C.c_int32), ("movavgn", C.c_int32),
                ("variant", C.c_int32), ("dc_mask", C.c_int32), ("device", C.c_int32),
                ("lambdamin", C.c_double), ("lambdamax", C.c_double)]


class _CTiming(C.Structure):
    _fields_ = [("last_process_ms", C.c_double), ("last_kernel_ms", C.c_double),
                ("resample_stage_ms", C.c_double), ("fft_stage_ms", C.c_double), ("ascans", C.c_uint64),
                ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64)]


@dataclass
class Config:
    """The ini values / locals the block reads (BscanFFT.cpp:395-484, 544-545)."""
    width: int
    height: int
    numfftpoints: int
    numdisplaypoints: int
    increasefftpointsmultiplier: int = 1
    averages: int = 1
    rowwisenormalize: int = 0
    donotnormalize: int = 1
    movavgn: int = 0
    variant: int = VARIANT_MAIN
    dc_mask: int = 1
    device: int = 0
    lambdamin: float = 816e-9
    lambdamax: float = 884e-9


def library_path():
    """The in-tree HIP library.  FDOCT_LIB names another build of the SAME product library (tuning variants from
    tools/mkvariant.sh, A/B runs): never a different implementation, and never anything under oracle/."""
    override = os.environ.get("FDOCT_LIB")
    if override:
        return os.path.abspath(override)
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfdoct_hip.so")


_lib = None

# every symbol include/fdoct.h declares
ABI_SYMBOLS = [
    "fdoct_version", "fdoct_create", "fdoct_destroy", "fdoct_last_error", "fdoct_set_stream",
    "fdoct_set_background", "fdoct_set_pi_frame", "fdoct_set_dark", "fdoct_set_window",
    "fdoct_set_resample_table", "fdoct_set_lambda_range", "fdoct_set_dispersion_phase",
    "fdoct_build_resample_table", "fdoct_build_window", "fdoct_build_colormap_jet", "fdoct_get_resample_table", "fdoct_get_window",
    "fdoct_process", "fdoct_process_async", "fdoct_synchronize", "fdoct_get_timing", "fdoct_set_launch",
    "fdoct_export_state", "fdoct_import_state", "fdoct_set_plan", "fdoct_set_staged", "fdoct_get_ylin", "fdoct_clone_to_device", "fdoct_device_count", "fdoct_shard_frames",
    "fdoct_set_frontend", "fdoct_frontend",
    "fdoct_set_timing", "fdoct_set_averages", "fdoct_set_bandpass", "fdoct_host_alloc", "fdoct_host_free", "fdoct_set_host_staging", "fdoct_get_host_staging", "fdoct_display", "fdoct_set_colormap", "fdoct_get_colormap", "fdoct_lockin_db",
    "fdoct_last_kernel", "fdoct_set_jit", "fdoct_jit_note", "fdoct_jit_compile_check", "fdoct_set_precise_division", "fdoct_prepare", "fdoct_broadcast_state_rccl",
]

# fdoct_kernel (include/fdoct.h): what fdoct_last_kernel returns
KERNEL_NONE, KERNEL_FUSED, KERNEL_FUSED_TRANSPOSED, KERNEL_FUSED_STAGED, KERNEL_WAVE, KERNEL_WAVE_JIT, KERNEL_GENERIC, KERNEL_LONG_ROWS = range(8)


def jit_compile_check(width, multiplier, numfftpoints, numdisplaypoints, dtype=None, gcn_arch="gfx950"):
    """fdoct_jit_compile_check: (code object bytes or -1, reason).  Needs no GPU."""
    buf = C.create_string_buffer(1024)
    n = load_library().fdoct_jit_compile_check(width, multiplier, numfftpoints, numdisplaypoints, DTYPE_U16 if dtype is None else dtype,
                                               gcn_arch.encode(), buf, len(buf))
    return int(n), buf.value.decode()


def shard_frames(nframes_total, averages, part, nparts):
    """fdoct_shard_frames: [start, stop) frames of part `part` (the rule fdoct_amd/dist.py::shard_frames states in Python)."""
    first, count = C.c_int(), C.c_int()
    rc = load_library().fdoct_shard_frames(nframes_total, averages, part, nparts, C.byref(first), C.byref(count))
    if rc:
        raise FdoctError(rc, "fdoct_shard_frames: bad arguments")
    return first.value, first.value + count.value


def load_library():
    """Loads the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1; the first copy of a
    # soname loaded into the process is the one everybody gets.  If torch is going to share this
    # process (tests, bench.py: device tensors and streams come from it), let its copy load first --
    # the other order leaves torch without a visible GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise FdoctError(-3, "%s not found: build it with `make -C fdoct_amd/csrc` "
                             "(or __graft_entry__.build()); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    lib.fdoct_version.restype = C.c_char_p
    lib.fdoct_last_error.restype = C.c_char_p
    lib.fdoct_last_error.argtypes = [C.c_void_p]
    lib.fdoct_create.argtypes = [C.POINTER(_CConfig), C.POINTER(C.c_void_p)]
    lib.fdoct_destroy.argtypes = [C.c_void_p]
    lib.fdoct_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    for name in ("fdoct_set_background", "fdoct_set_pi_frame", "fdoct_set_dark"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
    lib.fdoct_set_window.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.fdoct_set_resample_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.fdoct_set_lambda_range.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.fdoct_set_dispersion_phase.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.fdoct_build_resample_table.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p,
                                               C.c_void_p]
    lib.fdoct_build_window.argtypes = [C.c_int, C.c_void_p]
    lib.fdoct_build_colormap_jet.argtypes = [C.c_void_p]
    lib.fdoct_get_resample_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.fdoct_get_window.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.fdoct_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int]
    lib.fdoct_process_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                        C.c_void_p, C.c_int]
    lib.fdoct_synchronize.argtypes = [C.c_void_p]
    lib.fdoct_get_timing.argtypes = [C.c_void_p, C.POINTER(_CTiming)]
    lib.fdoct_set_launch.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.fdoct_set_plan.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.fdoct_set_staged.argtypes = [C.c_void_p, C.c_int]
    lib.fdoct_get_ylin.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
    lib.fdoct_clone_to_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.fdoct_shard_frames.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.fdoct_set_timing.argtypes = [C.c_void_p, C.c_int]
    lib.fdoct_set_bandpass.argtypes = [C.c_void_p, C.c_int]
    lib.fdoct_set_averages.argtypes = [C.c_void_p, C.c_int]
    lib.fdoct_host_alloc.argtypes = [C.c_size_t]
    lib.fdoct_host_alloc.restype = C.c_void_p
    lib.fdoct_host_free.argtypes = [C.c_void_p]
    lib.fdoct_host_free.restype = None
    lib.fdoct_set_host_staging.argtypes = [C.c_void_p, C.c_int]
    lib.fdoct_get_host_staging.argtypes = [C.c_void_p]
    lib.fdoct_set_frontend.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.fdoct_frontend.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p]
    lib.fdoct_display.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int]
    lib.fdoct_set_colormap.argtypes = [C.c_void_p, C.c_void_p]
    lib.fdoct_get_colormap.argtypes = [C.c_void_p, C.c_void_p]
    lib.fdoct_lockin_db.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
    lib.fdoct_last_kernel.argtypes = [C.c_void_p]
    lib.fdoct_set_jit.argtypes = [C.c_void_p, C.c_int]
    lib.fdoct_set_precise_division.argtypes = [C.c_void_p, C.c_int]
    lib.fdoct_prepare.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.fdoct_broadcast_state_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.fdoct_jit_note.argtypes = [C.c_void_p]
    lib.fdoct_jit_note.restype = C.c_char_p
    lib.fdoct_jit_compile_check.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    lib.fdoct_jit_compile_check.restype = C.c_longlong
    lib.fdoct_export_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.fdoct_import_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    _lib = lib
    return lib


class PinnedArray:
    """A numpy view of pinned host memory from fdoct_host_alloc (full-speed, overlappable PCIe copies in process())."""

    def __init__(self, shape, dtype):
        self._lib = load_library()
        self.array = None
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._ptr = self._lib.fdoct_host_alloc(max(n, 1))
        if not self._ptr:
            raise FdoctError(-4, "fdoct_host_alloc(%d) failed" % n)
        buf = (C.c_char * max(n, 1)).from_address(self._ptr)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self._ptr:
            self.array = None
            self._lib.fdoct_host_free(self._ptr)
            self._ptr = None

    def __del__(self):
        self.free()


def build_resample_table(width, multiplier, numfftpoints, lambdamin, lambdamax):
    """nearestkindex, fractionalk of BscanFFT.cpp:615-698 (host only, no device)."""
    idx = np.zeros(numfftpoints, np.int32)
    frac = np.zeros(numfftpoints, np.float64)
    rc = load_library().fdoct_build_resample_table(width, multiplier, numfftpoints, lambdamin, lambdamax,
                                                   idx.ctypes.data, frac.ctypes.data)
    if rc:
        raise FdoctError(rc, "fdoct_build_resample_table")
    return idx, frac


def build_window(width):
    """barthannwin of BscanFFT.cpp:936-944 (host only)."""
    w = np.zeros(width, np.float64)
    rc = load_library().fdoct_build_window(width, w.ctypes.data)
    if rc:
        raise FdoctError(rc, "fdoct_build_window")
    return w


def build_colormap_jet():
    """COLORMAP_JET (BscanFFT.cpp:1284) as OpenCV builds it: (256, 3) uint8, B,G,R (host only; fdoct_build_colormap_jet)."""
    t = np.zeros((256, 3), np.uint8)
    rc = load_library().fdoct_build_colormap_jet(t.ctypes.data)
    if rc:
        raise FdoctError(rc, "fdoct_build_colormap_jet")
    return t


class Reconstructor:
    """One handle = the processing state of one acquisition loop on one GPU."""

    def __init__(self, cfg: Config):
        self.lib = load_library()
        self.cfg = cfg
        c = _CConfig(C.sizeof(_CConfig), cfg.width, cfg.height, cfg.numfftpoints, cfg.numdisplaypoints,
                     cfg.increasefftpointsmultiplier, cfg.averages, cfg.rowwisenormalize, cfg.donotnormalize,
                     cfg.movavgn, cfg.variant, cfg.dc_mask, cfg.device, cfg.lambdamin, cfg.lambdamax)
        h = C.c_void_p()
        rc = self.lib.fdoct_create(C.byref(c), C.byref(h))
        if rc:
            raise FdoctError(rc, self.lib.fdoct_last_error(None).decode())
        self.h = h

    # -- plumbing
    def _check(self, rc):
        if rc:
            raise FdoctError(rc, self.lib.fdoct_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.fdoct_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ref(self, fn, data):
        if data is None:
            self._check(fn(self.h, None, DTYPE_F64, 0, 0))
            return
        a = np.ascontiguousarray(data)
        if a.dtype not in _NP2DT:
            a = a.astype(np.float64)
        if a.ndim == 1:
            a = a[None, :]
        self._check(fn(self.h, a.ctypes.data, _NP2DT[a.dtype], a.shape[0], a.strides[0]))

    # -- state, named after the reference's locals / key handlers
    def set_background(self, data_yb):
        self._ref(self.lib.fdoct_set_background, data_yb)

    def set_pi_frame(self, data_yp):
        self._ref(self.lib.fdoct_set_pi_frame, data_yp)

    def set_dark(self, data_yd):
        self._ref(self.lib.fdoct_set_dark, data_yd)

    def set_window(self, barthannwin):
        if barthannwin is None:
            self._check(self.lib.fdoct_set_window(self.h, None, 0))
        else:
            w = np.ascontiguousarray(barthannwin, np.float64)
            self._check(self.lib.fdoct_set_window(self.h, w.ctypes.data, w.size))

    def set_resample_table(self, nearestkindex, fractionalk):
        i = np.ascontiguousarray(nearestkindex, np.int32)
        f = np.ascontiguousarray(fractionalk, np.float64)
        self._check(self.lib.fdoct_set_resample_table(self.h, i.ctypes.data, f.ctypes.data, i.size))

    def set_lambda_range(self, lambdamin, lambdamax):
        self._check(self.lib.fdoct_set_lambda_range(self.h, lambdamin, lambdamax))

    def set_dispersion_phase(self, cos_sin_pairs):
        if cos_sin_pairs is None:
            self._check(self.lib.fdoct_set_dispersion_phase(self.h, None, 0))
        else:
            p = np.ascontiguousarray(cos_sin_pairs, np.float32)
            self._check(self.lib.fdoct_set_dispersion_phase(self.h, p.ctypes.data, p.size // 2))

    def get_resample_table(self):
        n = self.cfg.numfftpoints
        i = np.zeros(n, np.int32)
        f = np.zeros(n, np.float64)
        self._check(self.lib.fdoct_get_resample_table(self.h, i.ctypes.data, f.ctypes.data, n))
        return i, f

    def get_window(self):
        n = self.cfg.width   # W entries whatever the zero-pad multiplier is (applied before the upsampling)
        w = np.zeros(n, np.float64)
        self._check(self.lib.fdoct_get_window(self.h, w.ctypes.data, n))
        return w

    def set_stream(self, hip_stream_ptr):
        self._check(self.lib.fdoct_set_stream(self.h, hip_stream_ptr))

    def set_launch(self, threads_per_block=0, blocks=0):
        self._check(self.lib.fdoct_set_launch(self.h, threads_per_block, blocks))

    def set_plan(self, plan_id=-1, force_general_kernel=False):
        self._check(self.lib.fdoct_set_plan(self.h, plan_id, int(force_general_kernel)))

    def set_frontend(self, mediann=0, binx=1, biny=1):
        """medianBlur + INTER_AREA binning (BscanFFT.cpp:953-958): process() then takes RAW camera frames."""
        self._check(self.lib.fdoct_set_frontend(self.h, mediann, binx, biny))
        self._fe = (binx, biny)

    def frontend(self, raw, mediann=0, binx=1, biny=1):
        """The front end on its own: raw (nframes, h, w) u8/u16 -> binned frames (same dtype)."""
        a = np.ascontiguousarray(raw)
        if a.ndim == 2:
            a = a[None]
        n, hh, ww = a.shape
        out = np.empty((n, hh // biny, ww // binx), a.dtype)
        self._check(self.lib.fdoct_frontend(self.h, a.ctypes.data, _NP2DT[a.dtype], n, ww, hh, a.strides[1], mediann, binx, biny,
                                            out.ctypes.data))
        return out

    # -- display post-chain (BscanFFT.cpp:1242-1255, 1284, 1225-1230)
    def set_colormap(self, bgr256=None):
        """256 x (B,G,R) uint8 table for display(colour=True); None = built-in jet."""
        if bgr256 is None:
            self._check(self.lib.fdoct_set_colormap(self.h, None))
        else:
            t = np.ascontiguousarray(bgr256, np.uint8).reshape(768)
            self._check(self.lib.fdoct_set_colormap(self.h, t.ctypes.data))

    def colormap(self):
        t = np.empty(768, np.uint8)
        self._check(self.lib.fdoct_get_colormap(self.h, t.ctypes.data))
        return t.reshape(256, 3)

    def display(self, bscandb, bscanthreshold=-30.0, clampupper=False, colour=False):
        """bscandb: float32 (nbscans, rows, cols) or (rows, cols) on the host.  Returns the u8 display image(s)
        and, with colour=True, the colour-mapped (.., 3) BGR image(s) as well."""
        a = np.ascontiguousarray(bscandb, np.float32)
        single = a.ndim == 2
        if single:
            a = a[None]
        n, r, c = a.shape
        gray = np.empty((n, r, c), np.uint8)
        bgr = np.empty((n, r, c, 3), np.uint8) if colour else None
        self._check(self.lib.fdoct_display(self.h, a.ctypes.data, MEM_HOST, n, r, c, float(bscanthreshold), int(clampupper),
                                           gray.ctypes.data, bgr.ctypes.data if colour else None, MEM_HOST))
        if single:
            gray, bgr = gray[0], (bgr[0] if colour else None)
        return (gray, bgr) if colour else gray

    def display_device(self, d_db_ptr, nbscans, rows, cols, d_gray_ptr, d_bgr_ptr=None, bscanthreshold=-30.0,
                       clampupper=False):
        """Enqueue on the handle's stream; raw device addresses."""
        self._check(self.lib.fdoct_display(self.h, d_db_ptr, MEM_DEVICE, nbscans, rows, cols, float(bscanthreshold),
                                           int(clampupper), d_gray_ptr, d_bgr_ptr, MEM_DEVICE))

    def lockin_db(self, bscan, jscan):
        """J0 lock-in: 20*ln(max(bscan - jscan, 0) + 1e-3)/2.303 for linear B-scans against one saved jscan."""
        b = np.ascontiguousarray(bscan, np.float32)
        j = np.ascontiguousarray(jscan, np.float32)
        n = b.size // j.size
        if n * j.size != b.size:
            raise ValueError("bscan must hold a whole number of jscan-sized B-scans")
        out = np.empty_like(b)
        self._check(self.lib.fdoct_lockin_db(self.h, b.ctypes.data, j.ctypes.data, MEM_HOST, n, j.size, out.ctypes.data))
        return out

    def set_averages(self, averages):
        """Frames averaged per output B-scan from the next call on (the reference's averagestoggle)."""
        self._check(self.lib.fdoct_set_averages(self.h, int(averages)))
        self.cfg.averages = int(averages)

    def set_bandpass(self, on=True):
        """BscanDark.cpp's band-pass inside the zero-pad upsampling (needs increasefftpointsmultiplier > 1)."""
        self._check(self.lib.fdoct_set_bandpass(self.h, int(on)))

    def set_staged(self, on=True):
        """Two-kernel mode (resample stage, FFT stage) for per-stage roofline measurements."""
        self._check(self.lib.fdoct_set_staged(self.h, int(on)))

    def prepare(self, dtype=DTYPE_U16, layout=LAYOUT_ROWMAJOR):
        """Build the tables, resolve the kernel family and compile / load a run-time compiled kernel without frames
        (fdoct_prepare).  Returns KERNEL_*: what process() will take."""
        rc = self.lib.fdoct_prepare(self.h, dtype, layout)
        if rc < 0:
            self._check(rc)
        return rc

    def set_precise_division(self, on=True):
        """1/background as two floats on the fused fast path too (fdoct_set_precise_division)."""
        self._check(self.lib.fdoct_set_precise_division(self.h, int(on)))

    def set_jit(self, on=True):
        """Compile the wave-per-row kernel for this handle's geometry at run time when it is not a built-in shape (fdoct_set_jit)."""
        self._check(self.lib.fdoct_set_jit(self.h, int(on)))

    def jit_note(self):
        """Why the last run-time compile was refused ('' = nothing refused); the call itself fell back and succeeded."""
        return self.lib.fdoct_jit_note(self.h).decode()

    def last_kernel(self):
        """KERNEL_*: the kernel family the last process call launched (fdoct_last_kernel)."""
        return self.lib.fdoct_last_kernel(self.h)

    def clone_to_device(self, device):
        """A second Reconstructor with the same configuration, state and settings on another GPU of this process."""
        import dataclasses
        out = C.c_void_p()
        self._check(self.lib.fdoct_clone_to_device(self.h, device, C.byref(out)))
        r = object.__new__(Reconstructor)
        r.lib = self.lib
        r.cfg = dataclasses.replace(self.cfg, device=device)
        r.h = out
        return r

    def get_ylin(self, row0, nrows):
        """data_ylin rows of the last staged run (BscanFFTsim.cpp:901-909 dumps the first frame's): (nrows, numfftpoints)."""
        out = np.empty((nrows, self.cfg.numfftpoints), np.float64)
        self._check(self.lib.fdoct_get_ylin(self.h, row0, nrows, out.ctypes.data))
        return out

    # -- work
    def _out_shape(self, nframes, layout):
        g = nframes // self.cfg.averages
        if layout == LAYOUT_TRANSPOSED:
            return (g, self.cfg.numdisplaypoints, self.cfg.height)
        return (g, self.cfg.height, self.cfg.numdisplaypoints)

    def process(self, frames, want_db=True, want_bscan=True, layout=LAYOUT_ROWMAJOR, out_bscan=None, out_db=None):
        """frames: numpy (nframes, H, W) u8/u16/f32/f64 on the host.  Returns (bscan, bscandb)
        float32 arrays (None when not requested).  PCIe-inclusive, synchronous.  out_bscan / out_db: caller-owned
        float32 result arrays (e.g. PinnedArray(...).array) instead of fresh ones -- a loop should pass them: a fresh
        array's first-touch page faults cost a 64-frame call four fifths of its rate (profiles/r06_pcie_rate.txt)."""
        a = np.ascontiguousarray(frames)
        if a.ndim == 2:
            a = a[None]
        if a.dtype not in _NP2DT:
            raise FdoctError(-1, "unsupported frame dtype %s" % a.dtype)
        nframes = a.shape[0]
        shp = self._out_shape(nframes, layout)
        def _result(given, want):
            if given is not None:
                if given.dtype != np.float32 or given.shape != shp or not given.flags.c_contiguous:
                    raise FdoctError(-1, "result array must be C-contiguous float32 of shape %s" % (shp,))
                return given
            return np.empty(shp, np.float32) if want else None
        bscan = _result(out_bscan, want_bscan)
        db = _result(out_db, want_db)
        want_bscan, want_db = bscan is not None, db is not None
        self._check(self.lib.fdoct_process(self.h, a.ctypes.data, _NP2DT[a.dtype], MEM_HOST, nframes, a.strides[1],
                                           bscan.ctypes.data if want_bscan else None,
                                           db.ctypes.data if want_db else None, MEM_HOST, layout))
        return bscan, db

    def process_device(self, d_frames_ptr, dtype, nframes, pitch_bytes, d_bscan_ptr, d_db_ptr,
                       layout=LAYOUT_ROWMAJOR):
        """Enqueue on the handle's stream; pointers are raw device addresses (e.g. tensor.data_ptr())."""
        self._check(self.lib.fdoct_process_async(self.h, d_frames_ptr, dtype, nframes, pitch_bytes, d_bscan_ptr,
                                                 d_db_ptr, layout))

    def set_timing(self, on=True):
        """Device-side timing events for process_device() (process() always has them); see fdoct_set_timing."""
        self._check(self.lib.fdoct_set_timing(self.h, int(on)))

    def set_host_staging(self, threads=-1):
        """Pageable host buffers through the handle's pinned staging slots (see fdoct_set_host_staging): -1 default, 0 off, n threads."""
        self._check(self.lib.fdoct_set_host_staging(self.h, int(threads)))

    def host_staging_threads(self):
        """Copy threads a pageable batch would be staged with under the current setting (0: handed to the runtime as it is)."""
        n = self.lib.fdoct_get_host_staging(self.h)
        if n < 0:
            self._check(n)
        return n

    def synchronize(self):
        self._check(self.lib.fdoct_synchronize(self.h))

    def timing(self):
        t = _CTiming()
        self._check(self.lib.fdoct_get_timing(self.h, C.byref(t)))
        return {"process_ms": t.last_process_ms, "kernel_ms": t.last_kernel_ms,
                "resample_stage_ms": t.resample_stage_ms, "fft_stage_ms": t.fft_stage_ms, "ascans": t.ascans,
                "bytes_in": t.bytes_in, "bytes_out": t.bytes_out}

    def broadcast_state_rccl(self, nccl_comm, root=0):
        """Set-up broadcast over a caller-owned RCCL communicator (an ncclComm_t as an integer / c_void_p): fdoct_broadcast_state_rccl."""
        self._check(self.lib.fdoct_broadcast_state_rccl(self.h, C.c_void_p(nccl_comm), int(root)))

    def export_state(self):
        used = C.c_size_t()
        self._check(self.lib.fdoct_export_state(self.h, None, 0, C.byref(used)))
        buf = np.zeros(used.value, np.uint8)
        self._check(self.lib.fdoct_export_state(self.h, buf.ctypes.data, buf.size, C.byref(used)))
        return buf

    def import_state(self, blob):
        b = np.ascontiguousarray(blob, np.uint8)
        self._check(self.lib.fdoct_import_state(self.h, b.ctypes.data, b.size))
