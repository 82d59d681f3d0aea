"""ctypes binding of oracle/libfdoct_oracle.so (the CPU restatement).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (fdoct_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "libfdoct_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [
        ("W", C.c_int), ("H", C.c_int), ("N", C.c_int), ("D", C.c_int), ("M", C.c_int),
        ("rowwisenormalize", C.c_int), ("donotnormalize", C.c_int),
        ("movavgn", C.c_int), ("bandpass", C.c_int), ("threads", C.c_int), ("truth", C.c_int),
    ]


def build():
    # the sanitizer run of the restatement (tests/test_oracle.py::test_oracle_under_asan_and_ubsan) points this at
    # libfdoct_oracle_asan.so
    if os.environ.get("FDOCT_ORACLE_SO"):
        return os.environ["FDOCT_ORACLE_SO"]
    src = os.path.join(_ORACLE_DIR, "fdoct_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_version.restype = C.c_char_p
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def tables(W, M, N, lmin, lmax, debug=False):
    idx = np.zeros(N, np.int32)
    frac = np.zeros(N, np.float64)
    k = np.zeros(M * W) if debug else None
    kl = np.zeros(N) if debug else None
    dk = np.zeros(M * W) if debug else None
    lib().orc_tables(C.c_int(W), C.c_int(M), C.c_int(N), C.c_double(lmin), C.c_double(lmax),
                     _p(idx, C.c_int32), _p(frac, C.c_double), _p(k, C.c_double),
                     _p(kl, C.c_double), _p(dk, C.c_double))
    return (idx, frac, k, kl, dk) if debug else (idx, frac)


def barthann(W):
    w = np.zeros(W)
    lib().orc_barthann(C.c_int(W), _p(w, C.c_double))
    return w


def normalize_minmax(y, lo=0.0, hi=1.0):
    y = np.ascontiguousarray(y, np.float64).copy()
    lib().orc_normalize_minmax(_p(y, C.c_double), C.c_size_t(y.size), C.c_double(lo), C.c_double(hi))
    return y


def normalizerows(y, lo=0.0, hi=1.0):
    y = np.ascontiguousarray(y, np.float64).copy()
    H, W = y.shape
    lib().orc_normalizerows(_p(y, C.c_double), C.c_int(H), C.c_int(W), C.c_double(lo), C.c_double(hi))
    return y


def smoothmovavg(y, n):
    y = np.ascontiguousarray(y, np.float64)
    out = np.empty_like(y)
    H, W = y.shape
    lib().orc_smoothmovavg(_p(y, C.c_double), _p(out, C.c_double), C.c_int(H), C.c_int(W), C.c_int(n))
    return out


def zeropadrowwise(y, M, bandpass=0):
    y = np.ascontiguousarray(y, np.float64)
    H, W = y.shape
    out = np.empty((H, M * W))
    lib().orc_zeropadrowwise(_p(y, C.c_double), C.c_int(H), C.c_int(W), C.c_int(M), C.c_int(bandpass),
                             _p(out, C.c_double))
    return out


def dft_rows_f32(z, inverse=True, scale=False):
    """z: complex64 (H,N).  Returns complex64."""
    z = np.ascontiguousarray(z, np.complex64).copy()
    H, N = z.shape
    lib().orc_dft_rows_f32(z.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(H), C.c_int(N),
                           C.c_int(int(inverse)), C.c_int(int(scale)))
    return z


def dft_rows_f64(z, inverse=True, scale=False):
    z = np.ascontiguousarray(z, np.complex128).copy()
    H, N = z.shape
    lib().orc_dft_rows_f64(z.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(H), C.c_int(N),
                           C.c_int(int(inverse)), C.c_int(int(scale)))
    return z


def make_params(W, H, N, D, M=1, rowwisenormalize=0, donotnormalize=1, movavgn=0, bandpass=0, threads=1, truth=0):
    """truth=1: every float step of the reference (zero-pad DFTs, narrowing, cv::dft, magnitude) in double -- the exact value
    of the reference's mathematics, the adjudicator of helpers.check_truth (oracle/fdoct_oracle.h, orc_params.truth)."""
    return OrcParams(W, H, N, D, M, rowwisenormalize, donotnormalize, movavgn, bandpass, threads, truth)


def _full(a, H, W):
    """Broadcast a 1-row (W,) or (1,W) array to the H x W doubles the reference holds."""
    a = np.asarray(a, np.float64)
    if a.ndim == 1:
        a = a[None, :]
    if a.shape[0] == 1:
        a = np.broadcast_to(a, (H, W))
    return np.ascontiguousarray(a)


def frame_to_mag(p, data_y, yb, yp, win, idx, frac, yd=None, phase=None, want_ylin=False):
    H, W, N = p.H, p.W, p.N
    data_y = np.ascontiguousarray(data_y, np.float64)
    yb = _full(yb, H, W)
    yp = _full(np.zeros(W) if yp is None else yp, H, W)
    ydf = None if yd is None else _full(yd, H, W)
    win = np.ascontiguousarray(win, np.float64)
    idx = np.ascontiguousarray(idx, np.int32)
    frac = np.ascontiguousarray(frac, np.float64)
    ph = None if phase is None else np.ascontiguousarray(phase, np.float32)
    mag = np.empty((H, N), np.float64 if p.truth else np.float32)
    ylin = np.empty((H, N)) if want_ylin else None
    if p.truth:
        rc = lib().orc_frame_to_mag_f64(C.byref(p), _p(data_y, C.c_double), _p(yb, C.c_double), _p(yp, C.c_double),
                                        _p(ydf, C.c_double), _p(win, C.c_double), _p(idx, C.c_int32),
                                        _p(frac, C.c_double), _p(ph, C.c_float), _p(mag, C.c_double),
                                        _p(ylin, C.c_double))
        assert rc == 0
        return (mag, ylin) if want_ylin else mag
    rc = lib().orc_frame_to_mag(C.byref(p), _p(data_y, C.c_double), _p(yb, C.c_double), _p(yp, C.c_double),
                                _p(ydf, C.c_double), _p(win, C.c_double), _p(idx, C.c_int32),
                                _p(frac, C.c_double), _p(ph, C.c_float), _p(mag, C.c_float),
                                _p(ylin, C.c_double))
    assert rc == 0
    return (mag, ylin) if want_ylin else mag


def process_u16(p, A, eps, frames, yb, yp, win, idx, frac, yd=None, phase=None, sim_copy=False):
    """frames: uint16 (nframes,H,W).  Returns (mag_rowmajor (G,H,D), bscan (G,D,H), bscandb (G,D,H)).
    sim_copy: BscanFFTsim.cpp's grouping (sim:936-947): the last frame of every group of A, copied, not divided."""
    H, W, D = p.H, p.W, p.D
    frames = np.ascontiguousarray(frames, np.uint16)
    nframes = frames.shape[0]
    G = nframes // A
    yb = _full(yb, H, W)
    yp = _full(np.zeros(W) if yp is None else yp, H, W)
    ydf = None if yd is None else _full(yd, H, W)
    win = np.ascontiguousarray(win, np.float64)
    idx = np.ascontiguousarray(idx, np.int32)
    frac = np.ascontiguousarray(frac, np.float64)
    ph = None if phase is None else np.ascontiguousarray(phase, np.float32)
    mag = np.empty((G, H, D))
    bscan = np.empty((G, D, H))
    db = np.empty((G, D, H))
    fn = lib().orc_process_u16_sim if sim_copy else lib().orc_process_u16
    rc = fn(C.byref(p), C.c_int(A), C.c_double(eps), _p(frames, C.c_uint16), C.c_int(nframes),
                               _p(yb, C.c_double), _p(yp, C.c_double), _p(ydf, C.c_double), _p(win, C.c_double),
                               _p(idx, C.c_int32), _p(frac, C.c_double), _p(ph, C.c_float),
                               _p(mag, C.c_double), _p(bscan, C.c_double), _p(db, C.c_double))
    assert rc == 0, rc
    return mag, bscan, db


def median_blur(img, n):
    a = np.ascontiguousarray(img, np.uint16)
    out = np.empty_like(a)
    h, w = a.shape
    lib().orc_median_blur_u16(_p(a, C.c_uint16), _p(out, C.c_uint16), C.c_int(w), C.c_int(h), C.c_int(n))
    return out


def resize_area(img, binx, biny):
    a = np.ascontiguousarray(img, np.uint16)
    h, w = a.shape
    out = np.empty((h // biny, w // binx), np.uint16)
    lib().orc_resize_area_u16(_p(a, C.c_uint16), _p(out, C.c_uint16), C.c_int(w), C.c_int(h), C.c_int(binx), C.c_int(biny))
    return out


def display_u8(db, thr=-30.0, clampupper=False):
    a = np.ascontiguousarray(db, np.float64)
    rows, cols = a.shape
    out = np.empty((rows, cols), np.uint8)
    lib().orc_display_u8(_p(a, C.c_double), C.c_int(rows), C.c_int(cols), C.c_double(thr), C.c_int(int(clampupper)),
                         _p(out, C.c_uint8))
    return out


def apply_lut(gray, lut):
    g = np.ascontiguousarray(gray, np.uint8)
    lut = np.ascontiguousarray(lut, np.uint8).reshape(768)
    out = np.empty(g.shape + (3,), np.uint8)
    lib().orc_apply_lut(_p(g, C.c_uint8), C.c_size_t(g.size), _p(lut, C.c_uint8), _p(out, C.c_uint8))
    return out


def lockin_db(bscan, jscan):
    b = np.ascontiguousarray(bscan, np.float64)
    j = np.ascontiguousarray(jscan, np.float64)
    out = np.empty_like(b)
    lib().orc_lockin_db(_p(b, C.c_double), _p(j, C.c_double), C.c_size_t(b.size), _p(out, C.c_double))
    return out
