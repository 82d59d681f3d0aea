"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/fdoct.h declares, its host-side tables equal the oracle's bit for bit, and it refuses to
compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as orc
import fdoct_amd
from fdoct_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fdoct.h")).read()
    declared = sorted(set(re.findall(r"\b(fdoct_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = fdoct_amd.load_library()
    for name in declared:
        assert hasattr(lib, name), "missing export " + name
    assert sorted(capi.ABI_SYMBOLS) == declared
    assert b"gfx950" in lib.fdoct_version()


@pytest.mark.parametrize("W,M,N", [(128, 1, 1024), (2048, 1, 2048), (4096, 1, 4096), (640, 4, 2560)])
def test_host_tables_equal_oracle_bit_for_bit(W, M, N):
    """fdoct_build_resample_table / fdoct_build_window (product, C++) vs oracle (C): identical doubles."""
    idx, frac = fdoct_amd.build_resample_table(W, M, N, 816e-9, 884e-9)
    oidx, ofrac = orc.tables(W, M, N, 816e-9, 884e-9)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(frac, ofrac)
    np.testing.assert_array_equal(fdoct_amd.build_window(W), orc.barthann(W))


def test_create_fails_loudly_without_a_gpu_or_with_bad_config():
    import torch
    cfg = fdoct_amd.Config(width=2048, height=8, numfftpoints=2048, numdisplaypoints=1024)
    if not torch.cuda.is_available():
        with pytest.raises(fdoct_amd.FdoctError) as e:
            fdoct_amd.Reconstructor(cfg)
        assert e.value.code == -3 and "no CPU fallback" in str(e.value)
    lib = fdoct_amd.load_library()
    h = C.c_void_p()
    bad = capi._CConfig(7, 2048, 8, 2048, 1024, 1, 1, 0, 1, 0, 0, 1, 0, 816e-9, 884e-9)  # wrong struct_size
    assert lib.fdoct_create(C.byref(bad), C.byref(h)) == -1
    assert b"struct_size" in lib.fdoct_last_error(None)
    assert lib.fdoct_build_window(1, None) == -1
    # null handle is an error everywhere, never a crash
    assert lib.fdoct_synchronize(None) == -1 and lib.fdoct_set_launch(None, 0, 0) == -1
    assert lib.fdoct_destroy(None) == 0


def test_header_is_plain_c_and_a_c_caller_links(tmp_path):
    """include/fdoct.h is the boundary a C or C++ host includes: it must compile as C99 (no C++ in the interface), and a
    plain C program must link against libfdoct_hip.so and run its host-only entry points (no GPU needed: the version
    string, the frame-shard rule of fdoct_shard_frames, the k table of BscanFFT.cpp:615-698)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "fdoct.h"
int main(void) {
  int first = -1, count = -1;
  int32_t idx[1024];
  double frac[1024];
  fdoct_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  if (fdoct_shard_frames(80000, 1, 3, 8, &first, &count) != FDOCT_OK) return 2;
  if (fdoct_build_resample_table(128, 1, 1024, 816e-9, 884e-9, idx, frac) != FDOCT_OK) return 3;
  printf("%s|%d|%d|%d|%d.%d\n", fdoct_version(), first, count, (int)idx[512], FDOCT_VERSION_MAJOR, FDOCT_VERSION_MINOR);
  return 0;
}
''')
    exe = tmp_path / "caller"
    libdir = os.path.dirname(fdoct_amd.library_path())
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-lfdoct_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    ver, first, count, mid, hv = out.stdout.strip().split("|")
    assert "gfx950" in ver and (int(first), int(count)) == (30000, 10000)     # C5: rank 3 of 8 gets frames 30000..39999
    oidx, _ = orc.tables(128, 1, 1024, 816e-9, 884e-9)
    assert int(mid) == int(oidx[512]) and ver.split()[1].startswith(hv)


def test_run_time_compile_of_the_wave_kernel_needs_no_gpu():
    """fdoct_jit_compile_check: the device source that travels inside the library compiles for gfx950 through hipRTC for a
    geometry outside the built-in list (1280 samples, zero-pad x2, numfftpoints 2560), and a geometry the template cannot take
    (half-length transforms with a prime factor above 5) is refused with a reason instead."""
    from fdoct_amd import capi
    n, why = capi.jit_compile_check(1280, 2, 2560, 400)
    assert n > 4096 and why == "", (n, why)
    n, why = capi.jit_compile_check(208, 4, 2560, 320)
    assert n == -1 and "cannot take this shape" in why, (n, why)
    n, why = capi.jit_compile_check(1280, 2, 2560, 400, gcn_arch="gfx000")
    assert n == -1 and why, (n, why)


def opencv_jet_numpy():
    """OpenCV's COLORMAP_JET table restated with numpy float32 arithmetic, step by step as imgproc/src/colormap.cpp builds it:
    Octave's jet(256) rounded to float (the literals of the source), X = linspace(0.f, 1.f, 256), interp1(X, channel, X) in
    float -- low = i - 1, high = i, Y[low] + (X[i] - X[low]) * (Y[high] - Y[low]) / (X[high] - X[low]) --, convertTo(CV_8U, 255.)
    = round-half-even(v * 255.f).  B,G,R."""
    f = np.float32
    i = np.arange(256)
    x = i * (1.0 / 255.0)
    r = ((x >= 3 / 8) & (x < 5 / 8)) * (4 * x - 3 / 2) + ((x >= 5 / 8) & (x < 7 / 8)) + (x >= 7 / 8) * (-4 * x + 9 / 2)
    g = ((x >= 1 / 8) & (x < 3 / 8)) * (4 * x - 1 / 2) + ((x >= 3 / 8) & (x < 5 / 8)) + ((x >= 5 / 8) & (x < 7 / 8)) * (-4 * x + 7 / 2)
    b = (x < 1 / 8) * (4 * x + 1 / 2) + ((x >= 1 / 8) & (x < 3 / 8)) + ((x >= 3 / 8) & (x < 5 / 8)) * (-4 * x + 5 / 2)
    step = f(1.0) / f(255.0)
    X = (f(0.0) + i.astype(f) * step).astype(f)
    out = np.zeros((256, 3), np.uint8)
    for ch, y64 in enumerate((b, g, r)):
        Y = y64.astype(f)
        lut = np.empty(256, f)
        lut[0] = Y[0] + (X[0] - X[0]) * (Y[1] - Y[0]) / (X[1] - X[0])
        lo, hi = slice(0, 255), slice(1, 256)
        num = ((X[hi] - X[lo]) * (Y[hi] - Y[lo])).astype(f)
        lut[1:] = (Y[lo] + (num / (X[hi] - X[lo])).astype(f)).astype(f)
        out[:, ch] = np.clip(np.rint((lut * f(255.0)).astype(f)), 0, 255).astype(np.uint8)     # np.rint: half to even
    return out


def test_builtin_colormap_is_opencv_jet_by_construction():
    """fdoct_build_colormap_jet (C++, float, operation by operation) against the numpy restatement of the same OpenCV recipe,
    the end points every OpenCV build shows, and the structure of the table: every Octave value is (k + 1/2) / 255, so an entry
    is k or k + 1 -- the float roundings decide which, and the two implementations must decide alike."""
    from fdoct_amd.capi import build_colormap_jet
    got = build_colormap_jet()
    want = opencv_jet_numpy()
    np.testing.assert_array_equal(got, want)
    assert tuple(got[0]) == (128, 0, 0) and tuple(got[255]) == (0, 0, 128)            # B,G,R: dark blue .. dark red
    assert (got[96:160, 1] == 255).all() and (got[32:96, 0] == 255).all() and (got[160:224, 2] == 255).all()
    i = np.arange(256)
    rise = 4 * i[96:160] - 382.5                                                       # red on its rising ramp: k + 1/2
    assert np.all((got[96:160, 2] == np.floor(rise)) | (got[96:160, 2] == np.ceil(rise)))
    assert (np.diff(got[96:160, 2].astype(int)) >= 3).all() and (np.diff(got[96:160, 2].astype(int)) <= 5).all()
