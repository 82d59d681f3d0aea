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
