"""fdoct_amd -- MI355X-native FD-OCT A-scan reconstruction (drop-in for the
processing block of hn-88/FDOCT, BscanFFT.cpp:1123-1240).

The compute path is the HIP library ``libfdoct_hip.so`` behind the C ABI of
``include/fdoct.h``; this package is the Python host side over that ABI
(ctypes) plus the synthetic interferogram generator used by the tests and the
benchmark.  There is no CPU compute path: importing works anywhere, creating a
``Reconstructor`` needs a gfx950 device.
"""
from .capi import (DTYPE_F32, DTYPE_F64, DTYPE_U8, DTYPE_U16, LAYOUT_ROWMAJOR, LAYOUT_TRANSPOSED, VARIANT_MAIN,
                   VARIANT_SIM, Config, FdoctError, PinnedArray, Reconstructor, build_resample_table, build_window,
                   library_path, load_library)

__all__ = ["FdoctError", "PinnedArray", "Reconstructor", "Config", "build_resample_table", "build_window", "library_path",
           "load_library", "DTYPE_U8", "DTYPE_U16", "DTYPE_F32", "DTYPE_F64", "LAYOUT_ROWMAJOR",
           "LAYOUT_TRANSPOSED", "VARIANT_MAIN", "VARIANT_SIM"]
