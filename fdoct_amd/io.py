"""On-disk formats of the reference, for the harness and the tests.

* ``.ocv`` -- the raw cv::Mat dump of BscanFFTspinj.cpp:672-715 (``matwrite``/``matread``): four int32
  (rows, cols, OpenCV type code, channels) followed by the row-major payload.
* Matlab text -- ``name=[a, b, ...;\\n c, d, ...];`` as ``operator<<(cv::Mat)`` prints it and
  ``savematasdata`` writes it (BscanFFT.cpp:333-339).
"""
import struct

import numpy as np

# OpenCV depth codes (CV_8U .. CV_64F); type = depth + ((channels - 1) << 3)
_DEPTH2NP = {0: np.uint8, 1: np.int8, 2: np.uint16, 3: np.int16, 4: np.int32, 5: np.float32, 6: np.float64}
_NP2DEPTH = {np.dtype(v): k for k, v in _DEPTH2NP.items()}


def write_ocv(path, mat):
    a = np.ascontiguousarray(mat)
    if a.ndim == 2:
        a = a[:, :, None]
    rows, cols, ch = a.shape
    depth = _NP2DEPTH[a.dtype]
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", rows, cols, depth + ((ch - 1) << 3), ch))
        f.write(a.tobytes())


def read_ocv(path):
    with open(path, "rb") as f:
        rows, cols, typ, ch = struct.unpack("<4i", f.read(16))
        depth, ch2 = typ & 7, (typ >> 3) + 1
        if ch2 != ch or depth not in _DEPTH2NP or rows < 0 or cols < 0:
            raise ValueError("not an .ocv Mat dump: %s" % path)
        dt = np.dtype(_DEPTH2NP[depth])
        a = np.frombuffer(f.read(rows * cols * ch * dt.itemsize), dt)
    a = a.reshape(rows, cols, ch)
    return a[:, :, 0] if ch == 1 else a


def write_matlab_text(f, name, mat):
    """f: text file object.  Same layout as cv's default Mat formatter: '[a, b;\\n c, d]'."""
    m = np.asarray(mat)
    f.write("%s=[" % name)
    for r in range(m.shape[0]):
        f.write(", ".join(repr(float(x)) if m.dtype.kind == "f" else str(int(x)) for x in m[r]))
        f.write(";\n " if r + 1 < m.shape[0] else "")
    f.write("];\n")


def read_matlab_text(text, name):
    """Parses one 'name=[...];' assignment back into a float64 matrix."""
    start = text.index(name + "=[") + len(name) + 2
    body = text[start:text.index("]", start)]
    rows = [r for r in body.replace("\n", " ").split(";") if r.strip()]
    return np.array([[float(x) for x in r.split(",")] for r in rows])
