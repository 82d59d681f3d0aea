"""the reference's on-disk formats (.ocv Mat dump, Matlab text) -- layout and round trip."""
import io
import struct

import numpy as np

from fdoct_amd import io as fio


def test_ocv_layout_and_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    for dt, depth in ((np.uint8, 0), (np.uint16, 2), (np.float32, 5), (np.float64, 6)):
        a = (rng.random((7, 11)) * 200).astype(dt)
        p = tmp_path / ("m_%d.ocv" % depth)
        fio.write_ocv(str(p), a)
        raw = p.read_bytes()
        assert struct.unpack("<4i", raw[:16]) == (7, 11, depth, 1)      # rows, cols, type, channels (spinj:672-684)
        assert raw[16:] == a.tobytes()
        np.testing.assert_array_equal(fio.read_ocv(str(p)), a)
    c3 = (rng.random((4, 5, 3)) * 255).astype(np.uint8)
    fio.write_ocv(str(tmp_path / "c3.ocv"), c3)
    assert struct.unpack("<4i", (tmp_path / "c3.ocv").read_bytes()[:16]) == (4, 5, 16, 3)  # CV_8UC3 = 16
    np.testing.assert_array_equal(fio.read_ocv(str(tmp_path / "c3.ocv")), c3)


def test_matlab_text_roundtrip():
    m = np.array([[1.5, -2.25, 3.0], [4.0, 5.125, 1e-6]])
    f = io.StringIO()
    fio.write_matlab_text(f, "bscan001", m)
    txt = f.getvalue()
    assert txt.startswith("bscan001=[1.5, -2.25, 3.0;\n 4.0") and txt.rstrip().endswith("];")
    np.testing.assert_array_equal(fio.read_matlab_text(txt, "bscan001"), m)
