"""Regenerates tests/golden/*: (1) the reference's own input fixtures (Matlab files/imgi.png,
backg.png -- data files of the reference's manual test harness, BscanFFTsim.cpp:778,806) decoded
to raw little-endian u16, when /root/reference is present; (2) expected outputs produced by the
CPU oracle (oracle/) for those inputs and for seeded synthetic frames.

The expected outputs are ORACLE outputs, not reference outputs: the reference stores none and
cannot be built here (needs OpenCV).  They freeze the oracle so later edits cannot drift silently
and give the GPU tests size-independent anchors.  Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
from fdoct_amd import VARIANT_MAIN, VARIANT_SIM, Config, synth  # noqa: E402

REF = "/root/reference/Matlab files"
manifest = {}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


if os.path.isdir(REF):
    from PIL import Image
    for name in ("imgi", "backg"):
        a = np.array(Image.open(os.path.join(REF, name + ".png"))).astype("<u2")
        a.tofile(os.path.join(HERE, "%s_u16_96x128.bin" % name))
        manifest[name + "_u16_96x128.bin"] = {"shape": list(a.shape), "sha256": sha(a),
                                              "source": "Matlab files/%s.png (16-bit gray)" % name}

imgi = np.fromfile(os.path.join(HERE, "imgi_u16_96x128.bin"), "<u2").reshape(96, 128)
backg = np.fromfile(os.path.join(HERE, "backg_u16_96x128.bin"), "<u2").reshape(96, 128)

cases = {}
# C1 plumbing, BscanFFTsim.cpp settings: cv::imread -> 8 bit, whole-frame normalise, eps 1e-6
cfg = Config(width=128, height=96, numfftpoints=1024, numdisplaypoints=512, variant=VARIANT_SIM)
mag, bscan, db = helpers.oracle_reference(cfg, (imgi >> 8).astype(np.uint8)[None], (backg >> 8).astype(np.float64))
cases["fixture_sim_u8"] = dict(mag=mag.astype(np.float32), db=db.astype(np.float32))
# same frames, BscanFFT.cpp settings on the 16-bit data, 2-D background
cfg = Config(width=128, height=96, numfftpoints=1024, numdisplaypoints=512, variant=VARIANT_MAIN)
mag, bscan, db = helpers.oracle_reference(cfg, imgi[None], backg.astype(np.float64))
cases["fixture_main_u16"] = dict(mag=mag.astype(np.float32), db=db.astype(np.float32))
# seeded synthetic rows of the benchmark shape (C2) and of C3 (Hann + dispersion phase)
W, H, N, D = 2048, 8, 2048, 1024
frames = synth.make_frames(100, 1, W, H)
yb = synth.make_background(W)
cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
mag, bscan, db = helpers.oracle_reference(cfg, frames, yb)
cases["c2_8rows"] = dict(mag=mag.astype(np.float32), db=db.astype(np.float32))
mag, bscan, db = helpers.oracle_reference(cfg, frames, yb, window=synth.hann_window(W), phase=synth.dispersion_phase(N))
cases["c3_8rows"] = dict(mag=mag.astype(np.float32), db=db.astype(np.float32))
# C4 shape: 4096-point rows, 4 frames averaged
W, H, N, D, A = 4096, 4, 4096, 2048, 4
cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
mag, bscan, db = helpers.oracle_reference(cfg, synth.make_frames(200, A, W, H), synth.make_background(W))
cases["c4_4rows_avg4"] = dict(mag=mag.astype(np.float32), db=db.astype(np.float32))

flat = {}
for k, v in cases.items():
    for kk, a in v.items():
        flat["%s__%s" % (k, kk)] = a
        manifest["%s__%s" % (k, kk)] = {"shape": list(a.shape), "sha256": sha(a)}
np.savez_compressed(os.path.join(HERE, "oracle_outputs.npz"), **flat)
json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
print("wrote", len(flat), "arrays,", os.path.getsize(os.path.join(HERE, "oracle_outputs.npz")) // 1024, "KiB")
