"""Weak fringes with a pi-shifted / dark frame (main:1132, dark:1269): the reference subtracts them in double; here they are one f32
word each.  Integer-valued frames (what a camera delivers) subtract exactly; averaged or normalised ones add a rounding at the
size of the DC level.  Worst error / tolerance by kernel family, frame kind and fringe amplitude.
gpurun -- python tools/probe_weak_options.py"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, helpers
from fdoct_amd import Config, Reconstructor, synth, capi
rng = np.random.default_rng(1)
for name, (W, H, N, D, M, setup) in {
    "any-option": (2048, 16, 2048, 1024, 1, lambda r: r.set_plan(-1, True)),
    "generic": (2048, 8, 2048, 1024, 1, lambda r: r.set_plan(-2, False)),
    "wave jit": (160, 32, 2560, 320, 4, None),
    "long rows": (2048, 3, 65536, 2048, 8, None)}.items():
    for what in ("pi", "dark", "pi int", "dark int", "pi sim"):   # "pi sim": BscanFFTsim's whole-frame normalisation first, then a pi frame in [0, 1]
        for amp in (2e-2, 1e-3, 1e-4):
            frames, _ = synth.weak_fringe_frame(amp, W, H)
            yb = synth.make_background(W).astype(np.float64)
            cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M,
                         **({"variant": 1} if what == "pi sim" else {}))
            if what == "pi sim":
                yb = yb / 65535.0
            S = synth.source_spectrum(W)
            kw = {}
            if what == "pi sim":
                kw["yp"] = 0.45 * S[None, :] * (1 + 0.01 * rng.standard_normal((H, W)))
            elif what.startswith("pi"):
                yp = 0.45 * 65535 * S[None, :] * (1 + 0.01 * rng.standard_normal((H, W)))   # a pi-shifted frame: DC-sized
                kw["yp"] = np.rint(yp) if "int" in what else yp
            else:
                yd = 0.03 * 65535 * (1 + 0.1 * rng.standard_normal((H, W)))
                kw["yd"] = np.rint(yd) if "int" in what else yd
            r = Reconstructor(cfg); r.set_background(yb)
            if "yp" in kw: r.set_pi_frame(kw["yp"])
            if "yd" in kw: r.set_dark(kw["yd"])
            if setup: setup(r)
            b, d = r.process(frames); k = r.last_kernel(); r.close()
            mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, **kw)
            print("%-10s kernel %d %-8s amp %g: worst err/tol %.3f" % (name, k, what, amp, float(helpers.mag_ratio(b, mag_o).max())))
