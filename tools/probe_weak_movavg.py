"""Weak fringes with smoothmovavg (main:247-304, 990-991): worst error / tolerance by moving-average length and fringe amplitude.
gpurun -- python tools/probe_weak_movavg.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, helpers
from fdoct_amd import Config, Reconstructor, synth
for name, (W, H, N, D, M, setup) in {"fused": (2048, 16, 2048, 1024, 1, None), "workgroup-per-row": (2048, 8, 2048, 1024, 1, lambda r: r.set_plan(-2, False)),
                                      "wave-per-row": (160, 32, 2560, 320, 4, None)}.items():
    for n in (1, 2, 3):
        for amp in (2e-2, 1e-3, 1e-4):
            frames, _ = synth.weak_fringe_frame(amp, W, H)
            yb = synth.make_background(W).astype(np.float64)
            cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, movavgn=n)
            r = Reconstructor(cfg); r.set_background(yb); r.set_precise_division(True)
            if setup: setup(r)
            b, d = r.process(frames); k = r.last_kernel(); r.close()
            mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb)
            print("%-18s kernel %d movavgn %d (%d taps) amp %g: worst err/tol %.3f" % (name, k, n, 2 * n + 2, amp, float(helpers.mag_ratio(b, mag_o).max())))
