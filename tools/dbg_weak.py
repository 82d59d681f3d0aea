"""Debug aid: where the chain deviates from the oracle on weak fringes (fraction of the DC level given as argv[1])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers
from fdoct_amd import Config, Reconstructor, synth
amp = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
W, H, N, D = 2048, 64, 2048, 1024
lam = synth.lambdas(W); S = synth.source_spectrum(W)
depth = (40.0 + 6.0 * np.arange(H))[:, None] * 1e-6
fringe = amp * np.cos(4 * np.pi * synth.NS * depth / lam[None, :])
rng = np.random.default_rng(5)
I = S[None, :] * (1.0 + fringe)
frames = np.clip(np.rint(I * 0.9 * 65535.0 + rng.uniform(-0.5, 0.5, I.shape)), 0, 65535).astype(np.uint16)[None]
yb = synth.make_background(W)
cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
for general in (False, True):
    r = Reconstructor(cfg); r.set_background(yb)
    if general: r.set_plan(-1, True)
    b, d = r.process(frames); r.close()
    mag_o, _, _ = helpers.oracle_reference(cfg, frames, yb)
    ratio = helpers.mag_ratio(b, mag_o)[0]
    err = np.abs(b - mag_o)[0]
    print("general kernel" if general else "fast path", "rowmax", mag_o[0].max(axis=1)[:3], "worst ratio", ratio.max())
    bad = np.argwhere(ratio > 1.0)
    print("  failing bins:", len(bad), "bin histogram:", np.bincount(bad[:, 1], minlength=8)[:8], "max bin", bad[:, 1].max() if len(bad) else None)
    print("  err by bin (max over rows) bins 0..7:", err.max(axis=0)[:8], " bins>=8 max:", err[:, 8:].max(), " ratio bins>=8 max:", ratio[:, 8:].max())
