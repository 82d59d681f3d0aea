"""The wave-per-row kernel's compile-time pruning (blocks nobody reads, input blocks that are zero) must not change a bit:
runs a few shapes through the run-time compiled kernel with FDOCT_JIT_DEFINES = "" and "-DFDOCT_WAVE_PRUNE=0 -DFDOCT_WAVE_ZPRUNE=0"
(two child processes, a JIT cache each) and compares the outputs byte for byte.  usage (gpurun): python3 tools/prune_bitcheck.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(160, 4, 2560, 320, 10), (640, 4, 2560, 320, 3), (720, 4, 2880, 360, 2), (640, 4, 2560, 500, 2), (2000, 2, 80, 29, 1),
          (1080, 2, 150, 10, 1), (320, 4, 2560, 64, 2), (640, 1, 640, 320, 2), (300, 8, 2400, 100, 1)]

if len(sys.argv) > 1:  # child: compute and dump
    sys.path.insert(0, ROOT)
    from fdoct_amd import Config, Reconstructor, capi, synth
    out = {}
    for i, (W, M, N, D, A) in enumerate(SHAPES):
        H = 24
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A)
        frames = synth.make_frames(7 + i, 2 * A, max(W, 64), H)[:, :, :W].copy()
        r = Reconstructor(cfg)
        r.set_background(synth.make_background(max(W, 64))[:W].astype(np.float64) + 10.0)
        if i < 4 or i in (6, 7):   # shapes compiled into the library: a pi frame sends them to the run-time compiler (OPT_PI)
            r.set_pi_frame(np.full(W, 3.0))
        b, d = r.process(frames)
        out["b%d" % i], out["d%d" % i], out["k%d" % i] = b, d, np.int32(r.last_kernel())
        r.close()
    np.savez(sys.argv[1], **out)
    sys.exit(0)

res = []
for tag, defs in (("on", " "), ("off", "-DFDOCT_WAVE_PRUNE=0 -DFDOCT_WAVE_ZPRUNE=0")):
    env = dict(os.environ, FDOCT_JIT_DEFINES=defs, FDOCT_JIT_CACHE="/tmp/prune_%s" % tag)
    path = "/tmp/prune_%s.npz" % tag
    subprocess.run([sys.executable, os.path.abspath(__file__), path], env=env, check=True)
    res.append(np.load(path))
bad = 0
for i, s in enumerate(SHAPES):
    same = np.array_equal(res[0]["b%d" % i], res[1]["b%d" % i]) and np.array_equal(res[0]["d%d" % i], res[1]["d%d" % i])
    print("W=%d M=%d N=%d D=%d A=%d: kernel family %d / %d, %s" % (*s, res[0]["k%d" % i], res[1]["k%d" % i], "bit-identical" if same else "DIFFERENT"))
    bad += not same
sys.exit(1 if bad else 0)
