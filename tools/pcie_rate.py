"""PCIe-inclusive rate of the host-buffer entry point (fdoct_process): frames in pageable host memory in,
dB out to host memory, per call.  Not the bench.py value (that one is HBM-resident)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdoct_amd import Config, Reconstructor, synth
W, H, N, D = 2048, 1000, 2048, 1024
rec = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
rec.set_background(synth.make_background(W))
frames = np.tile(synth.make_frames(0, 4, W, H), (16, 1, 1))  # 64 frames = 256 MiB
rec.process(frames, want_bscan=False)
t0 = time.perf_counter()
for _ in range(5):
    rec.process(frames, want_bscan=False)
dt = (time.perf_counter() - t0) / 5
print("PCIe-inclusive: %.1f M A-scans/s (%.1f ms per 64-frame call, %.2f GB/s of host traffic)" %
      (64 * H / dt / 1e6, dt * 1e3, (frames.nbytes + 64 * H * D * 4) / dt / 1e9))
