"""PCIe-inclusive rate of the host-buffer entry point (fdoct_process): frames in host memory in, dB out to host
memory, per call -- pageable numpy buffers and pinned ones (fdoct_host_alloc).  Not the bench.py value (that one is
HBM-resident)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from fdoct_amd import Config, PinnedArray, Reconstructor, synth  # noqa: E402

W, H, N, D = 2048, 1000, 2048, 1024
NF = 64
rec = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
rec.set_background(synth.make_background(W))
frames = np.tile(synth.make_frames(0, 4, W, H), (NF // 4, 1, 1))  # 64 frames = 256 MiB


def rate(fr, out_db, label):
    rec.process(fr, want_bscan=False, out_db=out_db)
    t0 = time.perf_counter()
    for _ in range(5):
        _, db = rec.process(fr, want_bscan=False, out_db=out_db)
    dt = (time.perf_counter() - t0) / 5
    print("PCIe-inclusive, %s: %.1f M A-scans/s (%.1f ms per %d-frame call, %.1f GB/s of host traffic)" %
          (label, NF * H / dt / 1e6, dt * 1e3, NF, (fr.nbytes + NF * H * D * 4) / dt / 1e9))
    return db


ref = rate(frames, None, "pageable").copy()
pin_in = PinnedArray(frames.shape, frames.dtype)
pin_out = PinnedArray((NF, H, D), np.float32)
pin_in.array[...] = frames
got = rate(pin_in.array, pin_out.array, "pinned  ")
assert np.array_equal(got, ref), "pinned and pageable results differ"
rec.close()
