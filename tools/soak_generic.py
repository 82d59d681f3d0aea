"""Determinism soak of the 1024-thread workgroup-per-row kernels (two DFT buffers with radix-16 passes / one buffer in place): the
same frames through the same handle SECONDS long, every result compared bit for bit with the first one -- a missing barrier
between the in-place steps would show as a difference.  gpurun -- python tools/soak_generic.py [seconds]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fdoct_amd import DTYPE_U16, Config, Reconstructor, capi, synth  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
out = []
for W, M, N, D in [(4096, 8, 32768, 2048), (4096, 4, 16384, 2048), (3000, 8, 24000, 1500), (4096, 1, 16384, 3000)]:
    H, nframes = 64, 16
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M)
    r = Reconstructor(cfg)
    r.set_background(synth.make_background(W))
    if M == 1:
        r.set_dispersion_phase(synth.dispersion_phase(N))
    fr = torch.from_numpy(synth.make_frames(0, 4, W, H).view(np.int16)).cuda().repeat(nframes // 4, 1, 1).contiguous()
    o = torch.empty((nframes, H, D), dtype=torch.float32, device="cuda")
    r.process_device(fr.data_ptr(), DTYPE_U16, nframes, W * 2, None, o.data_ptr())
    r.synchronize()
    assert r.last_kernel() == capi.KERNEL_GENERIC, r.last_kernel()
    first = o.clone()
    assert torch.equal(first[:4], first[4:8])   # the frames repeat: so do the results
    t0, n, bad = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < SECONDS / 4:
        for _ in range(5):
            o.zero_()
            r.process_device(fr.data_ptr(), DTYPE_U16, nframes, W * 2, None, o.data_ptr())
            r.synchronize()
            bad += int(not torch.equal(o, first))
            n += 1
    out.append({"shape": "%d x%d -> %d, %d bins" % (W, M, N, D), "launches": n, "a_scans": n * nframes * H, "launches_that_differ": bad})
    r.close()
print(json.dumps({"what": "every launch's results against the first launch's, bit for bit", "seconds": SECONDS, "shapes": out}))
sys.exit(1 if any(s["launches_that_differ"] for s in out) else 0)
