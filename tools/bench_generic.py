"""Throughput of the any-configuration (generic) path on the reference's shipped ini (build/BscanFFT.ini):
320x240 8-bit camera frames, 2x2 software binning -> 160x120, zero-pad x4, N = 2560 (2^9*5), D = 320, 10 averages.
Run on the GPU box: python tools/bench_generic.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fdoct_amd import DTYPE_U8, Config, Reconstructor, synth  # noqa: E402

nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
W, H, N, D, M, A = 160, 120, 2560, 320, 4, 10
cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A,
             lambdamin=840.5e-9, lambdamax=859.5e-9)
r = Reconstructor(cfg)
r.set_background((synth.make_background(W) >> 8).astype(np.uint8) + 1)
r.set_frontend(0, 2, 2)
rng = np.random.default_rng(0)
raw = torch.from_numpy(rng.integers(0, 256, (nframes, 2 * H, 2 * W)).astype(np.uint8)).cuda()
out = torch.empty((nframes // A, H, D), dtype=torch.float32, device="cuda")
st = torch.cuda.Stream()
torch.cuda.synchronize()
r.set_stream(st.cuda_stream)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(5):
        r.process_device(raw.data_ptr(), DTYPE_U8, nframes, 2 * W, None, out.data_ptr())
    r.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("shipped ini: %d raw frames -> %.2f ms, %.3g input A-scans/s (%.3g frames/s), raw input %.1f GB/s"
          % (nframes, dt * 1e3, nframes * H / dt, nframes / dt, nframes * 4 * H * W / dt / 1e9))
r.close()
