"""Whole-frame normalisation (main:1128, always on in BscanFFTsim) costs a min/max pass over the batch before the chain.
Does running the two passes over chunks small enough for the second read to hit the Infinity Cache pay?  Whole batch in one
call against chunks of 8 .. 128 frames (each chunk = min/max pass + chain), same total work."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdoct_amd import DTYPE_U16, VARIANT_SIM, Config, Reconstructor, synth  # noqa: E402

W, H, N, D = 2048, 1000, 2048, 1024
nf = 256
frames = np.tile(synth.make_frames(0, 2, W, H), (nf // 2, 1, 1))
d_in = torch.from_numpy(frames.view(np.int16)).cuda()
d_out = torch.empty((nf, H, D), dtype=torch.float32, device="cuda")
r = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, variant=VARIANT_SIM))
r.set_background(synth.make_background(W))
st = torch.cuda.Stream()
torch.cuda.synchronize()
r.set_stream(st.cuda_stream)
for chunk in (256, 128, 64, 32, 16, 8):
    def step():
        for f0 in range(0, nf, chunk):
            r.process_device(d_in[f0].data_ptr(), DTYPE_U16, chunk, W * 2, None, d_out[f0].data_ptr())
    for i in range(100):
        step()
    r.synchronize()
    t0 = time.perf_counter()
    for i in range(100):
        step()
    r.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print("chunks of %3d frames (%4d MB of samples): %.3f ms  %.1f M A-scans/s" % (chunk, chunk * W * H * 2 >> 20, dt * 1e3, nf * H / dt / 1e6))
r.close()
