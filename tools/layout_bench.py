"""Throughput with the reference's transposed (D x H) output layout vs the row-major fast layout (C2 shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fdoct_amd import Config, Reconstructor, synth, DTYPE_U16, LAYOUT_ROWMAJOR, LAYOUT_TRANSPOSED
W, H, N, D = 2048, 1000, 2048, 1024
nf = 262
frames = np.tile(synth.make_frames(0, 2, W, H), (nf // 2, 1, 1))
d_in = torch.from_numpy(frames.view(np.int16)).cuda()
d_db = torch.empty((nf, H, D), dtype=torch.float32, device='cuda')
d_mag = torch.empty((nf, H, D), dtype=torch.float32, device='cuda')
r = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
r.set_background(synth.make_background(W))
st = torch.cuda.Stream(); torch.cuda.synchronize(); r.set_stream(st.cuda_stream)
for name, layout, mag in (("row-major dB", LAYOUT_ROWMAJOR, None), ("transposed dB", LAYOUT_TRANSPOSED, None),
                          ("row-major dB+bscan", LAYOUT_ROWMAJOR, d_mag), ("transposed dB+bscan", LAYOUT_TRANSPOSED, d_mag)):
    mp = mag.data_ptr() if mag is not None else None
    for i in range(300): r.process_device(d_in.data_ptr(), DTYPE_U16, nf, W * 2, mp, d_db.data_ptr(), layout)
    r.synchronize(); t0 = time.perf_counter()
    for i in range(200): r.process_device(d_in.data_ptr(), DTYPE_U16, nf, W * 2, mp, d_db.data_ptr(), layout)
    r.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("%-22s %.3f ms  %.1f M A-scans/s" % (name, dt * 1e3, nf * H / dt / 1e6))
r.close()
