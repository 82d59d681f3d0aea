"""Achieved HBM rates of the rows either side of the block: front end (median + binning) and display chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fdoct_amd import Config, Reconstructor, synth, DTYPE_U16

W, H, N, D = 2048, 1000, 2048, 1024
r = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
r.set_background(synth.make_background(W))
st = torch.cuda.Stream(); torch.cuda.synchronize(); r.set_stream(st.cuda_stream)

def timeit(fn, reps=50):
    for _ in range(10): fn()
    r.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    r.synchronize(); return (time.perf_counter() - t0) / reps

# display chain on 64 B-scans of D x H dB values
nb = 64
db = torch.randn((nb, D, H), device='cuda') * 20 - 20
g = torch.empty((nb, D, H), dtype=torch.uint8, device='cuda')
c = torch.empty((nb, D, H, 3), dtype=torch.uint8, device='cuda')
px = nb * D * H
dt = timeit(lambda: r.display_device(db.data_ptr(), nb, D, H, g.data_ptr(), None))
print("display grey      : %.3f ms, %.2f Gpixel/s, %.0f GB/s algorithmic (4 B in + 1 B out per pixel; the map pass re-reads the input)" % (dt * 1e3, px / dt / 1e9, px * 5 / dt / 1e9))
dt = timeit(lambda: r.display_device(db.data_ptr(), nb, D, H, g.data_ptr(), c.data_ptr()))
print("display grey+BGR  : %.3f ms, %.2f Gpixel/s, %.0f GB/s algorithmic (4 B in + 4 B out per pixel)" % (dt * 1e3, px / dt / 1e9, px * 8 / dt / 1e9))

# front end: raw 2x-binned camera frames (4096 x 2000 u16) -> 2048 x 1000, with and without the 3x3 median, then the chain
nf = 32
raw = torch.randint(0, 30000, (nf, 2 * H, 2 * W), dtype=torch.int16, device='cuda')
out = torch.empty((nf, H, D), dtype=torch.float32, device='cuda')
for med in (0, 3):
    r.set_frontend(med, 2, 2)
    dt = timeit(lambda: r.process_device(raw.data_ptr(), DTYPE_U16, nf, 2 * W * 2, None, out.data_ptr()), 20)
    print("raw frames, median %d, 2x2 binning + chain: %.3f ms per %d frames, %.1f M A-scans/s, raw input %.0f GB/s"
          % (med, dt * 1e3, nf, nf * H / dt / 1e6, nf * 4 * H * W * 2 / dt / 1e9))
r.close()
