"""How long the fused C2 launch takes at the nominal clock and how the package power cap stretches it: after 3 s of idle,
bursts of 1 ... 3000 back-to-back launches (262 frames of 2048 x 1000 u16 each), one HIP event pair around each burst, with
package power and sclk read from the amdgpu hwmon files at the end of the burst.   gpurun -- python tools/burst_clock.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PowerSampler  # noqa: E402
from fdoct_amd import DTYPE_U16, Config, Reconstructor, synth  # noqa: E402

W, H, N, D = 2048, 1000, 2048, 1024
fps, ring = 262, 524
dev = torch.device("cuda", 0)
rec = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=1, device=0,
                           lambdamin=synth.LAMBDAMIN, lambdamax=synth.LAMBDAMAX))
rec.set_background(synth.make_background(W))
frames = np.stack([synth.make_frame(f, W, H) for f in range(4)])
d_ring = torch.from_numpy(frames).to(dev).repeat(ring // 4, 1, 1).contiguous()
d_out = torch.empty((fps * H, D), dtype=torch.float32, device=dev)
stream = torch.cuda.Stream(device=dev)
rec.set_stream(stream.cuda_stream)


def step(i):
    rec.process_device(d_ring[(i % 2) * fps].data_ptr(), DTYPE_U16, fps, W * 2, None, d_out.data_ptr())


for i in range(3):
    step(i)
torch.cuda.synchronize()
ps = PowerSampler(0)
print("launches per burst | ms per launch | A-scans/s | of 8 TB/s | package W, sclk MHz at the end of the burst")
for n in (1, 2, 5, 10, 20, 50, 100, 200, 500, 1000, 3000):
    time.sleep(3.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(n):
        step(i)
    e1.record(stream)
    e1.synchronize()
    p, f = ps._read("power1_input"), ps._read("freq1_input")
    ms = e0.elapsed_time(e1) / n
    rate = fps * H / (ms * 1e-3)
    print("%6d | %.4f | %.1f M | %.3f | %s W, %s MHz" % (n, ms, rate / 1e6, rate * 8192 / 8e12, None if p is None else round(p * 1e-6),
                                                      None if f is None else round(f * 1e-6)))
    sys.stdout.flush()
