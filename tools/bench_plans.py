"""Throughput of wave-per-row shapes whose transforms take the in-register radix-9 / radix-15 passes (A/B of the pass plans:
build the library with -DFDOCT_WAVE_R9_MIN=100000 / -DFDOCT_WAVE_R15_MIN=100000 on fdoct_wave.hip and fdoct_capi.cpp and point
FDOCT_LIB at it; FDOCT_JIT=0 keeps run-time compiled kernels, which carry the default plans, out of the comparison).
python tools/bench_plans.py [W,M,N,D ...]   (8-bit 240-row frames, 10 averages, dB out, 0.3 s ramp + 1 s timed)"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from fdoct_amd import DTYPE_U8, Config, Reconstructor, capi, synth
A, H = 10, 240
SHAPES = [(240, 4, 2560, 320), (480, 4, 2560, 320), (960, 4, 2560, 320), (120, 4, 2560, 320)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for W, M, N, D in SHAPES:
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A, lambdamin=840.5e-9, lambdamax=859.5e-9)
    r = Reconstructor(cfg)
    r.set_background((synth.make_background(max(W, 64))[:W] >> 8).astype(np.uint8) + 1)
    nframes = max(A, (256 << 20) // (W * H) // A * A)
    one = np.random.default_rng(0).integers(0, 200, (A, H, W)).astype(np.uint8)
    raw = torch.from_numpy(one).cuda().repeat(nframes // A, 1, 1).contiguous()
    out = torch.empty((nframes // A, H, D), dtype=torch.float32, device="cuda")
    def run_for(seconds):
        n, t0 = 0, time.perf_counter()
        while True:
            for _ in range(3):
                r.process_device(raw.data_ptr(), DTYPE_U8, nframes, W, None, out.data_ptr())
            r.synchronize(); n += 3
            dt = time.perf_counter() - t0
            if dt >= seconds: return dt / n
    run_for(0.3)
    print("%4d x%d -> %d: %.3g input A-scans/s (kernel family %d)" % (W, M, N, nframes * H / run_for(1.0), r.last_kernel()))
    r.close()
