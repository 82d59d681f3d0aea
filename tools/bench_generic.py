"""Throughput of the configurations the reference ships (build/*.ini): all of them use a non-power-of-two numfftpoints
and (but for the webcam) the x4 zero-pad upsampling; they run on the wave-per-row kernels (fdoct_wave.hip).  Raw camera
frames in (8- or 16-bit), software binning on the GPU, 10 averages, dB B-scans out.  Each configuration runs 0.3 s untimed
(clock / memory ramp, then the power controller settles), then SECONDS (default 1.0) timed, with the package power and sclk of
the timed part.  Run on the GPU box: python tools/bench_generic.py [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import PowerSampler  # noqa: E402
from fdoct_amd import DTYPE_U8, DTYPE_U16, Config, Reconstructor, synth  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0

# name, raw width, raw height, bits, binvalue, numfftpoints, multiplier, numdisplaypoints, lambdamin, lambdamax
INIS = [
    ("BscanFFT.ini (QHY, ROI 320x240)", 320, 240, 8, 2, 2560, 4, 320, 840.5e-9, 859.5e-9),
    ("BscanFFTspin/peak.ini (1280x960)", 1280, 960, 8, 2, 2560, 4, 320, 840.5e-9, 859.5e-9),
    ("BscanDark.ini (1280x960, 16-bit)", 1280, 960, 16, 2, 2560, 4, 320, 840.5e-9, 859.5e-9),
    ("BscanFFTspinj.ini (720x480, 16-bit)", 720, 480, 16, 1, 2880, 4, 360, 840.5e-9, 859.5e-9),
    ("BscanFFTwebcam.ini (640x480)", 640, 480, 8, 1, 640, 1, 320, 840.5e-9, 859.5e-9),
]
A = 10
for name, rw, rh, bits, binv, N, M, D, lmin, lmax in INIS:
    W, H = rw // binv, rh // binv
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A,
                 lambdamin=lmin, lambdamax=lmax)
    r = Reconstructor(cfg)
    dt_np, dt_id = (np.uint8, DTYPE_U8) if bits == 8 else (np.uint16, DTYPE_U16)
    bg = synth.make_background(W)
    r.set_background((bg >> 8).astype(np.uint8) + 1 if bits == 8 else bg + 1)
    if binv > 1:
        r.set_frontend(0, binv, binv)
    nframes = max(A, (768 << 20) // (rw * rh * (bits // 8)) // A * A)      # ~768 MB of raw frames: >= 15 output A-scans per wave, so the last round of the persistent waves is a small part
    rng = np.random.default_rng(0)
    one = rng.integers(0, 200 if bits == 8 else 40000, (A, rh, rw)).astype(dt_np)
    raw = torch.from_numpy(one.view(np.int16) if bits == 16 else one).cuda().repeat(nframes // A, 1, 1).contiguous()
    out = torch.empty((nframes // A, H, D), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    r.set_stream(st.cuda_stream)
    def run_for(seconds):
        n = 0
        t0 = time.perf_counter()
        while True:
            for k in range(5):
                r.process_device(raw.data_ptr(), dt_id, nframes, rw * (bits // 8), None, out.data_ptr())
            r.synchronize()
            n += 5
            dt = time.perf_counter() - t0
            if dt >= seconds:
                return dt / n

    run_for(0.3)
    ps = PowerSampler(0)
    ps.start()
    best = run_for(SECONDS)
    ps.stop()
    pw = ps.summary() or {}
    print("%-38s W=%4d H=%3d N=%4d M=%d: %9.3g input A-scans/s  (%8.3g camera frames/s, raw input %5.1f GB/s)  %s W, sclk %s MHz"
          % (name, W, H, N, M, nframes * H / best, nframes / best, nframes * rw * rh * (bits // 8) / best / 1e9,
             pw.get("package_w_last_half"), pw.get("sclk_mhz_avg")))
    r.close()
