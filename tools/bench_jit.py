"""What run-time specialisation (fdoct_set_jit) buys: geometries an operator could type into the ini that are not among the
library's compiled wave-per-row shapes, timed on the workgroup-per-row kernel (the default for them) and on the kernel hipRTC
compiles for them, with the compile time of the first call.  8-bit frames, 10 averages, dB B-scans out, 0.3 s ramp + 1 s timed.
Run on the GPU box: python tools/bench_jit.py"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FDOCT_JIT_CACHE", tempfile.mkdtemp(prefix="fdoct_jit_"))   # a fresh cache: the compile is timed
import torch  # noqa: E402

from fdoct_amd import DTYPE_U8, Config, Reconstructor, capi, synth  # noqa: E402

# width after binning, multiplier, numfftpoints, numdisplaypoints: zero-pad x2 of a 1280-wide camera; x4 of 320 samples into 1280
# points; no zero-pad, 1920 points; a 5120-point transform; a built-in neighbour shape displayed deeper than its compiled variants
SHAPES = [(1280, 2, 2560, 320), (320, 4, 1280, 320), (960, 1, 1920, 320), (640, 4, 5120, 512), (480, 4, 2560, 1000), (160, 2, 1280, 160),
          (200, 4, 2560, 320), (600, 4, 2560, 320), (1000, 4, 2560, 320),   # ROIs of 200 / 600 / 1000 columns: rows that do not split evenly over 64 lanes
          # round 4: options of the template that used to stay on the workgroup-per-row kernel -- the dispersion phase (complex rows)
          # on shipped-ini geometries, and a display beyond numfftpoints / 2
          (160, 4, 2560, 320, "phase"), (640, 4, 2560, 320, "phase"), (640, 1, 640, 320, "phase"), (160, 4, 2560, 2560, "phase"),
          (160, 4, 2560, 2000), (640, 4, 2560, 2560)]
if len(sys.argv) > 1 and sys.argv[1] == "new":
    SHAPES = [s for s in SHAPES if len(s) > 4 or s[3] > s[2] // 2]
A, H = 10, 240
for shape in SHAPES:
    W, M, N, D = shape[:4]
    with_phase = len(shape) > 4
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A,
                 lambdamin=840.5e-9, lambdamax=859.5e-9)
    r = Reconstructor(cfg)
    r.set_background((synth.make_background(max(W, 64))[:W] >> 8).astype(np.uint8) + 1)
    if with_phase:
        r.set_dispersion_phase(synth.dispersion_phase(N))
    nframes = max(A, (256 << 20) // (W * H) // A * A)
    one = np.random.default_rng(0).integers(0, 200, (A, H, W)).astype(np.uint8)
    raw = torch.from_numpy(one).cuda().repeat(nframes // A, 1, 1).contiguous()
    out = torch.empty((nframes // A, H, D), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    r.set_stream(st.cuda_stream)

    def run_for(seconds):
        n, t0 = 0, time.perf_counter()
        while True:
            for _ in range(3):
                r.process_device(raw.data_ptr(), DTYPE_U8, nframes, W, None, out.data_ptr())
            r.synchronize()
            n += 3
            dt = time.perf_counter() - t0
            if dt >= seconds:
                return dt / n

    res = {}
    for jit in (False, True):
        r.set_jit(jit)
        t0 = time.perf_counter()
        r.process_device(raw.data_ptr(), DTYPE_U8, nframes, W, None, out.data_ptr())
        r.synchronize()
        first = time.perf_counter() - t0
        run_for(0.3)
        res[jit] = (nframes * H / run_for(1.0), r.last_kernel(), first, out[:2].clone())
    assert res[False][1] == capi.KERNEL_GENERIC and res[True][1] == capi.KERNEL_WAVE_JIT, (res[False][1], res[True][1], r.jit_note())
    rel = float((res[True][3] - res[False][3]).abs().max())
    print("%5d x%d -> %4d, %4d bins%s: workgroup-per-row %8.3g input A-scans/s, compiled for the shape %8.3g (x %.2f); first call %.2f s "
          "(compile + load); max |dB difference| between the two %.2g" % (W, M, N, D, ", dispersion phase" if with_phase else "", res[False][0], res[True][0], res[True][0] / res[False][0], res[True][2], rel))
    r.close()
