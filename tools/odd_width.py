"""Rate of odd row widths under the zero-pad (main:215-241) next to their even neighbours, device-resident frames:
the default route (round 6: full-length transforms in generic_kernel's LDS buffers), the long-row path forced
(fdoct_set_plan(h, -3): round 5's route) and, for the even neighbour, the wave-per-row and the workgroup-per-row kernel.
usage (gpurun): python3 tools/odd_width.py > gpurun_out/r6_odd_width.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fdoct_amd import Config, Reconstructor, capi, synth  # noqa: E402

FAM = {capi.KERNEL_GENERIC: "workgroup-per-row (LDS)", capi.KERNEL_LONG_ROWS: "long-row path (HBM)", capi.KERNEL_WAVE: "wave-per-row",
       capi.KERNEL_WAVE_JIT: "wave-per-row, run-time compiled"}


def rate(W, M, N, D, plan, H=240, nframes=128, reps=5):
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M)
    r = Reconstructor(cfg)
    r.set_background(synth.make_background(max(W, 64))[:W].astype(np.float64) + 10)
    r.set_plan(plan)
    fr = synth.make_frames(0, 8, max(W, 64), H)[:, :, :W].copy()
    fr = np.ascontiguousarray(np.tile(fr, (nframes // 8, 1, 1)))
    pitch = (W * 2 + 15) // 16 * 16
    buf = torch.zeros(nframes * H * pitch, dtype=torch.uint8, device="cuda")
    buf.view(nframes * H, pitch)[:, :W * 2] = torch.from_numpy(fr.view(np.uint8).reshape(nframes * H, W * 2)).cuda()
    out = torch.empty(nframes * H * D, dtype=torch.float32, device="cuda")
    call = lambda: r.process_device(buf.data_ptr(), capi.DTYPE_U16, nframes, pitch, None, out.data_ptr(), 0)  # noqa: E731
    call()
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    r.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fam = r.last_kernel()
    r.close()
    return nframes * H / dt, fam


print("device-resident u16 frames, 240 lines, 128 frames per call (30 720 A-scans), dB out; A-scans/s")
for odd, even in [((321, 4, 1284, 320), (320, 4, 1280, 320)), ((161, 4, 2560, 320), (160, 4, 2560, 320)), ((225, 3, 1024, 300), (224, 3, 1024, 300)),
                  ((641, 4, 2560, 320), (640, 4, 2560, 320)), ((1281, 2, 2560, 640), (1280, 2, 2560, 640))]:
    ro, fo = rate(*odd, -1)
    rb, fb = rate(*odd, -3)
    re_, fe = rate(*even, -1)
    rg, fg = rate(*even, -2)
    print("W=%4d M=%d N=%d D=%d: %.3g on %s | forced to the %s: %.3g | even neighbour W=%d: %.3g on %s, %.3g on %s | odd / even neighbour's "
          "workgroup-per-row kernel %.2f, odd default / round 5's route %.1f x" % (
              *odd, ro, FAM.get(fo, fo), FAM.get(fb, fb), rb, even[0], re_, FAM.get(fe, fe), rg, FAM.get(fg, fg), ro / rg, ro / rb))
