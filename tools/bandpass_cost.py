"""What BscanDark's band-pass (dark:218-236) costs since the row is formed and its kept bins are evaluated in double (round 6):
rate with the band-pass off and on, device-resident frames, on the wave-per-row kernel and on the workgroup-per-row kernel.
usage (gpurun): python3 tools/bandpass_cost.py > gpurun_out/r6_bandpass_cost.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fdoct_amd import Config, Reconstructor, capi, synth  # noqa: E402

FAM = {capi.KERNEL_GENERIC: "workgroup-per-row", capi.KERNEL_LONG_ROWS: "long-row path", capi.KERNEL_WAVE: "wave-per-row",
       capi.KERNEL_WAVE_JIT: "wave-per-row (run-time compiled)"}


def rate(W, M, N, D, bandpass, plan, H=240, nframes=256, reps=5):
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M)
    r = Reconstructor(cfg)
    r.set_background(synth.make_background(max(W, 64))[:W].astype(np.float64) + 10)
    r.set_bandpass(bool(bandpass))
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


print("device-resident u16 frames, 240 lines, 256 frames per call (61 440 A-scans), dB out; A-scans/s")
for shp in [(160, 4, 2560, 320), (640, 4, 2560, 320), (720, 4, 2880, 360), (1280, 2, 2560, 640), (320, 4, 1280, 320), (2048, 2, 4096, 1024), (135, 2, 512, 256)]:
    for plan, name in ((-1, "default route"), (-2, "workgroup-per-row forced")):
        off, f0 = rate(*shp, 0, plan)
        on, f1 = rate(*shp, 1, plan)
        print("W=%4d M=%d N=%d D=%d (%d bins kept), %s: off %.3g on %s | on %.3g on %s | on / off %.2f" % (
            *shp, max(shp[0] // 10 - 3, 0), name, off, FAM.get(f0, f0), on, FAM.get(f1, f1), on / off))
