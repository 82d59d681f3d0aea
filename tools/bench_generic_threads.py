"""Workgroup size of the workgroup-per-row kernel (generic_kernel<NT, MINB>): rows of which the LDS holds only one or two per
CU run with 1024 / 512 threads instead of 256 (round 4).  FDOCT_GENERIC_THREADS=256|512|1024 forces one size; this tool runs
each shape under all three, under the library's own choice, and on the one-buffer in-place kernel (generic_kernel<1024, 1, true>,
the route of rows whose two buffers do not fit; FDOCT_GENERIC_INPLACE_ABOVE=81920 takes it for every row of which a CU holds one).  gpurun -- python tools/bench_generic_threads.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch
from fdoct_amd import DTYPE_U16, Config, Reconstructor, synth, capi
W, M, N, D = %(shape)r
H, nframes = 64, 16
cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M)
r = Reconstructor(cfg)
r.set_background(synth.make_background(W))
r.set_jit(False)
r.set_plan(-2, False)
fr = torch.from_numpy(synth.make_frames(0, 2, W, H).view(np.int16)).cuda().repeat(nframes // 2, 1, 1).contiguous()
out = torch.empty((nframes, H, D), dtype=torch.float32, device="cuda")
def go(n):
    for _ in range(n):
        r.process_device(fr.data_ptr(), DTYPE_U16, nframes, W * 2, None, out.data_ptr())
    r.synchronize()
go(3)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 0.6:
    go(5); n += 5
dt = (time.perf_counter() - t0) / n
assert r.last_kernel() == capi.KERNEL_GENERIC, r.last_kernel()
print("%%.4g" %% (nframes * H / dt))
"""
SHAPES = [(4096, 8, 32768, 2048), (4096, 4, 16384, 2048), (2048, 4, 8192, 1024), (1536, 4, 6144, 1024), (1000, 4, 4000, 500), (2048, 1, 6000, 3000), (160, 4, 2560, 320), (4000, 1, 4000, 2000)]
for shape in SHAPES:
    res = []
    for nt in ("", "256", "512", "1024", "inplace"):
        env = dict(os.environ)
        env.pop("FDOCT_GENERIC_THREADS", None)
        env.pop("FDOCT_GENERIC_INPLACE_ABOVE", None)
        env.pop("FDOCT_GENERIC_RADIX16", None)
        if nt == "inplace":   # the one-buffer kernel (1024 threads, radix-16 passes) wherever a CU holds one two-buffer row only
            env["FDOCT_GENERIC_INPLACE_ABOVE"] = str(80 * 1024)
        elif nt:
            env["FDOCT_GENERIC_THREADS"] = nt
            env["FDOCT_GENERIC_RADIX16"] = "0"   # (a plan with radix-16 passes runs on the 1024-thread kernels whatever is forced)
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "shape": shape}], env=env, capture_output=True, text=True, timeout=300)
        res.append(p.stdout.strip().splitlines()[-1] if p.returncode == 0 and p.stdout.strip() else "fail: " + p.stderr[-200:])
    print("%5d x%d -> %5d, %4d bins: library's choice %9s | 256 threads %9s | 512 %9s | 1024 %9s | one buffer in place %9s  A-scans/s" % (*shape, *res))
