"""A GPU comparison point, not part of the product: the vendor FFT library (rocFFT / hipFFT behind torch.fft) on
the rows of BASELINE's C2, next to this library's own FFT stage and fused chain on the same card.
  (a) c2c: torch.fft.ifft of complex64 rows of 2048 points -- what the reference asks cv::dft for (main:1185: a complex row
      whose imaginary plane is zeros);
  (b) r2c: torch.fft.rfft of float32 rows of 2048 points -- the transform a real row needs (1025 bins);
  (c) r2c + |.| + crop to 1024 bins + 20 ln / 2.303: the FFT stage's whole job with library calls (separate kernels, HBM in between);
  (d) this library: the FFT stage of fdoct_set_staged and the fused chain (HIP events via fdoct_get_timing / torch events).
usage (gpurun): python3 tools/rocfft_point.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fdoct_amd import DTYPE_U16, LAYOUT_ROWMAJOR, Config, Reconstructor, synth  # noqa: E402

dev = torch.device("cuda:0")
N, D, H = 2048, 1024, 1000
F = 64                      # frames per call: 64 000 rows (c2c: 1 GB in, 1 GB out)
rows = F * H


def timed(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


x = torch.randn(rows, N, device=dev, dtype=torch.float32)
xc = torch.complex(x, torch.zeros_like(x))
t_c2c = timed(lambda: torch.fft.ifft(xc, dim=1, norm="forward"))
t_r2c = timed(lambda: torch.fft.rfft(x, dim=1))


def chain():
    s = torch.fft.rfft(x, dim=1)[:, :D]
    return (20.0 / 2.303) * torch.log(s.abs() + 1e-5)


t_chain = timed(chain)
del xc
torch.cuda.empty_cache()
print("C2 rows (2048 points, %d rows per call), 1 x MI355X; rates in rows = A-scans per second" % rows)
print("  rocFFT c2c 2048 (complex64 in and out, 32 KiB per row):        %8.3f ms  %7.1f M rows/s  %5.2f TB/s of its own traffic" % (t_c2c, rows / t_c2c / 1e3, rows * 32768 / t_c2c / 1e9))
print("  rocFFT r2c 2048 (float32 in, 1025 complex out, 16.2 KB per row): %8.3f ms  %7.1f M rows/s  %5.2f TB/s" % (t_r2c, rows / t_r2c / 1e3, rows * (8192 + 8200) / t_r2c / 1e9))
print("  r2c + abs + crop + log with library calls (the FFT stage's job): %8.3f ms  %7.1f M rows/s" % (t_chain, rows / t_chain / 1e3))

cfg = Config(width=N, height=H, numfftpoints=N, numdisplaypoints=D)
r = Reconstructor(cfg)
r.set_background(synth.make_background(N))
frames = torch.from_numpy(synth.make_frames(3, 8, N, H).view(np.int16)).to(dev).repeat(F // 8, 1, 1).contiguous()
out = torch.empty((F, H, D), dtype=torch.float32, device=dev)
stream = torch.cuda.Stream(device=dev)
r.set_stream(stream.cuda_stream)


def ours():
    r.process_device(frames.data_ptr(), DTYPE_U16, F, N * 2, None, out.data_ptr(), LAYOUT_ROWMAJOR)


def timed_s(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_fused = timed_s(ours)
r.set_staged(True)
r.set_timing(True)          # per-stage device times from the library's own events
fs = []
for i in range(15):
    ours()
    t_st = r.timing()
    if i >= 5:
        fs.append(t_st["fft_stage_ms"])
t_st = {"fft_stage_ms": sum(fs) / len(fs)}
r.close()
print("  this library, FFT stage of the staged mode (float32 k-linear row in, 1024 dB bins out: 12 KiB per row): %8.3f ms  %7.1f M rows/s"
      % (t_st["fft_stage_ms"], rows / t_st["fft_stage_ms"] / 1e3))
print("  this library, the whole fused chain (u16 samples in, dB out: division, mean, window, resample, IDFT, |.|, log): %8.3f ms  %7.1f M rows/s"
      % (t_fused, rows / t_fused / 1e3))
