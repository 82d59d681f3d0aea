"""PCIe-inclusive rate of the host-buffer entry point (fdoct_process): frames in host memory in, dB out to host
memory, per call -- pageable numpy buffers and pinned ones (fdoct_host_alloc).  Not the bench.py value (that one is
HBM-resident)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from fdoct_amd import Config, PinnedArray, Reconstructor, synth  # noqa: E402

W, H, N, D = 2048, 1000, 2048, 1024
NF = 64
rec = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
rec.set_background(synth.make_background(W))
frames = np.tile(synth.make_frames(0, 4, W, H), (NF // 4, 1, 1))  # 64 frames = 256 MiB


def rate(fr, out_db, label):
    rec.process(fr, want_bscan=False, out_db=out_db)
    t0 = time.perf_counter()
    for _ in range(5):
        _, db = rec.process(fr, want_bscan=False, out_db=out_db)
    dt = (time.perf_counter() - t0) / 5
    print("PCIe-inclusive, %s: %.2f M A-scans/s (%.1f ms per %d-frame call, %.1f GB/s of host traffic)" %
          (label, NF * H / dt / 1e6, dt * 1e3, NF, (fr.nbytes + NF * H * D * 4) / dt / 1e9))
    return db


print("host threads: os.cpu_count() %s, affinity %d" % (os.cpu_count(), len(os.sched_getaffinity(0))))
rec.set_host_staging(0)
ref = rate(frames, None, "pageable, a FRESH result array per call (its page faults included), the runtime's bounce copies (rounds 1-5)").copy()
rec.set_host_staging(-1)
got = rate(frames, None, "pageable, a FRESH result array per call, pinned staging slots, the library's default (%d copy threads)" % rec.host_staging_threads())
assert np.array_equal(got, ref), "staged and unstaged results differ"
# from here on the result array is the caller's and reused, as a cv::Mat in an acquisition loop is
keep = np.empty((NF, H, D), np.float32)
rec.set_host_staging(0)
got = rate(frames, keep, "pageable, the runtime's bounce copies (rounds 1-5)")
assert np.array_equal(got, ref)
for threads in [int(t) for t in os.environ.get("FDOCT_PCIE_THREADS", "1,2,4,8,16").split(",")]:
    rec.set_host_staging(threads)
    got = rate(frames, keep, "pageable, pinned staging slots, %2d copy thread(s)" % threads)
    assert np.array_equal(got, ref), "staged and unstaged results differ"
rec.set_host_staging(-1)
got = rate(frames, keep, "pageable, pinned staging slots, the library's default (%d copy threads)" % rec.host_staging_threads())
assert np.array_equal(got, ref), "staged and unstaged results differ"
pin_in = PinnedArray(frames.shape, frames.dtype)
pin_out = PinnedArray((NF, H, D), np.float32)
pin_in.array[...] = frames
got = rate(pin_in.array, pin_out.array, "pinned  ")
assert np.array_equal(got, ref), "pinned and pageable results differ"
rec.close()

# The drop-in call as BscanFFTsim.cpp would make it: ONE frame per call, host pointers, both images in
# the reference's D x H layout -- latency per call from pageable (cv::Mat) and from pinned buffers.
from fdoct_amd import LAYOUT_TRANSPOSED  # noqa: E402

rec = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
rec.set_background(synth.make_background(W))
one = frames[:1].copy()
pin1 = PinnedArray(one.shape, one.dtype)
pin1.array[...] = one
pb, pd = PinnedArray((1, D, H), np.float32), PinnedArray((1, D, H), np.float32)
for label, fr, ob, od in (("pageable", one, None, None), ("pinned  ", pin1.array, pb.array, pd.array)):
    for _ in range(5):
        rec.process(fr, layout=LAYOUT_TRANSPOSED, out_bscan=ob, out_db=od)
    t0 = time.perf_counter()
    for _ in range(50):
        rec.process(fr, layout=LAYOUT_TRANSPOSED, out_bscan=ob, out_db=od)
    dt = (time.perf_counter() - t0) / 50
    t = rec.timing()
    print("one 2048 x 1000 frame per call, bscan + bscandb in D x H, %s: %.3f ms per call (device part %.3f ms, kernel %.3f ms) = %.0f frames/s"
          % (label, dt * 1e3, t["process_ms"], t["kernel_ms"], 1.0 / dt))
rec.close()
