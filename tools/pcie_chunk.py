"""PCIe-inclusive rate of fdoct_process on pageable host buffers against the batch size and the pipeline's chunk size
(FDOCT_HOST_CHUNK_MB, read per call): where does a batch start to be worth chunking?  usage: gpurun -- python3 tools/pcie_chunk.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from fdoct_amd import Config, PinnedArray, Reconstructor, synth  # noqa: E402

W, H, N, D = 2048, 1000, 2048, 1024
CHUNKS = [int(c) for c in os.environ.get("FDOCT_PCIE_CHUNKS", "0,8,16,32").split(",")]
rec = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
rec.set_background(synth.make_background(W))
base = synth.make_frames(0, 4, W, H)
for nf in (4, 8, 16, 32, 64, 64, 32, 16, 8):
    frames = np.tile(base, ((nf + 3) // 4, 1, 1))[:nf].copy()
    keep = np.empty((nf, H, D), np.float32)
    pin_in, pin_out = PinnedArray(frames.shape, frames.dtype), PinnedArray((nf, H, D), np.float32)
    pin_in.array[...] = frames
    line = "%2d frames per call (%3d MB in, %3d MB out):" % (nf, frames.nbytes >> 20, keep.nbytes >> 20)
    for label, fr, out in (("pageable", frames, keep), ("pinned", pin_in.array, pin_out.array)):
        for mb in CHUNKS:
            os.environ["FDOCT_HOST_CHUNK_MB"] = str(mb)       # 0: the library's own choice
            rec.process(fr, want_bscan=False, out_db=out)
            reps = max(6, 256 // nf)
            t0 = time.perf_counter()
            for _ in range(reps):
                rec.process(fr, want_bscan=False, out_db=out)
            dt = (time.perf_counter() - t0) / reps
            line += "  %s %s %.2f M" % (label if mb == CHUNKS[0] else "", "%2d MB" % mb if mb else "library's choice", nf * H / dt / 1e6)
    print(line, flush=True)
    pin_in.free()
    pin_out.free()
rec.close()
