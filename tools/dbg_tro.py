"""Debug aid: where the fused transposed store differs from the row-major result (frames, rows, depth bins)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdoct_amd import Config, Reconstructor, synth, LAYOUT_TRANSPOSED
W, H, N, D = 2048, 1000, 2048, 1024
nframes = 6
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 3
both = int(sys.argv[2]) if len(sys.argv) > 2 else 1
frames = synth.make_frames(11, nframes, W, H)
r = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
r.set_background(synth.make_background(W))
bscan, db = r.process(frames)
r.set_launch(0, blocks)
for rep in range(3):
    bt, dt = r.process(frames, want_bscan=bool(both), layout=LAYOUT_TRANSPOSED)
    bad = np.argwhere(dt != np.transpose(db, (0, 2, 1)))
    print("rep", rep, "mismatches", len(bad))
    if len(bad):
        g, d, rr = bad[:, 0], bad[:, 1], bad[:, 2]
        for gg in np.unique(g):
            rows = np.unique(rr[g == gg])
            for row in rows:
                dd = np.sort(d[(g == gg) & (rr == row)])
                # what does the wrong data equal? another row of the row-major result?
                got = dt[gg, :, row]
                src = None
                for g2 in range(nframes):
                    m = (db[g2][:, dd[0]] == got[dd[0]]) & (db[g2][:, dd[-1]] == got[dd[-1]])
                    if m.any():
                        src = (g2, int(np.argmax(m)))
                        break
                print("  frame %d row %d (tile %d, row-in-tile %d): %d bins, %d..%d; data equals row %s" % (gg, row, row // 32, row % 32, len(dd), dd[0], dd[-1], src))
r.close()
