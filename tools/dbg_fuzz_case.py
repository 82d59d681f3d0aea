"""Re-runs a seeded fuzz sweep and, for the configurations whose description contains the given text, prints where the
largest parity errors sit (bin, magnitude relative to the row maximum, error / tolerance).
python tools/dbg_fuzz_case.py <seed> <count> <text> [jit_share big_share route_share weak_share tall_share dev_share reuse_share]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases  # noqa: E402
import helpers  # noqa: E402

seed, count, text = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
orig_mag, orig_db = helpers.check_mag, helpers.check_db


def check_mag(gpu, cpu, what=""):
    if text in what:
        gpu64, cpu64 = np.asarray(gpu, np.float64), np.asarray(cpu, np.float64)
        rowmax = np.abs(cpu64).max(axis=-1, keepdims=True)
        tol = helpers.RTOL * np.abs(cpu64) + helpers.ATOL_ROWMAX * rowmax
        r = np.abs(gpu64 - cpu64) / tol
        print("LINEAR", what)
        print("  ratio percentiles 50/90/99/99.9/max:", np.percentile(r, [50, 90, 99, 99.9, 100]).round(4))
        for idx in np.argsort(r.ravel())[-6:][::-1]:
            i = np.unravel_index(idx, r.shape)
            print("  at", i, "cpu %.6g gpu %.6g rowmax %.6g cpu/rowmax %.3g ratio %.3f" % (cpu64[i], gpu64[i], rowmax[i[:-1] + (0,)], cpu64[i] / rowmax[i[:-1] + (0,)], r[i]))
    return orig_mag(gpu, cpu, what)


def check_db(gpu_db, cpu_db, cpu_mag, what=""):
    if text in what:
        g, c, m = (np.asarray(x, np.float64) for x in (gpu_db, cpu_db, cpu_mag))
        m = np.abs(m)
        rowmax = m.max(axis=-1, keepdims=True)
        tol_lin = helpers.RTOL * m + helpers.ATOL_ROWMAX * rowmax
        tol_db = (20.0 / 2.303) * np.log1p(tol_lin / np.maximum(m, 1e-300)) + helpers.DB_SLACK
        r = np.abs(g - c) / tol_db
        print("DB", what)
        for idx in np.argsort(r.ravel())[-6:][::-1]:
            i = np.unravel_index(idx, r.shape)
            print("  at", i, "cpu_db %.6f gpu_db %.6f mag %.6g mag/rowmax %.3g tol_db %.3g ratio %.3f" % (c[i], g[i], m[i], m[i] / rowmax[i[:-1] + (0,)], tol_db[i], r[i]))
    return orig_db(gpu_db, cpu_db, cpu_mag, what)


helpers.check_mag, helpers.check_db = check_mag, check_db
if os.environ.get("DBG_FORCE_PREC"):     # keep both words of the reciprocal background on in every case (the stream of cases is unchanged)
    _R = fuzz_cases.Reconstructor
    _R.set_precise_division = lambda self, on: None
if os.environ.get("DBG_KERNEL"):
    _R2 = fuzz_cases.Reconstructor
    _proc = _R2.process

    def process(self, *a, **k):
        out = _proc(self, *a, **k)
        print("  kernel family", self.last_kernel(), self.describe_kernel() if hasattr(self, "describe_kernel") else "")
        return out
    _R2.process = process
fuzz_cases.run_sweep(seed, count, log=lambda s: print(s) if text in s else None, **dict(zip(("jit_share", "big_share", "route_share", "weak_share", "tall_share", "dev_share", "reuse_share"),
                                               (float(x) for x in sys.argv[4:11]))))
