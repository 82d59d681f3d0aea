"""Shared test helpers: the parity tolerance of SURVEY.md 8(d) and the oracle-side
reference computation for a fdoct_amd.Config."""
import numpy as np

import oracle_lib as orc
from fdoct_amd import VARIANT_SIM, Config

# |gpu - cpu| <= RTOL*|cpu| + ATOL_ROWMAX*max_row|cpu|   (linear magnitudes, float path)
RTOL = 1e-4
ATOL_ROWMAX = 1e-6
DB_SLACK = 2e-4         # dB, on top of the bound implied by the linear tolerance
TRUTH_LIMIT = 0.5       # of the tolerance: what the HIP result may be away from the fp64 evaluation of the chain (check_truth)


# Every linear image oracle_reference returns is remembered here with the arguments that made it, so that check_mag can put
# the SAME call through the fp64 evaluation of the chain (oracle_truth) and hold the HIP result to check_truth as well --
# every parity test that compares against an oracle output is adjudicated against the exact chain without naming it.
_ORACLE_CALLS = {}      # id(mag array) -> (the array, cfg, frames, yb, kwargs, [cached truth])
TRUTH_LOG = []          # (test id, what, |gpu - truth| / tol, |f32 oracle - truth| / tol): conftest.py writes the table
TRUTH_MAX_CELLS = 1 << 26   # skip the second oracle run for huge inputs (none in the suites today)


def _truth_for(cpu):
    e = _ORACLE_CALLS.get(id(cpu))
    if e is None or e[0] is not cpu:
        return None
    if e[5][0] is None:
        e[5][0] = oracle_reference(e[1], e[2], e[3], truth=1, _register=False, **e[4])[0]
    return e[5][0]


def check_mag(gpu, cpu, what=""):
    """gpu, cpu: (..., H, D) linear magnitudes in row-major layout.  When cpu is an output of oracle_reference the HIP result
    is held to check_truth against the fp64 evaluation of the same call too."""
    truth = _truth_for(cpu) if isinstance(cpu, np.ndarray) else None
    cpu_arr = cpu
    gpu = np.asarray(gpu, np.float64)
    cpu = np.asarray(cpu, np.float64)
    assert gpu.shape == cpu.shape or truth is not None and gpu.shape == cpu[:gpu.shape[0]].shape, (gpu.shape, cpu.shape)
    if truth is not None and np.isfinite(gpu).all():
        import os
        g, o = truth_ratios(gpu, truth[:gpu.shape[0]], cpu_arr[:gpu.shape[0]])
        TRUTH_LOG.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], what, g, o))
        assert g <= max(TRUTH_LIMIT, o), "%s: |gpu - truth| / tol = %.3g exceeds max(%.2g, the f32 oracle's own %.3g)" % (what, g, TRUTH_LIMIT, o)
    assert gpu.shape == cpu.shape, (gpu.shape, cpu.shape)
    rowmax = np.abs(cpu).max(axis=-1, keepdims=True)
    tol = RTOL * np.abs(cpu) + ATOL_ROWMAX * rowmax
    err = np.abs(gpu - cpu)
    worst = (err / np.maximum(tol, 1e-300)).max()
    assert np.isfinite(gpu).all(), what + ": non-finite output"
    assert worst <= 1.0, "%s: worst error/tolerance = %.3g (max abs err %.3g, rowmax %.3g)" % (
        what, worst, err.max(), rowmax.max())
    return worst


def mag_ratio(gpu, cpu, rowmax=None):
    """Per-bin |gpu - cpu| / tolerance of check_mag (rowmax: the A-scan's peak when the displayed bins do not hold it)."""
    gpu = np.asarray(gpu, np.float64)
    cpu = np.asarray(cpu, np.float64)
    if rowmax is None:
        rowmax = np.abs(cpu).max(axis=-1, keepdims=True)
    return np.abs(gpu - cpu) / np.maximum(RTOL * np.abs(cpu) + ATOL_ROWMAX * rowmax, 1e-300)


def db_ratio(gpu_db, cpu_db, cpu_mag, rowmax=None):
    """Per-bin |gpu_db - cpu_db| / tolerance of check_db."""
    gpu_db = np.asarray(gpu_db, np.float64)
    cpu_db = np.asarray(cpu_db, np.float64)
    cpu_mag = np.abs(np.asarray(cpu_mag, np.float64))
    if rowmax is None:
        rowmax = cpu_mag.max(axis=-1, keepdims=True)
    tol_lin = RTOL * cpu_mag + ATOL_ROWMAX * rowmax
    tol_db = (20.0 / 2.303) * np.log1p(tol_lin / np.maximum(cpu_mag, 1e-300)) + DB_SLACK
    if tol_db.shape[-1] > 4:
        tol_db[..., 0] = np.maximum(tol_db[..., 0], tol_db[..., 4])
        tol_db[..., 1] = np.maximum(tol_db[..., 1], tol_db[..., 4])
    return np.abs(gpu_db - cpu_db) / tol_db


def check_same(a, b, what="", scale=0.2):
    """Two kernels of this library on the same input (fast-path option vs the any-option kernel): the same arithmetic up
    to the order of a few f32 roundings (fma vs mul+add, f32 vs f64 row mean), so they must agree within `scale` of the
    oracle tolerance: |a - b| <= scale * (1e-4 |b| + 1e-6 rowmax).  (scale = 0.2 is what the arithmetic needs: at 0.1 the
    normalised variants fail, at 0.05 the full-frame background one -- measured in round 3.  Indexing / prefetch / hand-over
    bugs are caught bit for bit elsewhere: staged = fused, transposed store = row-major, one workgroup = whole grid.)"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    rowmax = np.abs(b).max(axis=-1, keepdims=True)
    tol = scale * (RTOL * np.abs(b) + ATOL_ROWMAX * rowmax)
    worst = (np.abs(a - b) / np.maximum(tol, 1e-300)).max()
    assert np.isfinite(a).all(), what + ": non-finite output"
    assert worst <= 1.0, "%s: kernels disagree, worst difference / (%.2g x tolerance) = %.3g" % (what, scale, worst)
    return worst


def check_db(gpu_db, cpu_db, cpu_mag, what=""):
    """dB parity.  The bound is the one the linear tolerance implies: with tol the allowed linear
    error of a bin, |d dB| <= (20/2.303)*ln(1 + tol/|cpu|) + DB_SLACK (the slack covers the
    hardware log2).  SURVEY 8(d)'s flat "1e-3 dB above 1e-4*rowmax" cannot hold next to its own
    linear bound (the 1e-6*rowmax term alone is 1% of such a bin = 0.09 dB), so the implied
    bound is used; it is <= 2e-3 dB wherever the magnitude exceeds 1e-2 of the row maximum."""
    gpu_db = np.asarray(gpu_db, np.float64)
    cpu_db = np.asarray(cpu_db, np.float64)
    cpu_mag = np.abs(np.asarray(cpu_mag, np.float64))
    rowmax = cpu_mag.max(axis=-1, keepdims=True)
    tol_lin = RTOL * cpu_mag + ATOL_ROWMAX * rowmax
    tol_db = (20.0 / 2.303) * np.log1p(tol_lin / np.maximum(cpu_mag, 1e-300)) + DB_SLACK
    err = np.abs(gpu_db - cpu_db)
    # depth bins 0,1 of the dB image are copies of bin 4 (DC mask, main:1237-1238): bin 4's bound applies to them
    # (cpu_mag[0], cpu_mag[1] are the unmasked DC bins, usually far larger than bin 4)
    if tol_db.shape[-1] > 4:
        tol_db[..., 0] = np.maximum(tol_db[..., 0], tol_db[..., 4])
        tol_db[..., 1] = np.maximum(tol_db[..., 1], tol_db[..., 4])
    worst = (err / tol_db).max()
    assert np.isfinite(gpu_db).all(), what + ": non-finite dB"
    assert worst <= 1.0, "%s: worst dB error/tolerance %.3g (max abs %.3g dB)" % (what, worst, err.max())
    strong = cpu_mag > 1e-2 * rowmax
    strong[..., :2] = False
    if strong.any():
        assert err[strong].max() <= 2.2e-3, "%s: %.3g dB on a strong bin" % (what, err[strong].max())
    return worst


def db_flat_pass_rate(gpu_db, cpu_db, cpu_mag, limit_db=1e-3, floor=1e-4):
    """SURVEY 8(d)'s flat criterion, reported next to the implied bound check_db enforces: the fraction of dB bins whose
    magnitude exceeds `floor` x the row maximum (DC-masked bins 0, 1 excluded) that agree within `limit_db`, and the
    largest dB difference among them.  Returns (fraction, max_abs_db, bins_counted)."""
    gpu_db = np.asarray(gpu_db, np.float64)
    cpu_db = np.asarray(cpu_db, np.float64)
    cpu_mag = np.abs(np.asarray(cpu_mag, np.float64))
    rowmax = cpu_mag.max(axis=-1, keepdims=True)
    sel = cpu_mag > floor * rowmax
    sel[..., :2] = False
    if not sel.any():
        return 1.0, 0.0, 0
    err = np.abs(gpu_db - cpu_db)[sel]
    return float((err <= limit_db).mean()), float(err.max()), int(sel.sum())


def oracle_reference(cfg: Config, frames, yb, yp=None, yd=None, window=None, table=None, phase=None, threads=1, bandpass=0, truth=0,
                     _register=True):
    """Runs the CPU restatement for cfg.  Returns (mag (G,H,D) = bscan without transpose incl. eps,
    bscan (G,D,H), bscandb (G,D,H)).  truth=1: the fp64 evaluation of the same chain (oracle_truth)."""
    W, H, N, D = cfg.width, cfg.height, cfg.numfftpoints, cfg.numdisplaypoints
    M = cfg.increasefftpointsmultiplier
    sim = cfg.variant == VARIANT_SIM
    p = orc.make_params(W, H, N, D, M, rowwisenormalize=cfg.rowwisenormalize,
                        donotnormalize=0 if sim else cfg.donotnormalize, movavgn=cfg.movavgn, bandpass=bandpass, threads=threads, truth=truth)
    win = orc.barthann(W) if window is None else np.asarray(window, np.float64)
    if table is None:
        idx, frac = orc.tables(W, M, N, cfg.lambdamin, cfg.lambdamax)
    else:
        idx, frac = table
    eps = 1e-6 if sim else 1e-5
    frames = np.asarray(frames)
    if frames.dtype != np.uint16:
        # the oracle driver takes u16; u8 frames embed exactly
        assert frames.dtype == np.uint8
        frames = frames.astype(np.uint16)
    mag, bscan, db = orc.process_u16(p, cfg.averages, eps, frames, yb, yp, win, idx, frac, yd=yd, phase=phase, sim_copy=sim)
    mag = mag + eps
    if _register and not truth and frames.size <= TRUTH_MAX_CELLS:
        if len(_ORACLE_CALLS) > 64:
            _ORACLE_CALLS.clear()
        _ORACLE_CALLS[id(mag)] = (mag, cfg, frames, yb, dict(yp=yp, yd=yd, window=window, table=table, phase=phase, bandpass=bandpass), [None])
    return mag, bscan, db


def oracle_truth(cfg: Config, frames, yb, **kw):
    """The reference's MATHEMATICS on these inputs: the oracle's chain with every float step (zero-pad DFTs main:209-242,
    narrowing main:1181, cv::dft main:1185, magnitude main:1190) in double.  Same return value as oracle_reference."""
    return oracle_reference(cfg, frames, yb, truth=1, **kw)



def truth_ratios(gpu, truth, oracle_f32=None, rowmax=None):
    """(worst |gpu - truth| / tol, worst |oracle_f32 - truth| / tol or None), tol = 1e-4 |truth| + 1e-6 rowmax(truth)."""
    truth = np.asarray(truth, np.float64)
    g = float(mag_ratio(gpu, truth, rowmax).max())
    o = None if oracle_f32 is None else float(mag_ratio(oracle_f32, truth, rowmax).max())
    return g, o


def check_truth(gpu, truth, oracle_f32, what="", limit=TRUTH_LIMIT):
    """Adjudication against the exact chain instead of the restatement's float roundings: the HIP result
    must be no farther from truth than max(the f32 oracle's own distance, `limit` x the tolerance).  Within 0.5 x the
    tolerance of the exact value, ANY correctly rounded float evaluation of the reference chain -- cv::dft with whatever radix
    decomposition the installed OpenCV picks -- that is itself within 0.5 lies within the tolerance of the HIP result.  Where
    the float chain itself cannot hold 0.5 (tiny outputs of large intermediates) the HIP path is held to the float chain's own
    distance.  Returns (gpu ratio, f32-oracle ratio)."""
    g, o = truth_ratios(gpu, truth, oracle_f32)
    assert np.isfinite(np.asarray(gpu)).all(), what + ": non-finite output"
    assert g <= max(limit, o), "%s: |gpu - truth| / tol = %.3g exceeds max(%.2g, the f32 oracle's own %.3g)" % (what, g, limit, o)
    return g, o


EPS32 = 2.0 ** -24   # unit roundoff of the reference's DFT type (main:1181-1185)


def truth_row_scales(cfg: Config, frames, yb, **kw):
    """Two scales of every A-scan taken from the fp64 evaluation of the chain over the WHOLE magI row (all numfftpoints bins,
    main:1190 -- the crop to numdisplaypoints comes after it, main:1192):
      peak   max_row|magI|: the row maximum SURVEY 8(d)'s tolerance names.  The fixed tests take the maximum of the DISPLAYED bins
             instead, which is the same thing whenever the display holds the A-scan's peak and stricter when it does not.
      floor  eps32 * sqrt(log2 N) * ||x||_2, x the DFT's input row (by Parseval from the magnitudes): the a-priori size of the
             per-bin error of ANY float narrowing + float DFT of that row (main:1181, 1185).  REPORTED next to a verdict, not part
             of the tolerance: in every case the sweeps have met it is 1-7 % of the 1e-6 * peak term.
    Returns (peak, floor), each (G, H, 1)."""
    import dataclasses
    N = cfg.numfftpoints
    full = oracle_truth(dataclasses.replace(cfg, numdisplaypoints=N), frames, yb, **kw)[0]
    peak = np.abs(full).max(axis=-1, keepdims=True)
    x2 = np.sqrt((full * full).sum(axis=-1, keepdims=True) / N)
    return peak, EPS32 * np.sqrt(np.log2(max(N, 2))) * x2


def truth_ratios_scaled(gpu, truth, oracle_f32, peak, floor=0.0):
    """(|gpu - truth| / tol', |oracle_f32 - truth| / tol') per case with tol' = 1e-4 |truth| + 1e-6 peak (+ floor), peak = the
    maximum of the whole magI row (truth_row_scales).  The sweeps pass no floor: SURVEY 8(d)'s tolerance as it is written."""
    truth = np.asarray(truth, np.float64)
    tol = RTOL * np.abs(truth) + ATOL_ROWMAX * peak + floor
    g = float((np.abs(np.asarray(gpu, np.float64) - truth) / tol).max())
    o = float((np.abs(np.asarray(oracle_f32, np.float64) - truth) / tol).max())
    return g, o


def db_ratios_scaled(gpu_db, truth_db, oracle_db, truth_mag, peak, floor=0.0):
    """The same for the dB images: the bound the linear tolerance tol' implies, + DB_SLACK; bins 0, 1 carry bin 4's bound."""
    tm = np.abs(np.asarray(truth_mag, np.float64))
    tol_lin = RTOL * tm + ATOL_ROWMAX * peak + floor
    tol_db = (20.0 / 2.303) * np.log1p(tol_lin / np.maximum(tm, 1e-300)) + DB_SLACK
    if tol_db.shape[-1] > 4:
        tol_db[..., 0] = np.maximum(tol_db[..., 0], tol_db[..., 4])
        tol_db[..., 1] = np.maximum(tol_db[..., 1], tol_db[..., 4])
    g = float((np.abs(np.asarray(gpu_db, np.float64) - truth_db) / tol_db).max())
    o = float((np.abs(np.asarray(oracle_db, np.float64) - truth_db) / tol_db).max())
    return g, o
