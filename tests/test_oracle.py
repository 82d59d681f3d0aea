"""CPU tests of the oracle (oracle/fdoct_oracle.c), the restatement of BscanFFT.cpp:615-698,
936-944, 1123-1240.  PARITY UNPINNED against the reference itself (it stores no outputs and
cannot be built here); the oracle is pinned by the reference's input fixtures, an analytic
known-answer test on them, numpy cross-checks of every stage and frozen golden outputs."""
import hashlib
import json
import os

import numpy as np
import pytest

import helpers
import oracle_lib as orc
from fdoct_amd import VARIANT_MAIN, VARIANT_SIM, Config, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture(name):
    return np.fromfile(os.path.join(GOLD, name + "_u16_96x128.bin"), "<u2").reshape(96, 128)


def test_input_fixtures_match_manifest_and_generator():
    """imgi.png / backg.png (as raw u16) are intact, and wangOCTimg.m:41-49 as restated in
    fdoct_amd.synth reproduces them to +-1 count (so the synthetic generator is the reference's)."""
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    imgi, backg = _fixture("imgi"), _fixture("backg")
    assert hashlib.sha256(imgi.tobytes()).hexdigest() == man["imgi_u16_96x128.bin"]["sha256"]
    assert hashlib.sha256(backg.tobytes()).hexdigest() == man["backg_u16_96x128.bin"]["sha256"]
    gi, gb = synth.reference_fixture_rows(128, 96)
    assert np.abs(np.rint(gi * 65535) - imgi.astype(np.int64)).max() <= 1
    assert np.abs(np.rint(gb * 65535) - backg.astype(np.int64)).max() <= 1
    assert (backg == backg[0]).all()  # every background row is the same spectrum


def test_tables_properties_and_numpy():
    """A0, main:615-698."""
    for W, M, N in [(128, 1, 1024), (2048, 1, 2048), (640, 4, 2560), (1024, 1, 512)]:
        lmin, lmax = 816e-9, 884e-9
        idx, frac, k, kl, dk = orc.tables(W, M, N, lmin, lmax, debug=True)
        MW = M * W
        dl = (lmax - lmin) / W
        lam = lmin + np.arange(MW) * dl / M
        kk = 2 * 3.141592653589793 / lam
        np.testing.assert_array_equal(k, kk)
        kmin, kmax = 2 * 3.141592653589793 / (lmax - dl), 2 * 3.141592653589793 / lmin
        np.testing.assert_array_equal(kl, kmin + (np.arange(N) + 1) * ((kmax - kmin) / N))
        # nearestkindex = first i with k[i] < klinear[f]; k decreases so idx is non-increasing in f
        want = np.array([np.argmax(kk < x) if (kk < x).any() else 0 for x in kl], np.int32)
        np.testing.assert_array_equal(idx, want)
        assert (np.diff(idx) <= 0).all() and idx.min() >= 1 and idx.max() <= MW - 1
        assert (frac > 0).all() and (frac <= 1.0 + 1e-12).all()
        d = np.empty(MW)
        d[1:] = kk[:-1] - kk[1:]
        d[0] = d[1]
        np.testing.assert_array_equal(frac, (kl - kk[idx]) / d[idx])


def test_window_float_division_quirk():
    """A1, main:936-944: nn/NN is a float division."""
    for W in (128, 2048, 1280):
        w = orc.barthann(W)
        r = (np.arange(W, dtype=np.float32) / np.float32(W - 1)).astype(np.float64)
        want = 0.62 - 0.48 * np.abs(r - 0.5) + 0.38 * np.cos(2 * 3.141592653589793 * (r - 0.5))
        np.testing.assert_allclose(w, want, rtol=0, atol=1e-15)
        assert abs(w[0]) < 1e-6 and abs(w.max() - 1.0) < 3e-3  # even W: no sample sits on the peak


def test_normalize_and_movavg():
    rng = np.random.default_rng(0)
    y = rng.random((5, 37)) * 1000
    n = orc.normalize_minmax(y)
    np.testing.assert_allclose(n, (y - y.min()) / (y.max() - y.min()), atol=1e-12)
    nr = orc.normalizerows(y)
    np.testing.assert_allclose(nr, (y - y.min(1, keepdims=True)) / np.ptp(y, axis=1, keepdims=True), atol=1e-12)
    flat = np.full((2, 9), 3.0)
    assert (orc.normalize_minmax(flat) == 0).all()  # max-min < eps -> scale 0 (cv::normalize)
    # smoothmovavg main:247-304: truncated taps replaced by the centre sample, centre counted twice
    k = 2
    out = orc.smoothmovavg(y, k)
    want = np.empty_like(y)
    for r in range(y.shape[0]):
        for j in range(y.shape[1]):
            s = 0.0
            for d in range(-k, k + 1):
                jj = j + d
                s += y[r, jj] if 0 <= jj < y.shape[1] else y[r, j]
            want[r, j] = (s + y[r, j]) / 2 / (k + 1)
    np.testing.assert_allclose(out, want, rtol=1e-14)


@pytest.mark.parametrize("N", [16, 320, 1024, 2048, 2560])
def test_dft_models_vs_numpy(N):
    """A7: cv::dft(DFT_INVERSE) = +i exponent, unscaled; f32 model within 1e-6 * rowmax, f64 within 1e-12."""
    rng = np.random.default_rng(N)
    z = (rng.standard_normal((3, N)) + 1j * rng.standard_normal((3, N)))
    ref = np.fft.ifft(z, axis=1) * N
    got64 = orc.dft_rows_f64(z, inverse=True)
    assert np.abs(got64 - ref).max() <= 1e-11 * np.abs(ref).max()
    got32 = orc.dft_rows_f32(z.astype(np.complex64), inverse=True)
    assert np.abs(got32 - ref).max() <= 2e-6 * np.abs(ref).max()
    fwd = orc.dft_rows_f64(z, inverse=False, scale=True)
    assert np.abs(fwd - np.fft.fft(z, axis=1) / N).max() <= 1e-12 * np.abs(z).max()


def test_zeropad_matches_spectral_model():
    """A4, main:180-245: forward DFT/W, fftshift, pad, ifftshift, real-output inverse (which reads only
    bins 0..n/2, so the original Nyquist bin is dropped)."""
    rng = np.random.default_rng(3)
    W, M = 64, 4
    y = rng.standard_normal((3, W))
    out = orc.zeropadrowwise(y, M)
    F = np.fft.fft(y, axis=1) / W
    n = np.arange(M * W)
    want = np.real(F[:, :1]) + 2 * np.real(sum(F[:, k:k + 1] * np.exp(2j * np.pi * k * n / (M * W)) for k in range(1, W // 2)))
    assert np.abs(out - want).max() <= 5e-6 * np.abs(want).max()


@pytest.mark.parametrize("N", [320, 643, 1018, 1283, 1400, 2002, 2047, 2560, 2880, 4097])
def test_dft_f32_model_vs_torch_pocketfft(N):
    """A third DFT next to the oracle's own float transform and numpy's double one: torch.fft.ifft on float32 CPU tensors
    (pocketfft in single precision -- an implementation that shares nothing with the oracle's radix-2 / mixed-radix / Bluestein
    code and, unlike numpy's, computes in the reference's arithmetic type, cv::dft on CV_32F, main:1185).  Non-power-of-two
    lengths, the shipped ini's 2560 and 2880, and lengths with large prime factors (643 and 1283 are prime: the padded spectra
    of odd widths; 1018 = 2 * 509, 2002 = 2 * 7 * 11 * 13, 2047 = 23 * 89, 4097 = 17 * 241).  Both float transforms must sit within
    float rounding of the double one, and of each other."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(N)
    z = (rng.standard_normal((4, N)) + 1j * rng.standard_normal((4, N))).astype(np.complex64)
    ref64 = np.fft.ifft(z.astype(np.complex128), axis=1) * N
    got_t = torch.fft.ifft(torch.from_numpy(z), dim=1, norm="forward").numpy()      # "forward" norm: the inverse is unscaled, as cv::dft's
    assert got_t.dtype == np.complex64
    got_o = orc.dft_rows_f32(z, inverse=True)
    scale = np.abs(ref64).max()
    e_t, e_o, e_to = np.abs(got_t - ref64).max() / scale, np.abs(got_o - ref64).max() / scale, np.abs(got_t - got_o).max() / scale
    assert e_t <= 2e-6 and e_o <= 2e-6 and e_to <= 3e-6, (N, e_t, e_o, e_to)
    fwd_t = torch.fft.fft(torch.from_numpy(z), dim=1, norm="forward").numpy()        # DFT_SCALE forward transform (main:211)
    fwd_o = orc.dft_rows_f32(z, inverse=False, scale=True)
    assert np.abs(fwd_t - fwd_o).max() <= 3e-6 * np.abs(z).max(), N


def _zeropad_numpy(y, M, dtype=np.float32):
    """main:180-245 restated with numpy, operation by operation, for any width: dft/W; swap the two halves of width cols/2
    (an odd last column stays, main:215-227); floor((MW - W)/2) zero columns either side (main:229); swap the halves of the
    padded spectrum (main:233-239); inverse real-output DFT of THAT length -- np.fft.irfft reads bins 0..n/2 and drops the
    imaginary parts of bins 0 and n/2 exactly like cv::dft's CCS reading (main:241); the Mat it returns has W + 2 pad columns."""
    y = np.atleast_2d(y)
    H, W = y.shape
    F = np.fft.fft(y.astype(dtype), axis=1) / W
    cx = W // 2
    sh = F.copy()
    sh[:, :cx], sh[:, cx:2 * cx] = F[:, cx:2 * cx], F[:, :cx]
    pad = (M * W - W) // 2
    zp = np.zeros((H, W + 2 * pad), complex)
    zp[:, pad:pad + W] = sh
    n = zp.shape[1]
    cz = n // 2
    g = zp.copy()
    g[:, :cz], g[:, cz:2 * cz] = zp[:, cz:2 * cz], zp[:, :cz]
    return np.fft.irfft(g[:, :n // 2 + 1], n, axis=1) * n


@pytest.mark.parametrize("W,M", [(64, 4), (63, 3), (63, 4), (161, 4), (161, 2), (45, 3), (100, 3), (25, 8)])
def test_zeropad_any_width_like_the_reference(W, M):
    """Odd widths (a ROI of an odd number of columns): the reference's fftshift leaves the last column of the spectrum in place
    and, with an even multiplier, its padded spectrum -- so its inverse transform and the row it returns -- is M W - 1 long
    (main:215-241).  The oracle follows it; the column M W - 1 the reference would read out of bounds is 0."""
    rng = np.random.default_rng(W * 10 + M)
    y = rng.standard_normal((3, W)) + 5.0
    out = orc.zeropadrowwise(y, M)
    want = _zeropad_numpy(y, M)
    n = want.shape[1]
    assert n == W + 2 * ((M * W - W) // 2) and out.shape[1] == M * W
    assert np.abs(out[:, :n] - want).max() <= 5e-6 * np.abs(want).max()
    assert (out[:, n:] == 0).all()


def test_frame_pipeline_against_numpy_restatement():
    """A2..A8 re-derived in numpy (float64 elementwise, np.fft for the IDFT) on seeded frames."""
    W, H, N, D = 256, 6, 512, 200
    rng = np.random.default_rng(7)
    x = rng.integers(100, 60000, (H, W)).astype(np.float64)
    yb = rng.integers(20000, 65000, W).astype(np.float64)
    yp = 50.0 * rng.random((H, W))
    idx, frac = orc.tables(W, 1, N, 816e-9, 884e-9)
    win = orc.barthann(W)
    p = orc.make_params(W, H, N, D)
    mag, ylin = orc.frame_to_mag(p, x, yb, yp, win, idx, frac, want_ylin=True)
    y = (x - yp) / yb[None]
    y = (y - y.mean(1, keepdims=True)) * win[None]
    sl = np.empty_like(y)
    sl[:, 1:] = y[:, 1:] - y[:, :-1]
    sl[:, 0] = sl[:, 1]
    lin = np.zeros((H, N))
    q = np.arange(1, N - 1)
    lin[:, q] = y[:, idx[q]] + frac[idx[q]] * sl[:, idx[q]]   # fractionalk[nearestkindex[q]]: the reference's quirk
    np.testing.assert_allclose(ylin, lin, rtol=1e-13, atol=1e-13)
    want = np.abs(np.fft.ifft(lin.astype(np.float32), axis=1) * N)
    assert np.abs(mag - want).max() <= 2e-6 * want.max()


def test_truth_mode_is_the_chain_in_double():
    """orc_params.truth: the reference's chain with every float step (zero-pad DFTs, narrowing, cv::dft, magnitude) in double.
    Against numpy in double -- np.fft.ifft on the un-narrowed data_ylin, the zero-pad stage by the numpy restatement of
    main:180-245 fed doubles -- to 1e-12 of the row maximum, on a power-of-two, a 2^a 3^b 5^c and a prime numfftpoints, with and
    without the zero-pad (even and odd widths) and the dispersion phasors."""
    rng = np.random.default_rng(11)
    for W, M, N in [(256, 1, 512), (160, 4, 2560), (161, 4, 1283), (90, 3, 640), (128, 1, 127)]:
        H, D = 3, N // 2
        x = rng.integers(100, 60000, (H, W)).astype(np.float64)
        yb = rng.integers(20000, 65000, W).astype(np.float64)
        idx, frac = orc.tables(W, M, N, 816e-9, 884e-9)
        win = orc.barthann(W)
        for phase in (None, synth.dispersion_phase(N)):
            p = orc.make_params(W, H, N, D, M, truth=1)
            mag, ylin = orc.frame_to_mag(p, x, yb, None, win, idx, frac, phase=phase, want_ylin=True)
            assert mag.dtype == np.float64
            y = x / yb[None]
            y = (y - y.mean(1, keepdims=True)) * win[None]
            if M > 1:
                y = _zeropad_numpy(y, M, dtype=np.float64)
                if y.shape[1] < M * W:
                    y = np.concatenate([y, np.zeros((H, M * W - y.shape[1]))], axis=1)
            sl = np.empty_like(y)
            sl[:, 1:] = y[:, 1:] - y[:, :-1]
            sl[:, 0] = sl[:, 1]
            lin = np.zeros((H, N))
            q = np.arange(1, N - 1)
            fr = np.where(idx[q] < N, frac[np.minimum(idx[q], N - 1)], 0.0)
            lin[:, q] = y[:, idx[q]] + fr * sl[:, idx[q]]
            np.testing.assert_allclose(ylin, lin, rtol=1e-9, atol=1e-9 * np.abs(lin).max())
            z = lin if phase is None else lin * (phase[:, 0].astype(np.float64) + 1j * phase[:, 1].astype(np.float64))[None]
            want = np.abs(np.fft.ifft(z, axis=1) * N)
            assert np.abs(mag - want).max() <= 1e-12 * want.max(), (W, M, N, np.abs(mag - want).max() / want.max())


def test_f32_restatement_sits_well_inside_half_the_tolerance_of_truth():
    """What check_truth's 0.5 rests on: on the BASELINE shapes, the shipped ini shape and the reference's own fixture the f32
    restatement (fp64 elementwise, float DFTs) is within a small fraction of the tolerance of the chain evaluated in double --
    so a HIP result within 0.5 of truth is within the tolerance of it, and of any other correctly rounded float chain."""
    imgi, backg = _fixture("imgi"), _fixture("backg")
    worst = {}
    cases = [("fixture", Config(width=128, height=96, numfftpoints=1024, numdisplaypoints=512), imgi[None], backg.astype(np.float64), {})]
    for name, W, H, N, D, M, A, ph in [("C1", 1024, 8, 1024, 512, 1, 1, False), ("C2", 2048, 8, 2048, 1024, 1, 1, False),
                                       ("C3", 2048, 6, 2048, 1024, 1, 1, True), ("C4", 4096, 3, 4096, 2048, 1, 4, False),
                                       ("INI", 160, 8, 2560, 320, 4, 2, False)]:
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A)
        kw = dict(phase=synth.dispersion_phase(N), window=synth.hann_window(W)) if ph else {}
        cases.append((name, cfg, synth.make_frames(5, A, W, H), synth.make_background(W).astype(np.float64), kw))
    for name, cfg, fr, yb, kw in cases:
        o = helpers.oracle_reference(cfg, fr, yb, **kw)[0]
        t = helpers.oracle_truth(cfg, fr, yb, **kw)[0]
        worst[name] = helpers.truth_ratios(o, t, o)[1]
    assert max(worst.values()) <= 0.15, worst


def test_known_answer_reflector_depth():
    """Physics KAT on the reference's fixture (wangOCTimg.m: row ii has reflectors at ii um and ii+50 um;
    wangOCTrec4.m:200-202: bin pitch = pi/(kmax-kmin)): the strongest peak is the first reflector
    (reflectivity 0.5), the second reflector (0.25) is a local maximum near its bin."""
    imgi, backg = _fixture("imgi"), _fixture("backg")
    cfg = Config(width=128, height=96, numfftpoints=1024, numdisplaypoints=512, variant=VARIANT_MAIN)
    mag, _, _ = helpers.oracle_reference(cfg, imgi[None], backg.astype(np.float64))
    mag = mag[0]
    for row in (9, 19, 39, 69, 95):
        ii = row + 1
        b1 = synth.expected_peak_bin(ii, 128)
        b2 = synth.expected_peak_bin(ii + 50, 128)
        got = 3 + mag[row, 3:200].argmax()
        assert abs(got - b1) <= 1.5 + 0.05 * b1, (row, got, b1)
        lo, hi = int(b2) - 3, int(b2) + 4
        assert mag[row, lo:hi].max() > 3 * np.median(mag[row, 40:200]), (row, b2)


def test_golden_outputs_frozen():
    """The committed oracle outputs (tests/golden/make_golden.py) still come out of the oracle."""
    z = np.load(os.path.join(GOLD, "oracle_outputs.npz"))
    imgi, backg = _fixture("imgi"), _fixture("backg")
    cfg = Config(width=128, height=96, numfftpoints=1024, numdisplaypoints=512, variant=VARIANT_SIM)
    mag, _, db = helpers.oracle_reference(cfg, (imgi >> 8).astype(np.uint8)[None], (backg >> 8).astype(np.float64))
    np.testing.assert_allclose(mag, z["fixture_sim_u8__mag"], rtol=2e-6)
    np.testing.assert_allclose(db, z["fixture_sim_u8__db"], atol=2e-4)
    W, H, N, D = 2048, 8, 2048, 1024
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    mag, _, db = helpers.oracle_reference(cfg, synth.make_frames(100, 1, W, H), synth.make_background(W))
    np.testing.assert_allclose(mag, z["c2_8rows__mag"], rtol=2e-6)


def test_averaging_and_finish_semantics():
    """A9/A10, main:1193-1240: accumulate, /A, +1e-5, 20*ln/2.303 (not ln 10), depth rows 0,1 <- row 4."""
    W, H, N, D, A = 256, 4, 256, 64, 3
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
    frames = synth.make_frames(0, A, W, H)
    yb = synth.make_background(W)
    mag, bscan, db = helpers.oracle_reference(cfg, frames, yb)
    one = [helpers.oracle_reference(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D),
                                    frames[i:i + 1], yb)[0] - 1e-5 for i in range(A)]
    np.testing.assert_allclose(mag - 1e-5, sum(one) / A, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(bscan[0], mag[0].T, rtol=0, atol=0)
    want_db = 20.0 * np.log(bscan[0]) / 2.303
    want_db[1] = want_db[4]
    want_db[0] = want_db[4]
    np.testing.assert_allclose(db[0], want_db, rtol=1e-14)


def test_frontend_median_and_binning_models():
    """cv::medianBlur (replicate border) vs scipy, INTER_AREA integer binning vs numpy."""
    from scipy import ndimage
    rng = np.random.default_rng(11)
    img = rng.integers(0, 65536, (37, 53)).astype(np.uint16)
    for n in (3, 5, 7):
        np.testing.assert_array_equal(orc.median_blur(img, n), ndimage.median_filter(img, size=n, mode="nearest"))
    img = rng.integers(0, 65536, (48, 60)).astype(np.uint16)
    s = img.astype(np.int64).reshape(24, 2, 30, 2).sum(axis=(1, 3))
    np.testing.assert_array_equal(orc.resize_area(img, 2, 2), (s + 2) >> 2)
    s = img.astype(np.int64).reshape(16, 3, 15, 4).sum(axis=(1, 3))
    want = np.rint(s.astype(np.float32) * np.float32(1.0 / 12.0)).astype(np.uint16)   # rint = round half to even
    np.testing.assert_array_equal(orc.resize_area(img, 4, 3), want)


def test_display_chain_against_numpy():
    """main:1242-1255 restated with numpy: threshold, min-max to [0,1], x255, round-half-even."""
    rng = np.random.default_rng(5)
    db = rng.standard_normal((64, 50)) * 25.0 - 20.0
    for thr, clamp in ((-30.0, False), (-5.0, True)):
        t = np.maximum(db, thr)
        if clamp:
            t[5, 5] = 50.0
        s = 1.0 / (t.max() - t.min())
        want = np.clip(np.rint((t * s + (0.0 - t.min() * s)) * 255.0), 0, 255).astype(np.uint8)
        got = orc.display_u8(db, thr, clamp)
        np.testing.assert_array_equal(got, want)
        assert got.min() == 0 and got.max() == 255
    lut = rng.integers(0, 256, (256, 3)).astype(np.uint8)
    np.testing.assert_array_equal(orc.apply_lut(got, lut), lut[got])
    b, j = np.abs(rng.standard_normal((20, 30))), np.abs(rng.standard_normal((20, 30)))
    np.testing.assert_allclose(orc.lockin_db(b, j), 20.0 * np.log(np.maximum(b - j, 0) + 1e-3) / 2.303, rtol=1e-14)


def test_oracle_under_asan_and_ubsan():
    """the reference has latent memory bugs on this path (slopes allocated with swapped dimensions,
    main:626/632; data_ylin columns never written, main:1164), so the CPU restatement runs under -fsanitize=address,undefined:
    `make -C oracle asan`, then this file's own cases (and the Octave cross-check's) in a child interpreter with the
    sanitizer runtime preloaded and the instrumented library in place of libfdoct_oracle.so.  Any report aborts the child."""
    import shutil
    import subprocess
    import sys
    if os.environ.get("FDOCT_ORACLE_SO"):
        pytest.skip("already inside the sanitizer run")
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    libasan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan next to gcc")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir, "-s", "asan"])
    env = dict(os.environ, FDOCT_ORACLE_SO=os.path.join(odir, "libfdoct_oracle_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle.py"),
                        os.path.join(ROOT, "tests", "test_octave_crosscheck.py"), "-k", "not asan and not pocketfft"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "runtime error" not in tail and "AddressSanitizer" not in tail, tail
