"""BASELINE.json's configurations run EXACTLY as stated (full frame sizes, full averaging depth) through the C ABI:

  C1  1024-pt x 512-line frames (the "plumbing" configuration)         (here, both layouts and both variants)
  C2  2048-pt x 1000-line u16 frames                                   (test_gpu_parity.py::test_size_independent_properties_full_size)
  C3  2048-pt x 1000-line, dispersion phase multiply + Hann window     (here)
  C4  4096-pt x 2048-line, averaging N = 16 frames                     (here; BscanFFT.cpp:1193-1222)

The oracle finishes a few rows in seconds, not 2048 x 16 of them, so each test checks (a) oracle parity on the first and the
last 8 A-scans of the first and the last output B-scan -- rows are independent (1-row background, no whole-frame
normalisation), so the oracle run on those rows alone IS the reference result for them -- and (b) size-independent
properties over the whole output: finite, analytic peak bin (wangOCTrec4.m:200-202) on every row, and the averaging
identities of main:1193-1222 (a group of identical frames reproduces the single-frame B-scan; frame order inside a group
does not matter).
"""
import numpy as np
import pytest

import helpers
from fdoct_amd import Config, Reconstructor, synth

pytestmark = pytest.mark.gpu


def _rows_parity(cfg_full, frames, yb, got_bscan, got_db, groups, row_slices, what, **kw):
    """Oracle parity of output B-scans `groups` on the given row slices (cfg_full.averages input frames per group)."""
    A = cfg_full.averages
    worst = 0.0
    for g in groups:
        for sl in row_slices:
            sub = np.ascontiguousarray(frames[g * A:(g + 1) * A, sl, :])
            ocfg = Config(width=cfg_full.width, height=sub.shape[1], numfftpoints=cfg_full.numfftpoints,
                          numdisplaypoints=cfg_full.numdisplaypoints, averages=A)
            mag_o, _, db_o = helpers.oracle_reference(ocfg, sub, yb, **kw)
            tag = "%s group %d rows %s" % (what, g, sl)
            worst = max(worst, helpers.check_mag(got_bscan[g:g + 1, sl], mag_o, tag))
            helpers.check_db(got_db[g:g + 1, sl], np.transpose(db_o, (0, 2, 1)), mag_o, tag)
    return worst


def _tall_frames(f0, n, W, H, base_rows=128):
    """n distinct frames of H rows built from base_rows-row synthetic frames (rolled copies stacked), cheap to generate;
    also returns the reflector depth (um) of every row for the analytic peak check."""
    base = synth.make_frames(f0, n, W, base_rows)
    reps = (H + base_rows - 1) // base_rows
    frames = np.empty((n, reps * base_rows, W), np.uint16)
    depth = np.empty((n, reps * base_rows))
    for i in range(n):
        ls1, _ = synth.frame_depths_um(f0 + i, base_rows)
        for k in range(reps):
            frames[i, k * base_rows:(k + 1) * base_rows] = np.roll(base[i], 5 * k, axis=0)
            depth[i, k * base_rows:(k + 1) * base_rows] = np.roll(ls1, 5 * k)
    return frames[:, :H].copy(), depth[:, :H].copy()


def test_c1_as_stated_1024pt_512_lines():
    """BASELINE configs[0] at its stated size: 1024-pt x 512-line frames, numfftpoints 1024, 512 depth bins -- through the
    C ABI in the library's row-major layout and in the reference's D x H (main:1220; round 6: written by the chain itself on
    the 512-point plan), in the BscanFFT.cpp and the BscanFFTsim.cpp variant (sim:845 normalises every frame, sim:949 adds 1e-6).
    Oracle parity on the first and last 8 A-scans of the first and last B-scan, the properties over all 3 x 512 rows."""
    from fdoct_amd import LAYOUT_TRANSPOSED, VARIANT_SIM
    W, H, N, D = 1024, 512, 1024, 512
    frames, depth = _tall_frames(20, 3, W, H)
    yb = synth.make_background(W)
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    r = Reconstructor(cfg)
    r.set_background(yb)
    b, d = r.process(frames)
    bt, dt = r.process(frames, layout=LAYOUT_TRANSPOSED)
    # halving frame and background leaves the B-scan unchanged (the (y - yp)/yb step)
    r.set_background(yb.astype(np.float64) * 0.5)
    even = frames // 2 * 2
    b_half, _ = r.process(even.astype(np.float32) * 0.5)
    r.set_background(yb)
    b_even, _ = r.process(even)
    r.close()
    assert b.shape == (3, H, D) and bt.shape == (3, D, H) and np.isfinite(b).all() and np.isfinite(d).all()
    np.testing.assert_array_equal(bt, np.transpose(b, (0, 2, 1)))      # the D x H images are the row-major ones, bit for bit
    np.testing.assert_array_equal(dt, np.transpose(d, (0, 2, 1)))
    _rows_parity(cfg, frames, yb, b, d, (0, 2), (slice(0, 8), slice(H - 8, H)), "C1 as stated")
    helpers.check_mag(b_half, b_even, "C1 scale invariance")
    want = synth.expected_peak_bin(depth, W)
    got = b[:, :, 3:].argmax(axis=2) + 3
    assert np.abs(got - want).max() <= 2.5, np.abs(got - want).max()
    # the sim variant: whole-frame min-max normalisation couples the rows of a frame, so the oracle takes whole frames
    scfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, variant=VARIANT_SIM)
    r = Reconstructor(scfg)
    r.set_background(yb)
    bs, ds = r.process(frames[:1])
    bst, dst = r.process(frames[:1], layout=LAYOUT_TRANSPOSED)
    r.close()
    np.testing.assert_array_equal(bst, np.transpose(bs, (0, 2, 1)))
    np.testing.assert_array_equal(dst, np.transpose(ds, (0, 2, 1)))
    mag_o, _, db_o = helpers.oracle_reference(scfg, frames[:1], yb)
    helpers.check_mag(bs, mag_o, "C1 as stated, sim variant")
    helpers.check_db(ds, np.transpose(db_o, (0, 2, 1)), mag_o, "C1 as stated, sim variant")


def test_c3_as_stated_2048pt_1000_lines_phase_and_hann():
    """BASELINE configs[2]: 2048-pt x 1000-line frames, Hann apodization, dispersion-compensation phase multiply."""
    W, H, N, D = 2048, 1000, 2048, 1024
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames, depth = _tall_frames(40, 3, W, H, base_rows=125)
    yb = synth.make_background(W)
    win, ph = synth.hann_window(W), synth.dispersion_phase(N)
    r = Reconstructor(cfg)
    r.set_background(yb)
    r.set_window(win)
    r.set_dispersion_phase(ph)
    b, d = r.process(frames)
    # linearity in the frame/background scale (the (y - yp)/yb step): halving both leaves the B-scan unchanged
    r.set_background(yb.astype(np.float64) * 0.5)
    even = frames // 2 * 2
    b_half, _ = r.process(even.astype(np.float32) * 0.5)
    r.set_background(yb)
    b_even, _ = r.process(even)
    r.close()
    assert b.shape == (3, H, D) and np.isfinite(b).all() and np.isfinite(d).all()
    _rows_parity(cfg, frames, yb, b, d, (0, 2), (slice(0, 8), slice(H - 8, H)), "C3 as stated", window=win, phase=ph)
    helpers.check_mag(b_half, b_even, "C3 scale invariance")
    # without the phase the peak is a single bin; the cubic phase term broadens it, so compare the energy centroid
    # loosely: the strongest bin stays within the chirp's spread of the analytic depth bin
    want = synth.expected_peak_bin(depth, W)
    got = b[:, :, 3:].argmax(axis=2) + 3
    assert np.abs(got - want).max() <= 40, np.abs(got - want).max()


def test_c4_as_stated_4096pt_2048_lines_average_16():
    """BASELINE configs[3]: 4096-pt x 2048-line high-res spectrometer, averaging N = 16 frames (main:1193-1222)."""
    W, H, N, D, A = 4096, 2048, 4096, 2048, 16
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
    distinct, depth = _tall_frames(60, A, W, H)
    order2 = (np.arange(A) * 5 + 3) % A                       # second group: the same frames in another order
    frames = np.concatenate([distinct, distinct[order2]])     # 2 groups x 16 frames x 2048 x 4096 u16 (512 MiB)
    yb = synth.make_background(W)
    r = Reconstructor(cfg)
    r.set_background(yb)
    b, d = r.process(frames)
    assert b.shape == (2, H, D) and np.isfinite(b).all() and np.isfinite(d).all()
    _rows_parity(cfg, frames, yb, b, d, (0, 1), (slice(0, 8), slice(H - 8, H)), "C4 as stated")
    # frame order inside a group does not matter (accumulation, main:1197): group 1 is a permutation of group 0
    helpers.check_mag(b[1:2], b[0:1], "C4 permutation of the averaged frames")
    # 16 copies of one frame average to that frame's own B-scan (A = 1 handle)
    same = np.ascontiguousarray(np.broadcast_to(distinct[3], (A, H, W)))
    b_same, _ = r.process(same)
    r.close()
    r1 = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=1))
    r1.set_background(yb)
    b_one, _ = r1.process(distinct[3:4])
    r1.close()
    helpers.check_mag(b_same, b_one, "C4 sixteen identical frames")
    # analytic KAT over all 2 x 2048 averaged rows: the 16 frames of a group put their reflector peaks at 16 different
    # depths (each weighs 1/16 in the mean), but every frame's two reflectors are 150 um apart, so their mutual
    # interference term lands on the same bin n*150um/deltax in all of them and is the strongest bin of the average
    want = synth.expected_peak_bin(150.0, W)
    got = b[:, :, 3:].argmax(axis=2) + 3
    assert np.abs(got - want).max() <= 2.5, (np.abs(got - want).max(), want)
    # and each frame's own reflector peak is still there with its 1/16 weight: around frame 3's depth bin the average
    # holds at least 0.9/16 of what that frame gives when processed alone
    own = np.rint(synth.expected_peak_bin(depth[3], W)).astype(int)
    rows = np.arange(H)
    near_one = np.stack([b_one[0][rows, np.clip(own + k, 0, D - 1)] for k in (-2, -1, 0, 1, 2)]).max(axis=0)
    near_avg = np.stack([b[0][rows, np.clip(own + k, 0, D - 1)] for k in (-2, -1, 0, 1, 2)]).max(axis=0)
    assert (near_avg >= near_one / A * 0.9).all()


def test_state_blob_and_window_on_the_shipped_ini_shape():
    """with increasefftpointsmultiplier > 1 (build/BscanFFT.ini: W = 640, N = 2560, M = 4) the window still has W
    entries (it is applied before the zero-pad, main:1142/1146): fdoct_set_window / fdoct_get_window take W, and an
    exported state blob -- the multi-GPU set-up broadcast -- imports into a second handle and reproduces the results."""
    from fdoct_amd import FdoctError
    W, H, N, D, M, A = 640, 6, 2560, 320, 4, 2
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A,
                 lambdamin=840.5e-9, lambdamax=859.5e-9)
    frames, yb = synth.make_frames(2, 2 * A, W, H), synth.make_background(W)
    r0 = Reconstructor(cfg)
    r0.set_background(yb)
    assert r0.get_window().shape == (W,)
    b_builtin, _ = r0.process(frames)
    hann = synth.hann_window(W)
    r0.set_window(hann)
    np.testing.assert_array_equal(r0.get_window(), hann)
    with pytest.raises(FdoctError):
        r0.set_window(np.ones(W * M))          # the zero-padded length is not a window length
    b0, d0 = r0.process(frames)
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, window=hann)
    helpers.check_mag(b0, mag_o, "shipped-ini shape, Hann window")
    assert np.abs(b0 - b_builtin).max() > 0
    blob = r0.export_state()
    r1 = Reconstructor(cfg)
    r1.import_state(blob)
    b1, d1 = r1.process(frames)
    np.testing.assert_array_equal(b1, b0)
    np.testing.assert_array_equal(d1, d0)
    np.testing.assert_array_equal(r1.get_window(), hann)
    # malformed blobs are rejected and leave the handle as it was
    bad = blob.copy()
    bad.view(np.int32)[9] = 7                   # phase float count: neither 0 nor 2N
    with pytest.raises(FdoctError):
        r1.import_state(bad)
    bad = blob.copy()
    bad.view(np.int32)[5] = 1                   # another zero-pad multiplier
    with pytest.raises(FdoctError):
        r1.import_state(bad)
    with pytest.raises(FdoctError):
        r1.import_state(blob[:len(blob) - 8])   # truncated
    nyb = W
    off = 48 + (nyb + W + N) * 8                # header, background, window, fractionalk -> nearestkindex
    bad = blob.copy()
    bad[off:off + 4].view(np.int32)[0] = W * M + 5
    with pytest.raises(FdoctError):
        r1.import_state(bad)
    b2, _ = r1.process(frames)
    np.testing.assert_array_equal(b2, b0)
    r0.close()
    r1.close()


def test_median_7_is_8_bit_only():
    """cv::medianBlur accepts ksize 7 for CV_8U only (BscanFFT.cpp:955 would throw on a 16-bit frame): rejected loudly."""
    from fdoct_amd import FdoctError
    cfg = Config(width=64, height=8, numfftpoints=64, numdisplaypoints=32)
    r = Reconstructor(cfg)
    raw16 = (np.arange(8 * 64, dtype=np.uint16).reshape(1, 8, 64) * 37) % 1000
    with pytest.raises(FdoctError):
        r.frontend(raw16, mediann=7)
    out = r.frontend(raw16.astype(np.uint8), mediann=7)
    assert out.shape == (1, 8, 64)
    r.close()


def test_get_ylin_matches_the_oracles_data_ylin():
    """fdoct_get_ylin: the k-linear rows the resample stage leaves in HBM (staged mode) against the oracle's data_ylin --
    the quantity BscanFFTsim.cpp:901-909 dumps as "debugzpaddedlin" for the Octave cross-check -- on the real path and,
    with the dispersion phasors divided out, on the complex path."""
    import oracle_lib as orc
    from fdoct_amd import FdoctError
    W, H, N, D = 2048, 12, 2048, 1024
    frames, yb = synth.make_frames(21, 2, W, H), synth.make_background(W)
    p = orc.make_params(W, H, N, D)
    idx, frac = orc.tables(W, 1, N, synth.LAMBDAMIN, synth.LAMBDAMAX)
    want = np.stack([orc.frame_to_mag(p, f.astype(np.float64), yb.astype(np.float64), None, orc.barthann(W), idx, frac,
                                      want_ylin=True)[1] for f in frames])               # (2, H, N) doubles
    r = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
    r.set_background(yb)
    r.process(frames)
    with pytest.raises(FdoctError):
        r.get_ylin(0, 1)                       # the last run was the fused chain: nothing was materialised
    r.set_staged(True)
    b_staged, _ = r.process(frames)
    got = r.get_ylin(0, 2 * H).reshape(2, H, N)
    scale = np.abs(want).max(axis=-1, keepdims=True)
    assert (np.abs(got - want) <= 4e-6 * scale).all(), (np.abs(got - want) / scale).max()
    assert (got[..., 0] == 0).all() and (got[..., N - 1] == 0).all()      # never written by the reference: defined 0
    np.testing.assert_array_equal(r.get_ylin(H + 3, 2), got[1, 3:5])
    with pytest.raises(FdoctError):
        r.get_ylin(2 * H - 1, 2)               # past the end of the batch
    # complex path: data_ylin = stored value * conj(phasor)
    r.set_dispersion_phase(synth.dispersion_phase(N))
    r.process(frames)
    got_c = r.get_ylin(0, 2 * H).reshape(2, H, N)
    assert (np.abs(got_c - 0.5 * 2.0 * want) <= 6e-6 * scale).all()
    r.close()


@pytest.mark.parametrize("W,M,N,D", [(160, 4, 2560, 320), (640, 4, 2560, 320), (720, 4, 2880, 360), (640, 1, 640, 320),
                                     (320, 4, 2560, 320), (160, 4, 2560, 1280), (640, 4, 2560, 700), (640, 1, 640, 77)])
def test_wave_per_row_kernel_on_the_shipped_configurations(W, M, N, D):
    """fdoct_wave.hip: the configurations of build/*.ini (numfftpoints 2560 / 2880 / 640, zero-pad x4 or x1) run with one
    wave per A-scan.  Against the oracle (u8 with a full-frame background and averaging; u16; f32), and against the
    workgroup-per-row kernel of fdoct_generic.hip (set_plan(-2)), which computes the same steps."""
    rng = np.random.default_rng(W + D)
    H, A = 7, 3
    lam = dict(lambdamin=840.5e-9, lambdamax=859.5e-9)
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A, **lam)
    frames16 = synth.make_frames(5, 2 * A, max(W, 64), H)[:, :, :W].copy()
    yb16 = synth.make_background(max(W, 64))[:W].astype(np.float64) + 10.0
    cases = [("u16", frames16, yb16),
             ("u8 2-D background", (frames16 >> 8).astype(np.uint8), (yb16 / 256.0 + 1.0)[None, :] * (0.8 + 0.4 * rng.random((H, 1))))]
    for name, frames, yb in cases:
        r = Reconstructor(cfg)
        r.set_background(yb)
        b, d = r.process(frames)
        r.set_plan(-2)                       # the workgroup-per-row kernel
        bg, dg = r.process(frames)
        r.close()
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb)
        what = "wave kernel W=%d M=%d N=%d D=%d %s" % (W, M, N, D, name)
        helpers.check_mag(b, mag_o, what)
        helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)
        helpers.check_same(b, bg, what + " vs generic kernel", scale=0.35)   # (two different DFT factorisations: 0.2 is for equal arithmetic)
        assert np.abs(b - bg).max() > 0 or D < 8, "both runs took the same kernel?"
    # f32 samples (what the moving-average pre-stage hands over) with movavgn on
    cfg_m = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=1, movavgn=2, **lam)
    r = Reconstructor(cfg_m)
    r.set_background(yb16)
    b, d = r.process(frames16[:2])
    r.close()
    mag_o, _, db_o = helpers.oracle_reference(cfg_m, frames16[:2], yb16)
    helpers.check_mag(b, mag_o, "wave kernel after smoothmovavg")


@pytest.mark.parametrize("W", [192, 240, 256, 288, 384, 400, 432, 480, 512, 576, 768, 800, 864, 960, 1024, 1152, 1200, 1280])
def test_wave_per_row_kernel_on_other_regions_of_interest(W):
    """The widths an operator gets by editing the ini's ROI / binvalue (build/BscanFFT.ini:9-12, 25-26) with the shipped
    numfftpoints 2560 and zero-pad x4 (FDOCT_WAVE_SHAPES_EXTRA): 8- and 16-bit frames on the wave-per-row kernel against the
    oracle and against the workgroup-per-row kernel; f32 frames and deeper displays of the same shapes get their kernel compiled
    at run time (tests/test_gpu_jit.py), the workgroup-per-row kernel where that is switched off."""
    M, N, D, H, A = 4, 2560, 320, 5, 2
    lam = dict(lambdamin=840.5e-9, lambdamax=859.5e-9)
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A, **lam)
    frames16 = synth.make_frames(7, 2 * A, W, H)
    yb16 = synth.make_background(W).astype(np.float64) + 10.0
    for name, frames, yb in (("u16", frames16, yb16), ("u8", (frames16 >> 8).astype(np.uint8), yb16 / 256.0 + 1.0)):
        r = Reconstructor(cfg)
        r.set_background(yb)
        b, d = r.process(frames)
        r.set_plan(-2)
        bg, _ = r.process(frames)
        r.close()
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb)
        what = "wave kernel W=%d %s" % (W, name)
        helpers.check_mag(b, mag_o, what)
        helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)
        helpers.check_same(b, bg, what + " vs generic kernel", scale=0.35)   # (two different DFT factorisations: 0.2 is for equal arithmetic)
        assert np.abs(b - bg).max() > 0, "both runs took the same kernel?"
    # off the compiled variants: f32 samples, and a display deeper than 512 bins -> same results from 8-/16-bit and float samples
    cfg2 = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=700, increasefftpointsmultiplier=M, averages=A, **lam)
    r = Reconstructor(cfg2)
    r.set_background(yb16)
    b2, _ = r.process(frames16)
    bf, _ = r.process(frames16.astype(np.float32))
    r.close()
    np.testing.assert_array_equal(b2, bf)
    mag_o, _, _ = helpers.oracle_reference(cfg2, frames16, yb16)
    helpers.check_mag(b2, mag_o, "W=%d, 700 depth bins (generic kernel)" % W)


@pytest.mark.parametrize("W,M,N,D,dt", [(160, 4, 2560, 320, np.uint8), (640, 4, 2560, 320, np.uint16), (640, 1, 640, 320, np.uint8)])
def test_wave_per_row_kernel_persistent_row_loop(W, M, N, D, dt):
    """the wave-per-row kernel's persistent loop (o += stride, with the prefetch of row o + stride issued in the last
    averaging pass) only runs when there are more output rows than waves in flight.  One workgroup (fdoct_set_launch blocks = 1)
    over 2 x 41 output A-scans of 3 averaged frames makes every wave stride through several rows, the last round ragged:
    against the oracle and against the workgroup-per-row kernel."""
    H, A, G = 41, 3, 2
    lam = dict(lambdamin=840.5e-9, lambdamax=859.5e-9)
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A, **lam)
    frames = synth.make_frames(9, G * A, max(W, 64), H, dtype=dt)[:, :, :W].copy()
    yb = synth.make_background(max(W, 64), dtype=dt)[:W].astype(np.float64) + 3.0
    r = Reconstructor(cfg)
    r.set_background(yb)
    b_all, d_all = r.process(frames)            # the whole grid
    r.set_launch(0, 1)                          # one workgroup: waves stride over the rows
    b, d = r.process(frames)
    r.set_launch(128, 1)                        # two waves only: ~41 rows per wave
    b2, d2 = r.process(frames)
    r.set_launch(0, 0)
    r.set_plan(-2)                              # the workgroup-per-row kernel
    bg, _ = r.process(frames)
    r.close()
    np.testing.assert_array_equal(b, b_all)
    np.testing.assert_array_equal(d, d_all)
    np.testing.assert_array_equal(b2, b_all)
    np.testing.assert_array_equal(d2, d_all)
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb)
    what = "wave kernel, one workgroup, W=%d M=%d N=%d" % (W, M, N)
    helpers.check_mag(b, mag_o, what)
    helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)
    helpers.check_same(b, bg, what + " vs generic kernel", scale=0.35)   # (two different DFT factorisations: 0.2 is for equal arithmetic)


@pytest.mark.parametrize("W,N,D,M,phase_on", [(700, 1400, 700, 1, False), (1001, 2002, 900, 1, False), (509, 1018, 509, 1, False),
                                              (945, 2047, 1000, 1, False), (640, 1778, 400, 2, False), (512, 1022, 1022, 1, True)])
def test_any_numfftpoints_like_cv_dft(W, N, D, M, phase_on):
    """cv::dft takes any length (BscanFFT.cpp:1185); so does the path: lengths with a prime factor above 5 (1400 = 2^3 5^2 7,
    2002 = 2*7*11*13, 1018 = 2*509, 2047 = 23*89 odd, 1778 = 2*7*127 with the zero-pad, 1022 = 2*7*73 on the complex
    path) run as Bluestein's algorithm inside the any-configuration kernel.  Against the oracle, whose DFT is a direct
    O(N^2)-free mixed-radix / naive transform for these lengths."""
    H, A = 5, 2
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A)
    frames = synth.make_frames(13, 2 * A, max(W, 64), H)[:, :, :W].copy()
    yb = synth.make_background(max(W, 64))[:W].astype(np.float64) + 10.0
    kw = {"phase": synth.dispersion_phase(N)} if phase_on else {}
    r = Reconstructor(cfg)
    r.set_background(yb)
    if phase_on:
        r.set_dispersion_phase(kw["phase"])
    b, d = r.process(frames)
    r.close()
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, **kw)
    helpers.check_mag(b, mag_o, "Bluestein N=%d" % N)
    helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, "Bluestein N=%d dB" % N)
