"""GPU parity tests proper: the HIP path, called through the C ABI (libfdoct_hip.so via
fdoct_amd.Reconstructor), against the CPU oracle on the same seeded inputs.

Tolerance (float path): |gpu-cpu| <= 1e-4*|cpu| + 1e-6*max_row|cpu| on linear magnitudes
(helpers.check_mag).  dB (helpers.check_db): the bound that linear tolerance implies for the bin,
(20/2.303) ln(1 + tol/|cpu|) + 2e-4 dB, and <= 2.2e-3 dB on bins above 1 % of the row maximum.  SURVEY 8d's flat
"1e-3 dB where the magnitude exceeds 1e-4 of the row maximum" is REPORTED (helpers.db_flat_pass_rate, bench.py's
`parity` object), not enforced: it cannot hold next to the linear bound it comes with.
"""
import os

import numpy as np
import pytest

import helpers
import oracle_lib as orc
from fdoct_amd import (LAYOUT_TRANSPOSED, VARIANT_MAIN, VARIANT_SIM, Config, FdoctError, Reconstructor, synth)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(cfg, frames, yb, **kw):
    r = Reconstructor(cfg)
    r.set_background(yb)
    if kw.get("yp") is not None:
        r.set_pi_frame(kw["yp"])
    if kw.get("yd") is not None:
        r.set_dark(kw["yd"])
    if kw.get("window") is not None:
        r.set_window(kw["window"])
    if kw.get("phase") is not None:
        r.set_dispersion_phase(kw["phase"])
    bscan, db = r.process(frames)
    r.close()
    return bscan, db


def _parity(cfg, frames, yb, what, **kw):
    bscan, db = _run(cfg, frames, yb, **kw)
    mag_o, bscan_o, db_o = helpers.oracle_reference(cfg, frames, yb, **kw)
    w = helpers.check_mag(bscan, mag_o, what)
    db_o_rm = np.transpose(db_o, (0, 2, 1))
    helpers.check_db(db, db_o_rm, mag_o, what)
    return w


@pytest.mark.parametrize("W,H,N,D", [(2048, 40, 2048, 1024), (1024, 33, 1024, 512), (512, 17, 512, 256),
                                     (4096, 12, 4096, 2048), (128, 96, 1024, 512), (640, 10, 2048, 320)])
def test_chain_u16(W, H, N, D):
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames = synth.make_frames(3, 2, W, H)
    yb = synth.make_background(W)
    _parity(cfg, frames, yb, "chain W=%d N=%d" % (W, N))


def test_reference_fixture_sim_variant():
    """C1 plumbing: the reference's own saved frames (Matlab files/imgi.png, backg.png as raw u16),
    BscanFFTsim.cpp settings: 8-bit imread, whole-frame normalise, eps 1e-6."""
    imgi = np.fromfile(os.path.join(GOLD, "imgi_u16_96x128.bin"), np.uint16).reshape(96, 128)
    backg = np.fromfile(os.path.join(GOLD, "backg_u16_96x128.bin"), np.uint16).reshape(96, 128)
    img8 = (imgi >> 8).astype(np.uint8)   # cv::imread default converts 16-bit PNGs to 8 bit
    bg8 = (backg >> 8).astype(np.uint8)
    cfg = Config(width=128, height=96, numfftpoints=1024, numdisplaypoints=512, variant=VARIANT_SIM)
    _parity(cfg, img8[None], bg8.astype(np.float64), "fixture sim")
    cfg16 = Config(width=128, height=96, numfftpoints=1024, numdisplaypoints=512, variant=VARIANT_MAIN)
    _parity(cfg16, imgi[None], backg.astype(np.float64), "fixture main u16 2-D background")


def test_dispersion_phase_and_hann():
    """C3: Hann apodization + dispersion-compensation phase multiply (complex IDFT)."""
    W, H, N, D = 2048, 24, 2048, 1024
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames = synth.make_frames(11, 2, W, H)
    yb = synth.make_background(W)
    _parity(cfg, frames, yb, "C3", window=synth.hann_window(W), phase=synth.dispersion_phase(N))


def test_set_averages_at_run_time():
    """fdoct_set_averages: the same handle with A = 1, then 4, then 2 equals handles created with those values."""
    W, H, N, D = 2048, 9, 2048, 1024
    frames, yb = synth.make_frames(3, 8, W, H), synth.make_background(W)
    r = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
    r.set_background(yb)
    for A in (1, 4, 2):
        r.set_averages(A)
        b, d = r.process(frames)
        ref = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A))
        ref.set_background(yb)
        b0, d0 = ref.process(frames)
        ref.close()
        assert b.shape == (8 // A, H, D)
        np.testing.assert_array_equal(b, b0)
        np.testing.assert_array_equal(d, d0)
    with pytest.raises(FdoctError):
        r.set_averages(0)
    r.close()


@pytest.mark.parametrize("where", ["host frames", "device frames", "host frames, 480 MB of them"])
def test_sim_variant_with_averages_emits_the_last_frame_of_each_group(where):
    """BscanFFTsim.cpp with averages = A > 1 (sim:936-947): the accumulate is commented out, every frame's magnitudes are copied
    over the previous one's and what the else branch emits -- undivided, + 1e-6 -- is the LAST copy: frame A - 1 of every group
    of A.  The library runs the chain on those frames only (one strided gather, fdoct_capi.cpp::sim_last_frames, or a frame
    stride through the host pipeline for a batch worth chunking); against the
    oracle's orc_process_u16_sim, and bit-equal to the same frames handed over one by one with averages = 1; fdoct_set_averages
    changes the grouping at run time."""
    W, H, N, D, A = 2048, 24, 2048, 1024, 3
    G = 40 if "480 MB" in where else 4             # (a host batch the plain path would pipeline in chunks)
    if "480 MB" in where:
        H = 1000
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A, variant=VARIANT_SIM)
    frames = synth.make_frames(11, 4 * A, W, H)
    frames = np.concatenate([frames] * (G // 4))[: G * A]
    yb = synth.make_background(W).astype(np.float64) / 65535.0       # sim:845 normalises every frame to [0, 1]
    r = Reconstructor(cfg)
    r.set_background(yb)
    if where == "device frames":
        import torch
        d_fr = torch.from_numpy(frames.view(np.int16)).cuda()
        d_b = torch.empty((G, H, D), dtype=torch.float32, device="cuda")
        d_d = torch.empty_like(d_b)
        torch.cuda.synchronize()
        from fdoct_amd import DTYPE_U16, LAYOUT_ROWMAJOR
        r.process_device(d_fr.data_ptr(), DTYPE_U16, G * A, W * 2, d_b.data_ptr(), d_d.data_ptr(), LAYOUT_ROWMAJOR)
        r.synchronize()
        b, d = d_b.cpu().numpy(), d_d.cpu().numpy()
    else:
        b, d = r.process(frames)
        if "480 MB" in where:
            # a batch worth pipelining: the chunks read every A-th frame where it lies (a frame stride through the three-stream
            # pipeline, round 6) -- through the pinned staging slots (default) and through the runtime's own copies
            r.set_host_staging(0)
            b0, d0 = r.process(frames)
            np.testing.assert_array_equal(b0, b)
            np.testing.assert_array_equal(d0, d)
    assert b.shape == (G, H, D)
    one = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=1, variant=VARIANT_SIM))
    one.set_background(yb)
    b1, d1 = one.process(frames[A - 1::A])
    one.close()
    np.testing.assert_array_equal(b, b1)
    np.testing.assert_array_equal(d, d1)
    # (whole frames: the sim variant's min-max normalisation, sim:845, is over the frame)
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames[: 2 * A], yb, threads=8)
    mag_l, _, _ = helpers.oracle_reference(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, variant=VARIANT_SIM), frames[A - 1:2 * A:A], yb, threads=8)
    np.testing.assert_array_equal(mag_o, mag_l)      # the oracle's own grouping: the last frame, undivided
    helpers.check_mag(b[:2], mag_o, "sim variant, averages = %d, %s" % (A, where))
    helpers.check_db(d[:2], np.transpose(db_o, (0, 2, 1)), mag_o, "sim variant, averages = %d, %s" % (A, where))
    if where == "host frames":
        r.set_averages(2)
        b2, _ = r.process(frames[: 4 * 2])
        np.testing.assert_array_equal(b2, one_by_one(cfg, frames[1:8:2], yb))
        with pytest.raises(FdoctError):
            r.process(frames[:3])                     # not a multiple of the group
    r.close()


def one_by_one(cfg, frames, yb):
    one = Reconstructor(Config(width=cfg.width, height=cfg.height, numfftpoints=cfg.numfftpoints, numdisplaypoints=cfg.numdisplaypoints,
                               averages=1, variant=cfg.variant))
    one.set_background(yb)
    b, _ = one.process(frames)
    one.close()
    return b


def test_averaging():
    """C4 shape: averaging A frames per B-scan (main:1193-1222)."""
    W, H, N, D, A = 4096, 6, 4096, 2048, 4
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
    frames = synth.make_frames(5, 2 * A, W, H)
    _parity(cfg, frames, synth.make_background(W), "averaging")


def test_options_normalise_pi_dark():
    W, H, N, D = 1024, 20, 1024, 512
    rng = np.random.default_rng(5)
    frames = synth.make_frames(2, 2, W, H)
    yb = synth.make_background(W).astype(np.float64) + 50.0
    yp = 200.0 * rng.random((H, W))
    yd = 30.0 * rng.random(W)
    for kw in (dict(rowwisenormalize=1), dict(donotnormalize=0), dict()):
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, **kw)
        scale = 1.0 if not kw else 1.0 / 65535.0
        _parity(cfg, frames, yb * scale, "opts %s" % kw, yp=yp * scale, yd=yd)


def test_transposed_layout_matches_reference_bscan():
    W, H, N, D = 1024, 50, 1024, 300
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames = synth.make_frames(0, 1, W, H)
    yb = synth.make_background(W)
    r = Reconstructor(cfg)
    r.set_background(yb)
    bscan_t, db_t = r.process(frames, layout=LAYOUT_TRANSPOSED)
    bscan, db = r.process(frames)
    r.close()
    assert bscan_t.shape == (1, D, H)
    np.testing.assert_array_equal(bscan_t, np.transpose(bscan, (0, 2, 1)))
    np.testing.assert_array_equal(db_t, np.transpose(db, (0, 2, 1)))
    _, bscan_o, _ = helpers.oracle_reference(cfg, frames, yb)
    helpers.check_mag(np.transpose(bscan_t, (0, 2, 1)), np.transpose(bscan_o, (0, 2, 1)), "transposed")


@pytest.mark.parametrize("H,D,A,nframes,dt", [(1000, 1024, 1, 2, np.uint16), (16, 1024, 1, 1, np.uint16), (20, 1024, 1, 3, np.uint16),
                                               (4, 1024, 1, 1, np.uint16), (52, 700, 1, 2, np.uint16), (48, 512, 1, 1, np.uint16),
                                               (36, 1024, 2, 6, np.uint16), (40, 1000, 1, 2, np.uint8), (8, 5, 1, 1, np.uint16)])
def test_transposed_layout_shapes(H, D, A, nframes, dt):
    """The reference's D x H layout (main:1220) with one or both output arrays requested, on the C2 row length: bit-identical
    to the row-major output transposed on the host.  Heights that are not multiples of 16 or 32 (partial transpose tiles), a
    single tile, cropped and ragged depths, averaging, 8-bit samples."""
    W, N = 2048, 2048
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
    frames = synth.make_frames(3, nframes, W, H, dtype=dt)
    yb = synth.make_background(W, dtype=dt)
    r = Reconstructor(cfg)
    r.set_background(yb)
    bscan, db = r.process(frames)
    _, db_t = r.process(frames, want_bscan=False, layout=LAYOUT_TRANSPOSED)
    bscan_t, _ = r.process(frames, want_db=False, layout=LAYOUT_TRANSPOSED)
    bscan_t2, db_t2 = r.process(frames, layout=LAYOUT_TRANSPOSED)
    r.close()
    G = nframes // A
    assert db_t.shape == (G, D, H)
    np.testing.assert_array_equal(db_t, np.transpose(db, (0, 2, 1)))
    np.testing.assert_array_equal(bscan_t, np.transpose(bscan, (0, 2, 1)))
    np.testing.assert_array_equal(db_t2, db_t)
    np.testing.assert_array_equal(bscan_t2, bscan_t)
    mag_o, _, _ = helpers.oracle_reference(cfg, frames, yb)
    helpers.check_mag(np.transpose(bscan_t, (0, 2, 1)), mag_o, "transposed")


@pytest.mark.parametrize("blocks,A,dt", [(3, 1, np.uint16), (1, 1, np.uint16), (0, 1, np.uint16), (5, 2, np.uint8)])
def test_fused_transposed_store_many_tiles_per_workgroup(blocks, A, dt):
    """The chain writing the reference's D x H layout itself (fused_kernel's TRO instantiations): 2048-sample
    rows, 1000 rows per B-scan -- 62 tiles of 16 A-scans and one of 8 per B-scan -- with the launch restricted to a few
    workgroups so that every workgroup walks through a hundred tiles and goes round its ring of finished rows over and over
    (the hand-over protocol between the waves: rows into the ring, write-out steps claimed from it), one and both outputs (the
    write-out step takes the logarithm then), averaging, 8-bit samples.  Bit-identical to the row-major output transposed on the host,
    and device-resident frames through fdoct_process_async too."""
    import torch
    W, H, N, D = 2048, 1000, 2048, 1024
    nframes = 6
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
    frames = synth.make_frames(11, nframes, W, H, dtype=dt)
    yb = synth.make_background(W, dtype=dt)
    r = Reconstructor(cfg)
    r.set_background(yb)
    bscan, db = r.process(frames)
    r.set_launch(0, blocks)
    bscan_t, db_t = r.process(frames, layout=LAYOUT_TRANSPOSED)
    _, db_t1 = r.process(frames, want_bscan=False, layout=LAYOUT_TRANSPOSED)
    G = nframes // A
    assert db_t.shape == (G, D, H)
    np.testing.assert_array_equal(db_t, np.transpose(db, (0, 2, 1)))
    np.testing.assert_array_equal(bscan_t, np.transpose(bscan, (0, 2, 1)))
    np.testing.assert_array_equal(db_t1, db_t)
    # device-resident frames, outputs pre-filled with NaN: every element of the D x H images is written
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(frames.view(np.int16) if dt == np.uint16 else frames).to(dev)
    d_b = torch.full((G, D, H), float("nan"), dtype=torch.float32, device=dev)
    d_d = torch.full((G, D, H), float("nan"), dtype=torch.float32, device=dev)
    from fdoct_amd import DTYPE_U8, DTYPE_U16
    r.process_device(d_in.data_ptr(), DTYPE_U16 if dt == np.uint16 else DTYPE_U8, nframes, W * frames.itemsize, d_b.data_ptr(),
                     d_d.data_ptr(), LAYOUT_TRANSPOSED)
    r.synchronize()
    r.close()
    np.testing.assert_array_equal(d_b.cpu().numpy(), bscan_t)
    np.testing.assert_array_equal(d_d.cpu().numpy(), db_t)


@pytest.mark.parametrize("H,D,blocks", [(500, 1024, 1), (500, 1024, 3), (500, 512, 2), (1000, 512, 1), (1000, 512, 3), (36, 256, 1)])
def test_fused_transposed_store_tiles_that_complete_out_of_order(H, D, blocks):
    """Tiles of the transposed store need not complete in order: with a 4-row last tile of a B-scan
    (H mod 16 == 4) the rows of tile q + 1 wait for tile q - 1 only, and with the 40-slot ring (D <= 512) no row of tile q + 1
    waits for anything of tile q, so tile q + 1 can be complete while a row of tile q is still in flight.  Write-out steps
    are claimed from an IN-ORDER count of complete tiles (tro_publish), so that never hands out a step of a tile whose rows
    are not all in the ring.  Few workgroups, so each walks through many tiles; several repeats, because the interleaving
    is a matter of timing; bit-identical to the row-major images transposed on the host."""
    W, N = 2048, 2048
    nframes = 4
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames = synth.make_frames(17, nframes, W, H)
    yb = synth.make_background(W)
    r = Reconstructor(cfg)
    r.set_background(yb)
    bscan, db = r.process(frames)
    r.set_launch(0, blocks)
    for rep in range(4):
        bscan_t, db_t = r.process(frames, layout=LAYOUT_TRANSPOSED)
        from fdoct_amd.capi import KERNEL_FUSED_TRANSPOSED
        assert r.last_kernel() == KERNEL_FUSED_TRANSPOSED
        np.testing.assert_array_equal(bscan_t, np.transpose(bscan, (0, 2, 1)))
        np.testing.assert_array_equal(db_t, np.transpose(db, (0, 2, 1)))
    for threads in (128, 256):   # two and four waves per workgroup
        r.set_launch(threads, blocks)
        _, db_t = r.process(frames, want_bscan=False, layout=LAYOUT_TRANSPOSED)
        np.testing.assert_array_equal(db_t, np.transpose(db, (0, 2, 1)))
    r.close()


@pytest.mark.parametrize("H,D,A,dt,blocks", [(512, 512, 1, np.uint16, 0), (500, 512, 1, np.uint16, 2), (36, 256, 1, np.uint16, 1), (132, 512, 3, np.uint16, 3),
                                             (260, 512, 1, np.uint8, 1), (100, 64, 2, np.uint8, 2)])
def test_fused_transposed_store_on_the_512_point_plan(H, D, A, dt, blocks):
    """Round 6: the chain writes the reference's D x H layout (main:1220) itself on the 512-point plan too -- C1, 1024 samples ->
    numfftpoints 1024, 16 lanes per row and FOUR rows per wave: tiles of 16 rows are owned by groups of four waves, the finished
    rows wait in the waves' own row buffers, the group meets, writes the tile out together and meets again (no ring).  Whole
    launches and launches of one to three workgroups (each walks through many tiles), short last tiles (H mod 16 = 4, 8, 12:
    waves without rows walk through the meetings all the same), cropped depths,
    averaging, 8-bit samples, one and both images; several repeats because the interleaving is a matter of timing.  Bit-identical
    to the row-major images transposed on the host."""
    from fdoct_amd.capi import KERNEL_FUSED_TRANSPOSED
    W, N = 1024, 1024
    nframes = 4 * A
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
    frames = synth.make_frames(23, nframes, W, H, dtype=dt)
    yb = synth.make_background(W, dtype=dt)
    r = Reconstructor(cfg)
    r.set_background(yb)
    bscan, db = r.process(frames)
    if blocks:
        r.set_launch(0, blocks)
    for rep in range(3):
        bscan_t, db_t = r.process(frames, layout=LAYOUT_TRANSPOSED)
        assert r.last_kernel() == KERNEL_FUSED_TRANSPOSED
        np.testing.assert_array_equal(bscan_t, np.transpose(bscan, (0, 2, 1)))
        np.testing.assert_array_equal(db_t, np.transpose(db, (0, 2, 1)))
    _, db_t1 = r.process(frames, want_bscan=False, layout=LAYOUT_TRANSPOSED)
    np.testing.assert_array_equal(db_t1, np.transpose(db, (0, 2, 1)))
    b_t1, _ = r.process(frames, want_db=False, layout=LAYOUT_TRANSPOSED)
    np.testing.assert_array_equal(b_t1, np.transpose(bscan, (0, 2, 1)))
    for threads in (64, 128, 192):   # one, two and three waves per workgroup
        r.set_launch(threads, blocks if blocks else 2)
        _, db_t = r.process(frames, want_bscan=False, layout=LAYOUT_TRANSPOSED)
        np.testing.assert_array_equal(db_t, np.transpose(db, (0, 2, 1)))
    r.close()
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames[:A], yb)
    helpers.check_mag(bscan[:1], mag_o, "512-point plan, transposed store")


def test_fused_transposed_store_through_the_host_pipeline_and_small_workgroups():
    """The host-pointer pipeline (fdoct_process cutting a batch into chunks over three streams) with the transposed layout:
    every chunk goes through the chain's own transposed store; and workgroups of two and three waves (fdoct_set_launch), where
    the few waves both compute and write out.  Bit-identical to the row-major images transposed on the host."""
    W, H, N, D = 2048, 200, 2048, 1024
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    base = synth.make_frames(3, 5, W, H)
    frames = np.ascontiguousarray(np.tile(base, (20, 1, 1)))           # 100 frames, 82 MB: three chunks of 40 frames
    yb = synth.make_background(W)
    r = Reconstructor(cfg)
    r.set_background(yb)
    bscan, db = r.process(base)
    bt, dt_ = r.process(frames, layout=LAYOUT_TRANSPOSED)
    assert bt.shape == (100, D, H)
    for f in (0, 4, 39, 40, 41, 79, 80, 99):
        np.testing.assert_array_equal(bt[f], bscan[f % 5].T)
        np.testing.assert_array_equal(dt_[f], db[f % 5].T)
    for threads in (128, 192):
        r.set_launch(threads, 2)
        b2, d2 = r.process(base, layout=LAYOUT_TRANSPOSED)
        np.testing.assert_array_equal(b2, np.transpose(bscan, (0, 2, 1)))
        np.testing.assert_array_equal(d2, np.transpose(db, (0, 2, 1)))
    r.close()


@pytest.mark.parametrize("bg2d,sim", [(True, False), (False, True), (True, True)])
def test_fused_transposed_store_frame_background_and_normalisation(bg2d, sim):
    """The drop-in call of INTEGRATION.md 1 as BscanFFTsim.cpp makes it -- both images in the reference's D x H layout, the
    whole-frame normalisation sim:845 always applies, the full background frame the 'b' key stores (sim:803-813) -- stays
    on the chain's own transposed store: bit-identical to the row-major images transposed on the host, and within the
    oracle tolerance."""
    W, H, N, D = 2048, 200, 2048, 1024
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, variant=VARIANT_SIM if sim else 0)
    frames = synth.make_frames(21, 3, W, H)
    yb = synth.make_background(W).astype(np.float64)
    if bg2d:
        yb = np.ascontiguousarray(np.broadcast_to(yb, (H, W))) * (1.0 + 0.01 * np.sin(np.arange(H))[:, None])
    if sim:
        yb = yb / 65535.0   # the normalised frame lives in [0, 1]
    r = Reconstructor(cfg)
    r.set_background(yb)
    bscan, db = r.process(frames)
    bscan_t, db_t = r.process(frames, layout=LAYOUT_TRANSPOSED)
    r.set_launch(0, 2)
    bscan_t2, db_t2 = r.process(frames, layout=LAYOUT_TRANSPOSED)
    r.close()
    np.testing.assert_array_equal(bscan_t, np.transpose(bscan, (0, 2, 1)))
    np.testing.assert_array_equal(db_t, np.transpose(db, (0, 2, 1)))
    np.testing.assert_array_equal(bscan_t2, bscan_t)
    np.testing.assert_array_equal(db_t2, db_t)
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames[:1], yb)
    helpers.check_mag(bscan[:1], mag_o, "transposed store, bg2d=%s sim=%s" % (bg2d, sim))


def test_u8_f32_f64_inputs_agree():
    W, H, N, D = 1024, 16, 1024, 512
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    f8 = synth.make_frames(1, 2, W, H, dtype=np.uint8)
    yb = synth.make_background(W, dtype=np.uint8)
    b8, _ = _run(cfg, f8, yb)
    b32, _ = _run(cfg, f8.astype(np.float32), yb)
    b64, _ = _run(cfg, f8.astype(np.float64), yb)
    np.testing.assert_array_equal(b32, b64)                 # f64 frames are narrowed once: the same kernel, the same bits
    helpers.check_same(b8, b32, "8-bit frames (fast path) vs the same values as floats (any-option kernel)")
    mag_o, _, _ = helpers.oracle_reference(cfg, f8, yb)
    helpers.check_mag(b8, mag_o, "u8")


def test_size_independent_properties_full_size():
    """At BASELINE's full frame size (2048 x 1000): (1) scaling the frame and the background by the
    same factor leaves the output unchanged (the (y-yp)/yb step); (2) analytic KAT: each row's peak
    sits at the depth bin n*ls/deltax (wangOCTrec4.m:200-202)."""
    W, H, N, D = 2048, 1000, 2048, 1024
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames = synth.make_frames(0, 1, W, H, noise=0.0)
    yb = synth.make_background(W).astype(np.float64)
    b1, _ = _run(cfg, frames, yb)
    b2, _ = _run(cfg, (frames // 2 * 2).astype(np.float32) * 0.5, yb * 0.5)
    b1e, _ = _run(cfg, (frames // 2 * 2), yb)
    helpers.check_mag(b2, b1e, "scale invariance")
    ls1, ls2 = synth.frame_depths_um(0, H)
    want = synth.expected_peak_bin(ls1, W)
    got = b1[0][:, 3:].argmax(axis=1) + 3
    assert np.abs(got - want).max() <= 2.5, np.abs(got - want).max()


WEAK_FAMILIES = {
    # name: (W, H, N, D, M, setup, kernel family expected, a fast-path kernel (whose second word is the default since round 5))
    "fused fast path": (2048, 64, 2048, 1024, 1, None, "KERNEL_FUSED", True),
    # (the reference's own background: a full H x W frame, 'b' key, main:1000-1075 -- on the fast path the second word of its
    # reciprocal rides along with the prefetched row as half floats)
    "fused fast path, full-frame background": (2048, 64, 2048, 1024, 1, "bg2d", "KERNEL_FUSED", True),
    "fused, transposed store, full-frame background": (2048, 64, 2048, 1024, 1, "transposed bg2d", "KERNEL_FUSED_TRANSPOSED", True),
    "fused fast path, 8-bit frames": (2048, 64, 2048, 1024, 1, "u8", "KERNEL_FUSED", True),
    "fused any-option kernel": (2048, 32, 2048, 1024, 1, lambda r: r.set_plan(-1, True), "KERNEL_FUSED", False),
    "fused, transposed store": (2048, 64, 2048, 1024, 1, "transposed", "KERNEL_FUSED_TRANSPOSED", True),
    "fused 4096-sample rows, 16 averages": (4096, 8, 4096, 2048, 1, "avg16", "KERNEL_FUSED", True),
    "fused complex rows": (2048, 32, 2048, 1024, 1, "phase", "KERNEL_FUSED", True),
    "fused 1024-sample rows": (1024, 32, 1024, 512, 1, None, "KERNEL_FUSED", True),
    # (averaging kernels with 32 samples per lane keep their constant planes in LDS, the low words among them: the long sweep of
    # round 4 found them reading a plane the host had not staged)
    "fused complex rows, 3 averages": (2048, 20, 2048, 512, 1, "phase avg3", "KERNEL_FUSED", True),
    "fused 2048-sample rows in 4096 points, 16 averages": (2048, 6, 4096, 1092, 1, "avg16", "KERNEL_FUSED", True),
    "fused, staged": (2048, 32, 2048, 1024, 1, lambda r: r.set_staged(True), "KERNEL_FUSED_STAGED", True),
    "workgroup-per-row kernel": (2048, 16, 2048, 1024, 1, lambda r: r.set_plan(-2, False), "KERNEL_GENERIC", False),
    "wave-per-row kernel (BscanFFT.ini shape)": (160, 64, 2560, 320, 4, None, "KERNEL_WAVE", False),
    "wave-per-row kernel (640 x 4)": (640, 32, 2560, 320, 4, None, "KERNEL_WAVE", False),
    "wave-per-row kernel, run-time compiled": (320, 32, 1280, 300, 4, None, "KERNEL_WAVE_JIT", False),
    "workgroup-per-row kernel, one buffer in place": (2048, 3, 32768, 2048, 8, None, "KERNEL_GENERIC", False),
    "long rows": (2048, 3, 65536, 2048, 8, None, "KERNEL_LONG_ROWS", False),
}


@pytest.mark.parametrize("family", sorted(WEAK_FAMILIES))
def test_weak_fringes_on_a_strong_background(family):
    """What an OCT sample arm returns: fringes of a few per cent, a thousandth or a ten-thousandth of the DC level.  The
    tolerance is relative to the row's peak, i.e. to the FRINGES, so every rounding of the chain has to be at the size of the
    fringe signal, not of the DC level it rides on.  main:1132 divides by the background in double; every kernel family here
    multiplies by the reciprocal as TWO floats (fdoct_capi.cpp::reciprocal_words): d = fma(v, ib, -c0) with a uniform mean
    estimate c0, d = fma(v, il, d), x - mean = d - mean(d) -- nothing rounds at the size of the DC level.
    The fused kernel's fast path does so by default since round 5 (fdoct_set_precise_division(h, 0) is the opt-out), in the
    half-float form of fdoct_kernels.h (FDOCT_PREC16: the correction c0 * (il / ib)) where a lane holds at most 32 samples;
    every other kernel always.
    (Rounds 2 and 3: lane sums of DC-sized products left 1e-8 of the DC level in the mean, 1.2 x the tolerance at 2 % fringes;
    the single f32 reciprocal a fixed pattern of <= 6e-8 of it per sample, 1.9 x the tolerance at 0.5 %, 6 x at 0.1 %.)
    The north-star tolerance (check_mag / check_db) at 2 %, 0.1 % and 0.01 % of the DC level."""
    import fdoct_amd.capi
    W, H, N, D, M, setup, want_kernel, need_flag = WEAK_FAMILIES[family]
    tokens = setup.split() if isinstance(setup, str) else []
    A = 16 if "avg16" in tokens else (3 if "avg3" in tokens else 1)
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A)
    yb = synth.make_background(W)
    dt = np.uint8 if "u8" in tokens else np.uint16
    if "u8" in tokens:
        yb = np.maximum(yb >> 8, 1).astype(np.uint8)
    if "bg2d" in tokens:   # rows that differ: every row brings its own reciprocal words
        yb = np.rint(yb[None, :] * (1.0 + 0.02 * np.sin(0.37 * np.arange(H))[:, None])).astype(np.float64)
    phase = synth.dispersion_phase(N) if "phase" in tokens else None
    worst = {}
    for amp in (2e-2, 1e-3, 1e-4):
        if "u8" in tokens and amp < 1e-3:
            continue   # (8-bit samples: fringes of 1e-4 of the DC level are a fortieth of a count)
        frames = np.concatenate([synth.weak_fringe_frame(amp, W, H, seed=5 + a, dtype=dt)[0] for a in range(A)])
        depth = synth.weak_fringe_frame(amp, W, 1)[1][0] + 6.0 * np.arange(H)
        r = Reconstructor(cfg)
        r.set_background(yb)
        if phase is not None:
            r.set_dispersion_phase(phase)
        if callable(setup):
            setup(r)
        if "transposed" in tokens:
            bt, dt_ = r.process(frames, layout=LAYOUT_TRANSPOSED)
            b, d = np.transpose(bt, (0, 2, 1)), np.transpose(dt_, (0, 2, 1))
        else:
            b, d = r.process(frames)
        assert r.last_kernel() == getattr(fdoct_amd.capi, want_kernel), (family, r.last_kernel(), r.jit_note())
        r.close()
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, phase=phase)
        if phase is None and M == 1:   # the reflector is where the generator put it (the dispersion phase smears it on purpose)
            want = synth.expected_peak_bin(depth, W)
            lo = 8
            got = b[0][:, lo:].argmax(axis=1) + lo
            ok = want < D - 4
            assert np.abs(got - want)[ok].max() <= 2.5, (family, amp)
        if amp <= 1e-3 and M == 1:
            assert mag_o[0, :, 8:].max() < 0.6 * (N / 2048.0)    # weak indeed: the DC level is 0.9 per sample
        what = "%s, fringes of %g of the DC level" % (family, amp)
        worst[amp] = (helpers.check_mag(b, mag_o, what), helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what))
    print(family, {k: (round(v[0], 3), round(v[1], 3)) for k, v in worst.items()})


NORM_FAMILIES = {
    # name: (W, H, N, D, M, setup, expected kernel family)
    "fused fast path": (2048, 32, 2048, 1024, 1, None, "KERNEL_FUSED"),
    "fused any-option kernel": (2048, 16, 2048, 1024, 1, lambda r: r.set_plan(-1, True), "KERNEL_FUSED"),
    "workgroup-per-row kernel": (2048, 8, 2048, 1024, 1, lambda r: r.set_plan(-2, False), "KERNEL_GENERIC"),
    "wave-per-row kernel": (160, 32, 2560, 320, 4, None, "KERNEL_WAVE_JIT"),
    "workgroup-per-row kernel, one buffer in place": (2048, 3, 32768, 2048, 8, None, "KERNEL_GENERIC"),
    "long rows": (2048, 3, 65536, 2048, 8, None, "KERNEL_LONG_ROWS"),
}


@pytest.mark.parametrize("mode", ["whole frame (the sim variant)", "row-wise"])
@pytest.mark.parametrize("family", sorted(NORM_FAMILIES))
def test_weak_fringes_with_a_normalisation(family, mode):
    """BscanFFTsim.cpp ALWAYS normalises the frame to [0, 1] before the division (sim:845; main:1126-1129 by ini switch): the
    normalised sample (v - min) / (max - min) is not a float, and rounding it is a rounding at the size of the DC level -- random
    from sample to sample, 3e-8 of full scale, which the chain turns into 1 x the tolerance at fringes of 0.1 % of the DC level
    and 10 x at 0.01 %.  The kernels therefore carry the normalised sample as TWO floats (p = (v - min) * scale rounded, and the
    exact residual of that product by fma) into the division by the two-word reciprocal: nothing of the normalise-and-divide
    step is rounded at the size of the DC level.  The tolerance at 2 %, 0.1 % and 0.01 % of the DC level."""
    import fdoct_amd.capi
    W, H, N, D, M, setup, want_kernel = NORM_FAMILIES[family]
    sim = mode.startswith("whole")
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M,
                 variant=VARIANT_SIM if sim else VARIANT_MAIN, rowwisenormalize=0 if sim else 1, donotnormalize=1)
    yb = synth.make_background(W).astype(np.float64) / 65535.0          # the normalised frame lives in [0, 1]
    worst = {}
    for amp in (2e-2, 1e-3, 1e-4):
        frames, _ = synth.weak_fringe_frame(amp, W, H)
        r = Reconstructor(cfg)
        r.set_background(yb)
        if setup:
            setup(r)
        b, d = r.process(frames)
        assert r.last_kernel() == getattr(fdoct_amd.capi, want_kernel), (family, r.last_kernel(), r.jit_note())
        r.close()
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb)
        what = "%s, %s normalisation, fringes of %g of the DC level" % (family, mode, amp)
        worst[amp] = (helpers.check_mag(b, mag_o, what), helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what))
    print(family, mode, {k: (round(v[0], 3), round(v[1], 3)) for k, v in worst.items()})


@pytest.mark.parametrize("options", ["plain", "dark and pi frames", "sim variant (whole-frame normalisation)", "row-wise normalisation, dark frame"])
@pytest.mark.parametrize("family", ["fused", "workgroup-per-row kernel", "wave-per-row kernel"])
def test_weak_fringes_with_the_moving_average(family, options):
    """smoothmovavg (BscanFFT.cpp:247-304, 990-991) divides its 2n + 2 taps by 2 (n + 1) in double.  n = 2: six taps, a quotient
    no f32 holds -- rounding it left 5 x the tolerance on fringes of 0.1 % of the DC level and 49 x at 0.01 % (round 4's probe,
    tools/probe_weak_movavg.py).  The pass hands on the tap sums, exact on the camera's integer samples, and the divisor is folded
    into the planes the chain subtracts and divides by (fdoct_capi.cpp::plane_scales): the dark frame always, the pi frame and the
    background unless a min-max normalisation comes first."""
    from fdoct_amd import capi
    W, H, N, D, M = {"fused": (2048, 12, 2048, 1024, 1), "workgroup-per-row kernel": (2048, 6, 2048, 1024, 1), "wave-per-row kernel": (160, 24, 2560, 320, 4)}[family]
    rng = np.random.default_rng(11)
    for n, amp in ((2, 2e-2), (2, 1e-3), (2, 1e-4), (4, 1e-3)):
        ckw = {}
        if options.startswith("sim"):
            ckw["variant"] = VARIANT_SIM
        if options.startswith("row-wise"):
            ckw["rowwisenormalize"] = 1
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, movavgn=n, **ckw)
        frames, _ = synth.weak_fringe_frame(amp, W, H)
        yb = synth.make_background(W).astype(np.float64)
        kw = {}
        normalised = "normalisation" in options
        if normalised:
            yb = yb / 65535.0
        if "dark" in options:   # integer-valued, as a camera delivers them (a non-integer one adds its own f32 rounding: DESIGN.md 4)
            kw["yd"] = np.rint(0.03 * 65535 * (1 + 0.1 * rng.standard_normal((H, W))))
        if "pi" in options:
            kw["yp"] = np.rint(0.45 * 65535 * synth.source_spectrum(W)[None, :] * (1 + 0.01 * rng.standard_normal((H, W))))
        r = Reconstructor(cfg)
        r.set_background(yb)
        if "yd" in kw:
            r.set_dark(kw["yd"])
        if "yp" in kw:
            r.set_pi_frame(kw["yp"])
        if family == "workgroup-per-row kernel":
            r.set_plan(-2, False)
        b, d = r.process(frames)
        fam = r.last_kernel()
        r.close()
        assert fam == {"fused": capi.KERNEL_FUSED, "workgroup-per-row kernel": capi.KERNEL_GENERIC}.get(family, fam), fam
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, **kw)
        what = "%s, %s, moving average of %d taps, fringes of %g of the DC level" % (family, options, 2 * n + 2, amp)
        helpers.check_mag(b, mag_o, what)
        helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)


@pytest.mark.parametrize("options", ["plain", "sim variant (whole-frame normalisation)", "row-wise normalisation", "moving average"])
@pytest.mark.parametrize("family", ["fused any-option kernel", "workgroup-per-row kernel", "long rows"])
def test_f64_frames_with_non_integer_samples(family, options):
    """data_y is CV_64F in the reference (main:987, 1125): a caller may hand over doubles that no camera delivers -- averaged or
    rescaled frames.  Narrowing such a sample to ONE float is a rounding at the size of the DC level, random from sample to
    sample (1 x the tolerance at fringes of 1e-3 of the DC level).  FDOCT_F64 frames are split once into two f32 planes, hi + lo,
    and the kernels that take float frames -- the fused any-option kernel, the workgroup-per-row kernel, the long-row path --
    carry lo into the division next to hi (FusedArgs::frames_lo), through a normalisation and the moving average too.  Frames of
    non-integer doubles with fringes of 1e-3 of the DC level against the oracle's double chain (orc_frame_to_mag)."""
    from fdoct_amd import capi
    W, H, N, D, M = {"fused any-option kernel": (2048, 12, 2048, 1024, 1), "workgroup-per-row kernel": (2048, 6, 2048, 1024, 1),
                     "long rows": (322, 6, 1288, 320, 4)}[family]
    ckw = {}
    if options.startswith("sim"):
        ckw["variant"] = VARIANT_SIM
    if options.startswith("row-wise"):
        ckw["rowwisenormalize"] = 1
    if options.startswith("moving"):
        ckw["movavgn"] = 2
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, **ckw)
    normalised = "normalisation" in options
    yb = synth.make_background(W).astype(np.float64) / (65535.0 if normalised else 7.0)
    sim = cfg.variant == VARIANT_SIM
    p = orc.make_params(W, H, N, D, M, rowwisenormalize=cfg.rowwisenormalize, donotnormalize=0 if sim else 1, movavgn=cfg.movavgn)
    idx, frac = orc.tables(W, M, N, cfg.lambdamin, cfg.lambdamax)
    lam, S = synth.lambdas(W), synth.source_spectrum(W)
    depth = 40.0 + 6.0 * np.arange(H)
    rng = np.random.default_rng(21)
    for amp in (1e-3, 1e-4):
        fringe = amp * np.cos(4 * np.pi * synth.NS * (depth[:, None] * 1e-6) / lam[None, :])
        frames = (S[None, :] * (1.0 + fringe) * 0.9 * 65535.0 / 7.0 + rng.uniform(-0.5, 0.5, (H, W)))[None]     # doubles, none of them an integer
        assert frames.dtype == np.float64 and not np.any(frames == np.rint(frames))
        r = Reconstructor(cfg)
        r.set_background(yb)
        if family == "workgroup-per-row kernel":
            r.set_plan(-2, False)
        if family == "long rows":
            r.set_plan(-3)   # (322 x 4 runs its full-length zero-pad transforms in LDS since round 6: the long-row path is asked for)
        b, d = r.process(frames)
        fam = r.last_kernel()
        want = {"fused any-option kernel": capi.KERNEL_FUSED, "workgroup-per-row kernel": capi.KERNEL_GENERIC, "long rows": capi.KERNEL_LONG_ROWS}[family]
        assert fam == want, (family, fam)
        b1, _ = r.process(frames.astype(np.float32))          # the same samples narrowed to one float by the caller
        r.close()
        mag = orc.frame_to_mag(p, frames[0], yb, None, orc.barthann(W), idx, frac)
        mag_o = mag[None, :, :D].astype(np.float64) + (1e-6 if sim else 1e-5)
        what = "f64 frames, %s, %s, fringes of %g of the DC level" % (family, options, amp)
        worst = helpers.check_mag(b, mag_o, what)
        narrowed = float(helpers.mag_ratio(b1, mag_o).max())
        print(what, "worst err/tol %.3f; narrowed to one float by the caller: %.3f" % (worst, narrowed))
        assert worst <= 0.35, (what, worst)
        # the narrowing's error grows with DC level / fringes, the two-word chain's does not.  (Not asserted under a min-max
        # normalisation: the normalised frame over this background keeps a DC-sized envelope, whose bins set the row maximum and
        # with it a tolerance either form passes easily.)
        if amp == 1e-4 and not normalised:
            assert narrowed > 2.0 * worst, (what, worst, narrowed)


@pytest.mark.parametrize("route", ["fused any-option kernel", "workgroup-per-row kernel"])
def test_moving_average_rows_whose_fringes_it_all_but_cancels(route):
    """smoothmovavg (main:247-304) is a 5-tap box with the centre counted twice: fringes near a quarter and a third of the sampling
    rate fall into its zeros, and what is left of a row whose fringes were 1e-3 of the DC level is 1e-5 of it -- an order below the
    weakest frames the fixed tests use.  Round 6's sweeps found the fused any-option kernel 0.6-0.9 x the tolerance (whole-magI-row
    maximum) from the chain in double on such rows of a 500-line frame where the f32 restatement sits at 0.06: its mean
    estimate c0 was the row's FIRST sample, in the tail of the source spectrum, and d = v / yb - c0 was rounded at the size of that
    sample's noise.  c0 is the average over the middle chunk's samples since (every row <= 0.05)."""
    from fdoct_amd import capi
    W, H, N, D = 2048, 500, 2048, 320
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, movavgn=2)
    frames = synth.weak_fringe_frame(0.001, W, H, seed=7)[0]
    yb = synth.make_background(W).astype(np.float64)[None, :] * (0.8 + 0.4 * np.random.default_rng(3).random((H, 1)))
    mag_o, _, _ = helpers.oracle_reference(cfg, frames, yb, _register=False)
    mag_t, _, _ = helpers.oracle_truth(cfg, frames, yb)
    peak, _ = helpers.truth_row_scales(cfg, frames, yb)
    for fin in (frames, frames.astype(np.float32)):
        r = Reconstructor(cfg)
        r.set_background(yb)
        if route.startswith("workgroup"):
            r.set_plan(-2, False)
        b, _ = r.process(fin)
        k = r.last_kernel()
        r.close()
        assert k == (capi.KERNEL_GENERIC if route.startswith("workgroup") else capi.KERNEL_FUSED), k
        g, o = helpers.truth_ratios_scaled(b, mag_t, mag_o, peak)
        per_row = (np.abs(b - mag_t) / (1e-4 * np.abs(mag_t) + 1e-6 * peak)).max(axis=-1)[0]
        print("%s, %s frames: worst row %.3f (f32 restatement %.3f), median row %.3f, rows beyond 0.2: %d" % (route, fin.dtype, g, o, float(np.median(per_row)), int((per_row > 0.2).sum())))
        helpers.TRUTH_LOG.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "moving average at its zeros, %s, %s frames (whole-row maximum)" % (route, fin.dtype), g, o))
        assert g <= max(0.2, o), (route, str(fin.dtype), g, o)


def test_weak_fringes_one_word_reciprocal_floor():
    """The OPT-OUT, fdoct_set_precise_division(h, 0) (or FDOCT_PRECISE_DIVISION=0): the fast path with one f32 reciprocal of the
    background, a fixed pattern of <= 6e-8 of the DC level per sample.  Inside the tolerance at fringes of 2 % of the DC level; at
    0.1 % the error is that floor -- <= 5e-6 of the DC level per depth bin, an order of magnitude less in the DC bins -- stated
    here as numbers, and it is OUTSIDE the north-star tolerance there (which is why the second word is the default since round 5:
    test_weak_fringes_on_a_strong_background runs the default through check_mag; INTEGRATION.md 4 says who may opt out)."""
    W, H, N, D = 2048, 64, 2048, 1024
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    yb = synth.make_background(W)
    for amp in (2e-2, 1e-3):
        frames, _ = synth.weak_fringe_frame(amp, W, H)
        r = Reconstructor(cfg)
        r.set_background(yb)
        r.set_precise_division(False)
        b, d = r.process(frames)
        r.close()
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb)
        if amp == 2e-2:
            helpers.check_mag(b, mag_o, "one-word reciprocal, fringes of 2 % of the DC level")
            helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, "one-word reciprocal, fringes of 2 % of the DC level")
        else:
            err = np.abs(b - mag_o)[0]
            assert err.max() <= 8e-6, err.max()                   # measured 3.9e-6
            assert err[:, :2].max() <= 2e-6, err[:, :2].max()     # measured 5.3e-7
            assert helpers.mag_ratio(b, mag_o).max() > 2.0        # ... i.e. outside the tolerance: the opt-out is one


def test_errors_are_loud():
    cfg = Config(width=2048, height=4, numfftpoints=2048, numdisplaypoints=1024)
    r = Reconstructor(cfg)
    with pytest.raises(FdoctError):  # no background yet
        r.process(synth.make_frames(0, 1, 2048, 4))
    r.close()
    with pytest.raises(FdoctError):  # numdisplaypoints beyond numfftpoints
        Reconstructor(Config(width=640, height=4, numfftpoints=640, numdisplaypoints=641))


@pytest.mark.parametrize("W,M,N,D,A,opts", [
    (4096, 4, 16384, 2048, 1, {}),                          # two buffers of 8192 values, 1024 threads: 8192 = 2 * 16^3
    (2400, 4, 9600, 1000, 2, {}),                           # ... 4800 = 5 * 5 * 3 * 16 * 4
    (4096, 8, 32768, 2048, 1, {}),                          # ONE buffer of 16384 values, every step in place: 16384 = 4 * 16^3
    (3000, 8, 24000, 1500, 2, dict(rowwisenormalize=1, dark=True)),   # ... 12000 = 5^3 * 3 * 16 * 2, 1500 = 5^3 * 3 * 4; options
    (4096, 1, 16384, 3000, 1, dict(phase=True)),            # ... complex rows: the 16384-point transform at full length
    (2000, 1, 20000, 12000, 1, {}),                         # ... a display beyond numfftpoints / 2 (the mirror)
    (5000, 2, 10000, 5000, 1, dict(phase=True)),            # ... dispersion phase on a 10000-point row (5^4 * 16), half-depth display
    (2048, 6, 24576, 777, 3, dict(sim=True)),               # ... 12288 = 3 * 16^3, whole-frame normalisation
])
def test_workgroup_per_row_kernel_on_rows_of_which_a_cu_holds_one(W, M, N, D, A, opts):
    """Rows whose DFT buffers take more than half of a CU's LDS run as one 1024-thread workgroup per CU with radix-16 passes
    (generic_kernel<1024, 1>), and rows whose two buffers do not fit at all -- transforms of 9000 ... 16384 complex points, e.g.
    4096 samples upsampled x8 (BscanFFT.cpp:1146-1147, 211, 241) -- with ONE buffer and every step in place
    (generic_kernel<1024, 1, true>) instead of leaving for the long-row path.  Against the oracle, both layouts."""
    from fdoct_amd import capi
    H = 3
    kw, ckw = {}, {}
    if opts.get("sim"):
        ckw["variant"] = VARIANT_SIM
        A = 1
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A,
                 rowwisenormalize=opts.get("rowwisenormalize", 0), **ckw)
    frames = synth.make_frames(17, 2 * A, W, H)
    yb = synth.make_background(W).astype(np.float64) + 10.0
    if opts.get("sim") or opts.get("rowwisenormalize"):
        yb = yb / 65535.0
    r = Reconstructor(cfg)
    r.set_background(yb)
    if opts.get("phase"):
        kw["phase"] = synth.dispersion_phase(N)
        r.set_dispersion_phase(kw["phase"])
    if opts.get("dark"):
        kw["yd"] = 0.02 * float(frames.max()) * np.random.default_rng(3).random((H, W))
        r.set_dark(kw["yd"])
    b, d = r.process(frames)
    assert r.last_kernel() == capi.KERNEL_GENERIC, r.last_kernel()
    bt, dt_ = r.process(frames, layout=LAYOUT_TRANSPOSED)
    r.close()
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, **kw)
    what = "workgroup-per-row kernel, 1024 threads W=%d M=%d N=%d D=%d A=%d %s" % (W, M, N, D, A, sorted(opts))
    helpers.check_mag(b, mag_o, what)
    helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)
    np.testing.assert_array_equal(bt, np.transpose(b, (0, 2, 1)))
    np.testing.assert_array_equal(dt_, np.transpose(d, (0, 2, 1)))


@pytest.mark.parametrize("W,M,N,D,A,opts,family", [
    (4096, 16, 8192, 1024, 2, {}, "long rows"),                    # 4096 samples upsampled x16 to 65536 points (main:1146-1147): beyond any LDS buffer
    (322, 4, 1288, 320, 1, {}, "in LDS"),                          # zero-pad lengths W/2 = 7 * 23, M W/2 = 2^2 * 7 * 23: Bluestein inside the zero-pad stage
    (640, 1, 16382, 320, 1, {}, "long rows"),                      # numfftpoints = 2 * 8191: Bluestein around two 32768-point transforms
    (10000, 2, 20000, 5000, 1, dict(phase=True), "long rows"),     # dispersion phase on a 20000-point row (complex: the transform runs at full length)
    (1162, 3, 3750, 256, 3, dict(rowwisenormalize=1, dark=True, sim=True), "in LDS"),   # the options, on a width whose zero-pad lengths need Bluestein (581 = 7 * 83)
    # ODD widths (a region of interest of an odd number of columns): the reference's fftshift leaves the last column of the spectrum
    # in place (main:215-227) and, under an even multiplier, pads to M W - 1 bins (main:229): the inverse transform and the row
    # it returns are M W - 1 long.  Full-length transforms of any length: on the long-row path in round 5 (60 x slower than the even
    # neighbour), inside the workgroup-per-row kernel's LDS buffers since round 6
    (321, 4, 1284, 320, 1, {}, "in LDS"),                          # 321 = 3 * 107, padded spectrum of 1283 points (a prime): Bluestein both times
    (161, 4, 2560, 320, 2, {}, "in LDS"),                          # the shipped ini's multiplier and numfftpoints on a 161-column ROI (643 points, prime)
    (225, 3, 1024, 300, 1, {}, "in LDS"),                          # odd width, odd multiplier: 675 = 3^3 5^2 points, M W itself
    (135, 2, 512, 256, 1, dict(bandpass=True, dark=True), "in LDS"),      # ... with BscanDark's band-pass, which spares the stray column
    (63, 8, 600, 200, 1, dict(sim=True), "in LDS"),                # 503 points; the sim variant's normalisation
    (1023, 4, 4096, 1024, 1, {}, "in LDS"),                        # a long odd row: 4091 points (a prime) around 8192 -- two buffers of 64 KB still fit
    (2049, 4, 4096, 1024, 1, {}, "long rows"),                     # ... and one that does not: 8195 points around 32768
])
def test_long_rows_and_any_zero_pad_length(W, M, N, D, A, opts, family):
    """Every width cv::dft / zeropadrowwise accept is accepted (BscanFFT.cpp:211, 241, 1185).  Rows too long for a compute
    unit's LDS run on the long-row path (fdoct_big.hip: rows in HBM, Stockham passes or Bluestein per length); odd widths and
    zero-pad lengths with prime factors above 5 run their FULL-length transforms inside the workgroup-per-row kernel's LDS
    buffers (round 6) as long as those fit, and on the long-row path when forced there (fdoct_set_plan(h, -3)): both routes
    against the oracle; row-major and the reference's transposed layout."""
    from fdoct_amd import capi
    H = 3
    kw, ckw = {}, {}
    if opts.get("sim"):
        ckw["variant"] = VARIANT_SIM
        A = 1
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A,
                 rowwisenormalize=opts.get("rowwisenormalize", 0), **ckw)
    frames = synth.make_frames(31, 2 * A, W, H)
    yb = synth.make_background(W).astype(np.float64) + 10.0
    if opts.get("sim") or opts.get("rowwisenormalize"):
        yb = yb / 65535.0
    r = Reconstructor(cfg)
    r.set_background(yb)
    if opts.get("phase"):
        kw["phase"] = synth.dispersion_phase(N)
        r.set_dispersion_phase(kw["phase"])
    if opts.get("dark"):
        kw["yd"] = 0.02 * float(frames.max()) * np.random.default_rng(3).random((H, W))
        r.set_dark(kw["yd"])
    if opts.get("bandpass"):
        kw["bandpass"] = 1
        r.set_bandpass(True)
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, **kw)
    routes = [(-1, capi.KERNEL_LONG_ROWS if family == "long rows" else capi.KERNEL_GENERIC)]
    if family == "in LDS":
        routes.append((-3, capi.KERNEL_LONG_ROWS))
    for plan, want_family in routes:
        r.set_plan(plan)
        b, d = r.process(frames)
        assert r.last_kernel() == want_family, (plan, r.last_kernel(), r.jit_note())
        if want_family == capi.KERNEL_LONG_ROWS:
            assert "long-row path" in r.jit_note()          # the cliff is announced (fdoct_jit_note)
        bt, dt_ = r.process(frames, layout=LAYOUT_TRANSPOSED)
        what = "%s W=%d M=%d N=%d D=%d A=%d %s" % ("long-row path" if want_family == capi.KERNEL_LONG_ROWS else "full-length zero-pad in LDS", W, M, N, D, A, sorted(opts))
        helpers.check_mag(b, mag_o, what)
        helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)
        np.testing.assert_array_equal(bt, np.transpose(b, (0, 2, 1)))
        np.testing.assert_array_equal(dt_, np.transpose(d, (0, 2, 1)))
    r.close()


def test_committed_golden_vectors():
    """The HIP path against the committed oracle outputs (tests/golden/oracle_outputs.npz, made by
    tests/golden/make_golden.py): the reference's own fixture frames and seeded C2/C3/C4 rows."""
    z = np.load(os.path.join(GOLD, "oracle_outputs.npz"))
    imgi = np.fromfile(os.path.join(GOLD, "imgi_u16_96x128.bin"), np.uint16).reshape(96, 128)
    backg = np.fromfile(os.path.join(GOLD, "backg_u16_96x128.bin"), np.uint16).reshape(96, 128)
    cfg = Config(width=128, height=96, numfftpoints=1024, numdisplaypoints=512, variant=VARIANT_SIM)
    b, d = _run(cfg, (imgi >> 8).astype(np.uint8)[None], (backg >> 8).astype(np.float64))
    helpers.check_mag(b, z["fixture_sim_u8__mag"], "golden fixture sim")
    cfg = Config(width=128, height=96, numfftpoints=1024, numdisplaypoints=512)
    b, d = _run(cfg, imgi[None], backg.astype(np.float64))
    helpers.check_mag(b, z["fixture_main_u16__mag"], "golden fixture main")
    helpers.check_db(d, np.transpose(z["fixture_main_u16__db"], (0, 2, 1)), z["fixture_main_u16__mag"], "golden fixture main")
    W, H, N, D = 2048, 8, 2048, 1024
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames, yb = synth.make_frames(100, 1, W, H), synth.make_background(W)
    b, d = _run(cfg, frames, yb)
    helpers.check_mag(b, z["c2_8rows__mag"], "golden C2")
    b, d = _run(cfg, frames, yb, window=synth.hann_window(W), phase=synth.dispersion_phase(N))
    helpers.check_mag(b, z["c3_8rows__mag"], "golden C3")
    W, H, N, D, A = 4096, 4, 4096, 2048, 4
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
    b, d = _run(cfg, synth.make_frames(200, A, W, H), synth.make_background(W))
    helpers.check_mag(b, z["c4_4rows_avg4__mag"], "golden C4")


def test_every_plan_and_both_kernels_agree_with_the_oracle():
    """N = 2048 has three compiled FFT plans (Stockham 16x16x4, 32x32 on half waves, row-swap 16|4|16)
    and two kernels (fast path / general): each one against the oracle."""
    W, H, N, D = 2048, 37, 2048, 1024
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames, yb = synth.make_frames(50, 2, W, H), synth.make_background(W)
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb)
    for plan in (2, 3, 5):
        for general in (False, True):
            r = Reconstructor(cfg)
            r.set_background(yb)
            r.set_plan(plan, general)
            b, d = r.process(frames)
            r.close()
            helpers.check_mag(b, mag_o, "plan %d general=%s" % (plan, general))
            helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, "plan %d general=%s" % (plan, general))


def test_device_pointer_batch_api_and_state_roundtrip():
    """fdoct_process_async on device-resident frames (the benchmark path) equals the host-buffer path,
    and an exported/imported state blob (the multi-GPU set-up broadcast) reproduces the results."""
    import torch
    from fdoct_amd import DTYPE_U16
    W, H, N, D = 2048, 64, 2048, 1024
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames, yb = synth.make_frames(7, 3, W, H), synth.make_background(W)
    r0 = Reconstructor(cfg)
    r0.set_background(yb)
    want_b, want_d = r0.process(frames)
    blob = r0.export_state()
    r1 = Reconstructor(cfg)
    r1.import_state(blob)
    d_in = torch.from_numpy(frames.view(np.int16)).cuda()
    d_b = torch.empty((3, H, D), dtype=torch.float32, device="cuda")
    d_d = torch.empty((3, H, D), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    r1.set_stream(st.cuda_stream)
    r1.process_device(d_in.data_ptr(), DTYPE_U16, 3, W * 2, d_b.data_ptr(), d_d.data_ptr())
    r1.synchronize()
    np.testing.assert_array_equal(d_b.cpu().numpy(), want_b)
    np.testing.assert_array_equal(d_d.cpu().numpy(), want_d)
    t = r1.timing()
    assert t["ascans"] == 3 * H and t["kernel_ms"] == 0      # async calls record no device events by default
    r1.set_timing(True)
    r1.process_device(d_in.data_ptr(), DTYPE_U16, 3, W * 2, d_b.data_ptr(), d_d.data_ptr())
    r1.synchronize()
    t = r1.timing()
    assert t["ascans"] == 3 * H and t["kernel_ms"] > 0
    r0.close()
    r1.close()


def test_staged_mode_equals_fused_chain():
    """The two-kernel mode (resample stage -> HBM -> FFT stage) must reproduce the fused chain bit for bit:
    real path at N = 2048 / 4096 and dispersion-phase (complex) path at N = 2048."""
    for N, phase_on in ((2048, False), (4096, False), (2048, True)):
        W, H, D = N, 70, N // 2
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
        frames, yb = synth.make_frames(21, 2, W, H), synth.make_background(W)
        r = Reconstructor(cfg)
        r.set_background(yb)
        if phase_on:
            r.set_dispersion_phase(synth.dispersion_phase(N))
        b0, d0 = r.process(frames)
        r.set_staged(True)
        b1, d1 = r.process(frames)
        t = r.timing()
        r.close()
        np.testing.assert_array_equal(b0, b1)
        np.testing.assert_array_equal(d0, d1)
        assert t["resample_stage_ms"] > 0 and t["fft_stage_ms"] > 0
    # with averaging (C4's shape in small: 4096 samples, 3 frames per B-scan; and the 1024-point plan): the buffer between the
    # stages holds one row per INPUT A-scan, the FFT stage averages -- bit for bit the fused chain, and get_ylin counts input rows
    for N, A in ((4096, 3), (2048, 2)):
        W, H, D = N, 9, N // 2
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
        frames, yb = synth.make_frames(5, 2 * A, W, H), synth.make_background(W)
        r = Reconstructor(cfg)
        r.set_background(yb)
        b0, d0 = r.process(frames)
        r.set_staged(True)
        b1, d1 = r.process(frames)
        y = r.get_ylin(0, 2 * A * H)
        r.close()
        if N == 4096:
            # (round 5: with 64 samples per lane the AVERAGING fast-path kernel applies the second word of 1/background as
            # c0 * rho with rho held as half floats in registers, while the resample stage -- the non-averaging instantiation,
            # it runs over input A-scans -- reads the low words as floats from its LDS plane: the same quotient to 2^-11 of a
            # term that is 6e-8 of the DC level, not the same bits.  Both are checked against the oracle on weak fringes,
            # test_weak_fringes_*; here the two must agree as closely as two kernels of the library do anywhere.)
            worst = helpers.check_same(b1, b0, "staged vs fused, 4096 samples, %d averages" % A)
            print("staged vs fused, 4096 samples with averaging: worst difference / (0.2 x tolerance) = %.3f" % worst)
            one = Reconstructor(cfg)
            one.set_background(yb)
            one.set_precise_division(False)       # one word of the reciprocal: the same arithmetic in both, bit for bit
            c0, e0 = one.process(frames)
            one.set_staged(True)
            c1, e1 = one.process(frames)
            one.close()
            np.testing.assert_array_equal(c0, c1)
            np.testing.assert_array_equal(e0, e1)
        else:
            np.testing.assert_array_equal(b0, b1)
            np.testing.assert_array_equal(d0, d1)
        assert y.shape == (2 * A * H, N) and np.isfinite(y).all() and np.abs(y[A * H:]).max() > 0
    # staged mode refuses configurations it is not built for, loudly
    cfg = Config(width=2048, height=8, numfftpoints=2048, numdisplaypoints=1024)
    r = Reconstructor(cfg)
    r.set_background(synth.make_background(2048))
    r.set_pi_frame(synth.make_background(2048) // 4)
    r.set_staged(True)
    with pytest.raises(FdoctError):
        r.process(synth.make_frames(0, 2, 2048, 8))
    r.close()
    # a phase vector at a size without a specialised complex-path kernel runs on the generic kernel
    cfg = Config(width=4096, height=8, numfftpoints=4096, numdisplaypoints=2048)
    fr, yb = synth.make_frames(0, 1, 4096, 8), synth.make_background(4096)
    ph = synth.dispersion_phase(4096)
    r = Reconstructor(cfg)
    r.set_background(yb)
    r.set_dispersion_phase(ph)
    b, _ = r.process(fr)
    r.close()
    mag_o, _, _ = helpers.oracle_reference(cfg, fr, yb, phase=ph)
    helpers.check_mag(b, mag_o, "complex N=4096 (generic kernel)")


def _parity_generic(cfg, frames, yb, what, force_generic=False, **kw):
    r = Reconstructor(cfg)
    r.set_background(yb)
    if force_generic:
        r.set_plan(-2)
    for k, fn in (("yp", r.set_pi_frame), ("yd", r.set_dark), ("window", r.set_window), ("phase", r.set_dispersion_phase)):
        if kw.get(k) is not None:
            fn(kw[k])
    bscan, db = r.process(frames)
    r.close()
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, **kw)
    helpers.check_mag(bscan, mag_o, what)
    helpers.check_db(db, np.transpose(db_o, (0, 2, 1)), mag_o, what)
    return bscan


def test_generic_kernel_shipped_ini_configuration():
    """build/BscanFFT.ini as shipped: numfftpoints 2560 (= 2^9 * 5), zero-pad multiplier 4, 320 display
    points, 10 averages; width 640 after 2x binning of 1280.  Runs on the any-configuration kernel:
    radix-5 DFT pass, zeropadrowwise (main:180-245) and averaging, against the oracle."""
    W, H, N, D, M, A = 640, 12, 2560, 320, 4, 10
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A,
                 lambdamin=840.5e-9, lambdamax=859.5e-9)
    frames = synth.make_frames(0, A, W, H)
    _parity_generic(cfg, frames, synth.make_background(W), "shipped ini")


@pytest.mark.parametrize("W,H,N,D,M,movavg", [(1280, 9, 1280, 640, 1, 0), (100, 7, 256, 100, 1, 0), (96, 5, 384, 384, 2, 0),
                                              (2048, 6, 2048, 2048, 1, 0), (512, 6, 1024, 512, 1, 3), (600, 4, 1500, 700, 1, 2),
                                              # the other shipped ini files: spin/peak/Dark, spinj (2880 = 2^6 3^2 5), webcam
                                              (640, 5, 2560, 320, 4, 0), (720, 4, 2880, 360, 4, 0), (640, 6, 640, 320, 1, 0)])
def test_generic_kernel_shapes(W, H, N, D, M, movavg):
    """Non-power-of-two N, widths that are not a multiple of 8, D up to N, zero-pad upsampling, moving average."""
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, movavgn=movavg)
    frames = synth.make_frames(1, 2, W, H)
    _parity_generic(cfg, frames, synth.make_background(W), "generic W=%d N=%d M=%d" % (W, N, M))


def test_generic_kernel_agrees_with_specialised_kernels():
    """Same configuration through both paths (and with every option on): generic vs oracle, generic vs fused."""
    W, H, N, D = 2048, 21, 2048, 1024
    rng = np.random.default_rng(9)
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, donotnormalize=0)
    frames = synth.make_frames(3, 2, W, H)
    yb = (synth.make_background(W).astype(np.float64) + 10.0) / 65535.0
    kw = dict(yp=0.01 * rng.random((H, W)), yd=20.0 * rng.random(W))
    g = _parity_generic(cfg, frames, yb, "generic all options", force_generic=True, **kw)
    f = _parity_generic(cfg, frames, yb, "fused all options", **kw)
    helpers.check_mag(g, f, "generic vs fused")
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    yb = synth.make_background(W)
    g = _parity_generic(cfg, frames, yb, "generic phase", force_generic=True, window=synth.hann_window(W),
                        phase=synth.dispersion_phase(N))
    for dt in (np.uint8, np.float32, np.float64):
        fr = (frames >> 8).astype(dt)
        a = _parity_generic(cfg, (frames >> 8).astype(np.uint8), yb, "generic u8", force_generic=True)
        r = Reconstructor(cfg)
        r.set_background(yb)
        r.set_plan(-2)
        b, _ = r.process(fr)
        r.close()
        np.testing.assert_array_equal(a, b)


def test_misaligned_device_frames_take_the_generic_kernel():
    """Device frames whose pitch is not a multiple of 16 bytes cannot use the vector-load kernels; the
    library must still produce the right answer (generic kernel), not fail or fall back to a CPU."""
    import torch
    from fdoct_amd import DTYPE_U16
    W, H, N, D = 1024, 10, 1024, 512
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames, yb = synth.make_frames(2, 1, W, H), synth.make_background(W)
    r = Reconstructor(cfg)
    r.set_background(yb)
    want, _ = r.process(frames)
    pitch = W * 2 + 6  # odd multiple of 2 bytes
    buf = np.zeros((H, pitch // 2), np.uint16)
    buf[:, :W] = frames[0]
    d_in = torch.from_numpy(buf.view(np.int16)).cuda()
    d_b = torch.empty((1, H, D), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    r.set_stream(st.cuda_stream)
    r.process_device(d_in.data_ptr(), DTYPE_U16, 1, pitch, d_b.data_ptr(), None)
    r.synchronize()
    helpers.check_mag(d_b.cpu().numpy(), want, "misaligned pitch")
    r.close()


def test_frontend_median_binning_bit_exact_and_end_to_end():
    """medianBlur + INTER_AREA binning on the GPU, bit-exact against the oracle (integer
    work), and raw camera frames through set_frontend() + process() against binning on the CPU followed by
    the oracle's reconstruction."""
    import oracle_lib as orc
    rng = np.random.default_rng(4)
    cfg = Config(width=512, height=24, numfftpoints=1024, numdisplaypoints=512)
    r = Reconstructor(cfg)
    for dt in (np.uint16, np.uint8):
        raw = rng.integers(0, np.iinfo(dt).max + 1, (2, 48, 1024)).astype(dt)
        for med, bx, by in ((0, 2, 2), (3, 2, 2), (5, 4, 2), (7, 1, 1), (0, 4, 3)):
            if 48 % by:
                continue
            if med == 7 and dt == np.uint16:   # cv::medianBlur takes ksize 7 for 8-bit frames only: no reference behaviour
                with pytest.raises(FdoctError):
                    r.frontend(raw, med, bx, by)
                continue
            got = r.frontend(raw, med, bx, by)
            want = []
            for f in raw:
                m = orc.median_blur(f, med) if med else f.astype(np.uint16)
                want.append(orc.resize_area(m, bx, by))
            np.testing.assert_array_equal(got, np.stack(want).astype(dt))
    # end to end: 2x2-binned synthetic camera frames (each sample replicated 2x2 plus +-1 count dither)
    W, H = 512, 24
    base = synth.make_frames(5, 2, W, H)
    raw = np.repeat(np.repeat(base, 2, axis=1), 2, axis=2).astype(np.int32)
    raw += rng.integers(-1, 2, raw.shape)
    raw = np.clip(raw, 0, 65535).astype(np.uint16)
    yb = synth.make_background(W)
    r.set_background(yb)
    r.set_frontend(3, 2, 2)
    b, d = r.process(raw)
    r.close()
    binned = np.stack([orc.resize_area(orc.median_blur(f, 3), 2, 2) for f in raw])
    mag_o, _, db_o = helpers.oracle_reference(cfg, binned, yb)
    helpers.check_mag(b, mag_o, "front end + chain")


@pytest.mark.parametrize("w,h", [(8, 1), (16, 3), (1000, 7), (1004, 5), (4096, 9), (24, 2)])
def test_median3_kernels_bit_exact(w, h):
    """cv::medianBlur(ksize 3) with its replicated border (main:987): the eight-pixels-per-thread kernel (widths in eights)
    and the per-pixel one (any width) against the oracle, 8- and 16-bit, several frames (no bleeding across frame borders)."""
    import oracle_lib as orc
    rng = np.random.default_rng(w * 131 + h)
    r = Reconstructor(Config(width=max(w, 8), height=h, numfftpoints=64, numdisplaypoints=32))
    for dt in (np.uint16, np.uint8):
        raw = rng.integers(0, np.iinfo(dt).max + 1, (3, h, w)).astype(dt)
        raw[0, 0, :] = np.iinfo(dt).max      # extremes on the borders
        raw[1, :, 0] = 0
        got = r.frontend(raw, 3, 1, 1)
        want = np.stack([orc.median_blur(f, 3) for f in raw]).astype(dt)
        np.testing.assert_array_equal(got, want)
    r.close()


def test_display_chain_bit_exact_and_lockin():
    """threshold / min-max normalise / x255 -> u8 / colour LUT (main:1242-1255, 1284) is byte work:
    bit-exact against the oracle on the same f32 dB input; J0 lock-in (main:1225-1230, 1260-1261) within f32 rounding.
    Also end to end: frames -> process() -> display() against the oracle's chain, which may move a pixel by one
    grey level where the f32 dB value sits on a rounding boundary."""
    import oracle_lib as orc
    rng = np.random.default_rng(11)
    cfg = Config(width=512, height=40, numfftpoints=512, numdisplaypoints=256)
    r = Reconstructor(cfg)
    lut = rng.integers(0, 256, (256, 3)).astype(np.uint8)
    for shape in ((3, 256, 40), (2, 33, 7), (1, 6, 6), (1, 1024, 1000)):
        db = (rng.standard_normal(shape) * 25.0 - 20.0).astype(np.float32)
        for thr, clamp in ((-30.0, False), (-10.0, True), (-1e9, False)):
            r.set_colormap(lut)
            gray, bgr = r.display(db, thr, clamp, colour=True)
            for b in range(shape[0]):
                want = orc.display_u8(db[b].astype(np.float64), thr, clamp)
                np.testing.assert_array_equal(gray[b], want)
                np.testing.assert_array_equal(bgr[b], orc.apply_lut(want, lut))
            np.testing.assert_array_equal(r.display(db, thr, clamp), gray)       # grey only
    # constant image: range below DBL_EPSILON => scale 0 => all zeros (cv::normalize)
    np.testing.assert_array_equal(r.display(np.full((8, 8), -50.0, np.float32), -30.0), np.zeros((8, 8), np.uint8))
    # built-in jet: blue -> cyan -> yellow -> red ramp with the published end points
    r.set_colormap(None)
    jet = r.colormap()
    assert tuple(jet[0]) == (128, 0, 0) and tuple(jet[255]) == (0, 0, 128)       # B,G,R
    assert jet[96:160, 1].min() == 255 and jet[:, 0].max() == 255 and jet[:, 2].max() == 255
    # device pointers
    import torch
    db = (rng.standard_normal((2, 256, 40)) * 25.0 - 20.0).astype(np.float32)
    d_db = torch.from_numpy(db).cuda()
    d_g = torch.empty(db.shape, dtype=torch.uint8, device="cuda")
    d_c = torch.empty(db.shape + (3,), dtype=torch.uint8, device="cuda")
    r.display_device(d_db.data_ptr(), 2, 256, 40, d_g.data_ptr(), d_c.data_ptr(), -25.0, True)
    r.synchronize()
    g2, c2 = r.display(db, -25.0, True, colour=True)
    np.testing.assert_array_equal(d_g.cpu().numpy(), g2)
    np.testing.assert_array_equal(d_c.cpu().numpy(), c2)
    # J0 lock-in
    bs = np.abs(rng.standard_normal((3, 256, 40))).astype(np.float32)
    js = np.abs(rng.standard_normal((256, 40))).astype(np.float32)
    got = r.lockin_db(bs, js)
    want = np.stack([orc.lockin_db(b.astype(np.float64), js.astype(np.float64)) for b in bs])
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=1e-5)
    # end to end
    frames = synth.make_frames(21, 1, 512, 40)
    yb = synth.make_background(512)
    r.set_background(yb)
    _, d = r.process(frames, layout=LAYOUT_TRANSPOSED)
    gray = r.display(d[0], -30.0)
    _, _, db_o = helpers.oracle_reference(cfg, frames, yb)
    want = orc.display_u8(db_o[0], -30.0, False)
    diff = np.abs(gray.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 0.01
    r.close()


def test_host_pipeline_chunks_equal_single_shot_and_pinned_buffers():
    """fdoct_process with host buffers on both sides pipelines large batches in 8-16 MB chunks over three streams:
    results must equal the device-pointer path bit for bit, for pageable and for pinned (fdoct_host_alloc) buffers,
    with averaging groups and the transposed layout crossing chunk boundaries."""
    import torch
    from fdoct_amd import PinnedArray
    W, H, N, D, A = 2048, 500, 2048, 1024, 2
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
    nf = 72                                                   # 2 MB per frame, 144 MB -> 8-frame chunks -> 9 chunks
    frames = np.tile(synth.make_frames(31, 8, W, H), (nf // 8, 1, 1))
    frames[40:] = frames[40:][::-1]                           # chunks must not be interchangeable
    r = Reconstructor(cfg)
    r.set_background(synth.make_background(W))
    for layout in (LAYOUT_TRANSPOSED, 0):
        shp = (nf // A, D, H) if layout == LAYOUT_TRANSPOSED else (nf // A, H, D)
        d_in = torch.from_numpy(frames.view(np.int16)).cuda()
        d_b = torch.empty(shp, dtype=torch.float32, device="cuda")
        d_d = torch.empty(shp, dtype=torch.float32, device="cuda")
        r.process_device(d_in.data_ptr(), 1, nf, W * 2, d_b.data_ptr(), d_d.data_ptr(), layout)
        r.synchronize()
        b, d = r.process(frames, layout=layout)              # pageable, pipelined
        np.testing.assert_array_equal(b, d_b.cpu().numpy())
        np.testing.assert_array_equal(d, d_d.cpu().numpy())
        pin_in, pin_b, pin_d = PinnedArray(frames.shape, np.uint16), PinnedArray(shp, np.float32), PinnedArray(shp, np.float32)
        pin_in.array[...] = frames
        r.process(pin_in.array, layout=layout, out_bscan=pin_b.array, out_db=pin_d.array)
        np.testing.assert_array_equal(pin_b.array, b)
        np.testing.assert_array_equal(pin_d.array, d)
        t = r.timing()
        assert t["ascans"] == nf * H and t["process_ms"] > 0
        # pageable buffers go through the handle's pinned staging slots and copy threads by default (fdoct_set_host_staging);
        # the same batch with the runtime's own bounce copies, with one and with five copy threads, and with only ONE side
        # pageable (staged per buffer), is the same batch
        for threads in (0, 1, 5, -1):
            r.set_host_staging(threads)
            b2, d2 = r.process(frames, layout=layout)
            np.testing.assert_array_equal(b2, b)
            np.testing.assert_array_equal(d2, d)
        pin_d.array[...] = 0
        b3, _ = r.process(frames, layout=layout, out_db=pin_d.array)          # pageable in, one image pageable, one pinned
        np.testing.assert_array_equal(b3, b)
        np.testing.assert_array_equal(pin_d.array, d)
        b4, d4 = r.process(pin_in.array, layout=layout)                       # pinned in, pageable out
        np.testing.assert_array_equal(b4, b)
        np.testing.assert_array_equal(d4, d)
        for pa in (pin_in, pin_b, pin_d):
            pa.free()
    r.close()


def test_host_pipeline_when_lines_times_bins_is_no_multiple_of_the_averages():
    """The pipelined host path sized a chunk's images as frames x (H D / A) -- an integer division: with 251 lines, 18 depth bins and
    16 averages every chunk was 6 floats short and landed 6 floats early (found by the sweep's tall frames, round 4).  Chunks are
    whole averaging groups: H D floats each."""
    W, H, N, D, A = 1557, 251, 225, 18, 16
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
    frames = synth.make_frames(3, 4 * A, W, H).astype(np.float32)   # 1.56 MB per frame, 100 MB -> chunks of 16 frames (one group) -> 4 chunks
    yb = synth.make_background(W).astype(np.float64) + 10.0
    r = Reconstructor(cfg)
    r.set_background(yb)
    b, d = r.process(frames)
    bt, dt_ = r.process(frames, layout=LAYOUT_TRANSPOSED)
    r.close()
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames.astype(np.uint16), yb)
    helpers.check_mag(b, mag_o, "pipelined host path, H D not a multiple of the averages")
    helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, "pipelined host path, H D not a multiple of the averages")
    np.testing.assert_array_equal(bt, np.transpose(b, (0, 2, 1)))
    np.testing.assert_array_equal(dt_, np.transpose(d, (0, 2, 1)))


def test_full_frame_background_fast_path():
    """A full H x W background frame (what the reference's 'b' key stores, main:1000-1075) stays on the fast-path kernel
    of the 1024-point plan: parity against the oracle, agreement with the general kernel to a few f32 roundings, averaging and u8 input,
    and more rows than one pass of the grid so that the per-row prefetch of the background row is exercised."""
    rng = np.random.default_rng(17)
    W, H, N, D = 2048, 37, 2048, 1024
    for A, dt in ((1, np.uint16), (3, np.uint16), (1, np.uint8)):
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
        nf = 3 * A * 30                                    # 3330+ rows: several rows per wave
        frames = np.tile(synth.make_frames(5, 3 * A, W, H), (30, 1, 1))
        yb = synth.make_background(W).astype(np.float64)[None, :] * (0.8 + 0.4 * rng.random((H, 1))) + 50.0 * rng.random((H, W))
        if dt == np.uint8:
            frames = (frames >> 8).astype(np.uint8)
            yb = yb / 256.0 + 1.0
        r = Reconstructor(cfg)
        r.set_background(yb)
        b, d = r.process(frames)
        r.set_plan(-1, True)                               # the predicated any-option kernel
        bg, dg = r.process(frames)
        r.close()
        helpers.check_same(b, bg, "fast-path option vs the any-option kernel")
        helpers.check_db(d, dg, bg, "fast-path option vs the any-option kernel (dB)")
        sel = slice(0, 2 * A)                              # oracle on the first two output frames
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames[sel], yb)
        helpers.check_mag(b[:2], mag_o, "2-D background A=%d %s" % (A, np.dtype(dt).name))
        helpers.check_db(d[:2], np.transpose(db_o, (0, 2, 1)), mag_o, "2-D background dB")


def test_row_wise_normalisation_fast_path():
    """rowwisenormalize (normalizerows, main:88-97, 1126) on the fast-path kernel: oracle parity and bit-equality with
    the general kernel, alone and with a full-frame background / averaging / 8-bit input."""
    rng = np.random.default_rng(37)
    W, H, N, D = 2048, 23, 2048, 1024
    for A, two_d, dt in ((1, False, np.uint16), (2, True, np.uint16), (1, True, np.uint8)):
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A, rowwisenormalize=1)
        frames = np.tile(synth.make_frames(13, 2 * A, W, H), (20, 1, 1))
        frames = (frames * rng.uniform(0.3, 1.0, (frames.shape[0], H, 1))).astype(np.uint16)   # rows differ in range
        frames[0, 3] = 777                                                                   # a constant row: scale 0
        if dt == np.uint8:
            frames = (frames >> 8).astype(np.uint8)
        yb = (synth.make_background(W).astype(np.float64) + 10.0) / 65535.0
        if two_d:
            yb = yb[None, :] * (0.8 + 0.4 * rng.random((H, 1)))
        r = Reconstructor(cfg)
        r.set_background(yb)
        b, d = r.process(frames)
        r.set_plan(-1, True)
        bg, dg = r.process(frames)
        r.close()
        helpers.check_same(b, bg, "fast-path option vs the any-option kernel")
        helpers.check_db(d, dg, bg, "fast-path option vs the any-option kernel (dB)")
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames[:2 * A], yb)
        helpers.check_mag(b[:2], mag_o, "row-wise normalised A=%d 2d=%s %s" % (A, two_d, np.dtype(dt).name))


def test_whole_frame_normalisation_fast_path():
    """Whole-frame min-max normalisation (main:1128-1129; always on in BscanFFTsim.cpp:845) with the streaming min/max
    pre-pass and the fast-path kernel option, alone and together with a full-frame background: oracle parity and
    agreement with the general kernel to a few f32 roundings."""
    rng = np.random.default_rng(23)
    W, H, N, D = 2048, 29, 2048, 1024
    for variant, A, two_d in ((VARIANT_SIM, 1, False), (VARIANT_MAIN, 2, True), (VARIANT_SIM, 1, True)):
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=1 if variant == VARIANT_SIM else A,
                     donotnormalize=0, variant=variant)
        A_eff = cfg.averages
        frames = np.tile(synth.make_frames(9, 4 * A_eff, W, H), (25, 1, 1))
        frames = (frames * rng.uniform(0.3, 1.0, (frames.shape[0], 1, 1))).astype(np.uint16)   # frames differ in range
        yb = (synth.make_background(W).astype(np.float64) + 10.0) / 65535.0
        if two_d:
            yb = yb[None, :] * (0.8 + 0.4 * rng.random((H, 1)))
        r = Reconstructor(cfg)
        r.set_background(yb)
        b, d = r.process(frames)
        r.set_plan(-1, True)
        bg, dg = r.process(frames)
        r.close()
        helpers.check_same(b, bg, "fast-path option vs the any-option kernel")
        helpers.check_db(d, dg, bg, "fast-path option vs the any-option kernel (dB)")
        sel = slice(0, 2 * A_eff)
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames[sel], yb)
        helpers.check_mag(b[:2], mag_o, "normalised variant=%d A=%d 2d=%s" % (variant, A_eff, two_d))


def test_2048_point_plans_agree_with_the_oracle():
    """NC = 2048 (N = 2048 with a dispersion phase, or N = 4096 real rows) has Stockham plans (32x8x8: ids 6 / 4) and
    the one-exchange row-swap plans (32|4|16: ids 7 / 8): each one, fast path and general kernel, against the oracle;
    with averaging on the 4096-point rows (the C4 shape)."""
    for (W, N, D, A, phase_on, plans) in ((2048, 2048, 1024, 1, True, (6, 7)), (2048, 2048, 2048, 1, True, (6, 7)),
                                          (4096, 4096, 2048, 2, False, (4, 8)), (2048, 4096, 1024, 1, False, (6, 7))):
        H = 19
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=A)
        frames, yb = synth.make_frames(60, 2 * A, W, H), synth.make_background(W)
        ph = synth.dispersion_phase(N) if phase_on else None
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, phase=ph)
        for plan in plans:
            for general in (False, True):
                r = Reconstructor(cfg)
                r.set_background(yb)
                if phase_on:
                    r.set_dispersion_phase(ph)
                r.set_plan(plan, general)
                b, d = r.process(frames)
                r.close()
                helpers.check_mag(b, mag_o, "W=%d N=%d D=%d plan %d general=%s" % (W, N, D, plan, general))
                helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, "dB plan %d general=%s" % (plan, general))


def test_generic_kernel_dark_variant_bandpass():
    """BscanDark.cpp: dark-frame subtraction (dark:1269) plus the band-pass inside the zero-pad upsampling (dark:218-236),
    on the BscanDark.ini shape."""
    W, H, N, D, M, A = 640, 6, 2560, 320, 4, 2
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A,
                 lambdamin=840.5e-9, lambdamax=859.5e-9)
    rng = np.random.default_rng(3)
    frames, yb = synth.make_frames(0, A, W, H), synth.make_background(W)
    yd = 30.0 * rng.random((H, W))
    r = Reconstructor(cfg)
    r.set_background(yb)
    r.set_dark(yd)
    r.set_bandpass(True)
    b, d = r.process(frames)
    r.set_bandpass(False)
    b_off, _ = r.process(frames)
    r.close()
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, yd=yd, bandpass=1)
    helpers.check_mag(b, mag_o, "BscanDark band-pass")
    helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, "BscanDark band-pass dB")
    mag_off, _, _ = helpers.oracle_reference(cfg, frames, yb, yd=yd)
    helpers.check_mag(b_off, mag_off, "BscanDark, band-pass off")
    assert np.abs(b - b_off).max() > 1e-3 * np.abs(b_off).max()      # the filter really changes the result


@pytest.mark.parametrize("W,M,N,D,A,dt", [(49, 3, 512, 14, 1, np.uint16), (64, 2, 2048, 1000, 3, np.float32), (80, 4, 100, 20, 16, np.uint16),
                                          (320, 4, 720, 300, 1, np.float32), (135, 2, 512, 256, 1, np.uint8), (720, 4, 2880, 360, 1, np.uint16)])
def test_bandpass_on_short_rows_against_the_chain_in_double(W, M, N, D, A, dt):
    """BscanDark's band-pass (dark:218-236) keeps 3 <= k < W/10 of a row's spectrum: on short rows that is a handful of bins, what is
    displayed is the window's leakage into them, and every float rounding in front of the blanking counts at the size of the whole
    row -- the shapes the round-6 sweeps reported (HIP 0.8 ... 1.2 x the tolerance from the chain in double with the forward
    transform in float).  The row is formed and the kept bins are evaluated in double since: every route (wave-per-row kernel,
    compiled or not, and the workgroup-per-row kernel; odd widths through the full-length form) is held to
    |gpu - truth| <= max(0.5 tol, the f32 restatement's own distance) with the row maximum of the DISPLAYED bins, weak fringes included."""
    H = 7
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A)
    rng = np.random.default_rng([W, N])
    top = 255 if dt == np.uint8 else 65535
    yb = synth.make_background(W)
    for weak in (None, 0.02, 0.001):
        f = synth.make_frames(5, 2 * A, W, H).astype(np.float64)
        if weak is not None:  # fringes of `weak` x the DC level
            dc = f.mean(axis=2, keepdims=True)
            f = dc + (f - dc) * (weak * float(dc.mean()) / max(1e-12, float(np.abs(f - dc).max())))
        f = f * (0.9 * top / f.max())
        ints = np.clip(np.rint(f), 0, top).astype(np.uint8 if dt == np.uint8 else np.uint16)   # (float frames: integer-valued, as a camera's are)
        frames = ints.astype(dt)
        yd = 0.02 * float(ints.max()) * rng.random((H, W))
        mag_o, _, db_o = helpers.oracle_reference(cfg, ints, yb, yd=yd, bandpass=1, _register=False)
        mag_t, _, db_t = helpers.oracle_truth(cfg, ints, yb, yd=yd, bandpass=1)
        for jit in (True, False):
            r = Reconstructor(cfg)
            r.set_background(yb)
            r.set_dark(yd)
            r.set_bandpass(True)
            r.set_jit(jit)
            b, d = r.process(frames)
            k = r.last_kernel()
            r.close()
            what = "band-pass %dx%d -> %d, %d averages, weak=%s, kernel %d" % (W, M, N, A, weak, k)
            g, o = helpers.check_truth(b, mag_t, mag_o, what)
            helpers.TRUTH_LOG.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], what, g, o))
            gd = float(helpers.db_ratio(d, np.transpose(db_t, (0, 2, 1)), mag_t).max())
            od = float(helpers.db_ratio(np.transpose(db_o, (0, 2, 1)), np.transpose(db_t, (0, 2, 1)), mag_t).max())
            assert gd <= max(helpers.TRUTH_LIMIT, od), "%s: dB image %.3g x its tolerance from the chain in double (f32 restatement %.3g)" % (what, gd, od)


def test_edge_cases_small_degenerate_and_ragged():
    """One row, one frame, one display point; all-zero and saturated frames; zeros in the background (x/0 = 0 as in
    OpenCV's Mat division); a padded row pitch; frame counts that do not fill an averaging group."""
    rng = np.random.default_rng(29)
    # smallest shapes through both paths (fused plans need W % 8 == 0 and power-of-two N; the rest is generic)
    for (W, H, N, D) in ((256, 1, 512, 1), (256, 1, 512, 256), (8, 1, 8, 4), (10, 2, 30, 30), (2048, 1, 2048, 1024)):
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
        frames = synth.make_frames(2, 1, max(W, 64), H)[:, :, :W].copy()
        _parity(cfg, frames, synth.make_background(max(W, 64))[:W].copy(), "tiny W=%d H=%d N=%d D=%d" % (W, H, N, D))
    # degenerate rows: zeros, saturation, constant rows (mean removal leaves nothing: output = epsilon)
    W, H, N, D = 512, 6, 512, 256
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames = synth.make_frames(4, 1, W, H)
    frames[0, 0] = 0
    frames[0, 1] = 65535
    frames[0, 2] = 1234
    yb = synth.make_background(W).astype(np.float64)
    yb[[5, 77, 300]] = 0.0                                   # division by zero -> 0
    r = Reconstructor(cfg)
    r.set_background(np.ones(W))
    b, d = r.process(frames)
    assert np.all(np.abs(b[0, [0, 1, 2]] - 1e-5) < 1e-9)     # constant rows: pure epsilon (main:1222)
    assert np.all(np.isfinite(b)) and np.all(np.isfinite(d))
    r.close()
    _parity(cfg, frames, yb, "zeros in the background")
    # padded pitch on the host side (cv::Mat rows with a step larger than the row)
    pitch_w = W + 24
    padded = np.zeros((2, H, pitch_w), np.uint16)
    fr2 = synth.make_frames(6, 2, W, H)
    padded[:, :, :W] = fr2
    r = Reconstructor(cfg)
    r.set_background(yb)
    want_b, want_d = r.process(fr2)
    import ctypes
    from fdoct_amd import capi
    got_b = np.empty_like(want_b)
    rc = r.lib.fdoct_process(r.h, padded.ctypes.data, capi.DTYPE_U16, capi.MEM_HOST, 2, pitch_w * 2, got_b.ctypes.data, None,
                             capi.MEM_HOST, capi.LAYOUT_ROWMAJOR)
    assert rc == 0
    np.testing.assert_array_equal(got_b, want_b)
    r.close()
    # ragged batch: nframes must be a multiple of averages, and zero frames is an error, not a no-op
    r = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=3))
    r.set_background(yb)
    with pytest.raises(FdoctError):
        r.process(synth.make_frames(0, 4, W, H))
    with pytest.raises(FdoctError):
        r.process(np.zeros((0, H, W), np.uint16))
    r.close()


def test_generic_kernel_options_matrix():
    """The any-configuration kernel with the options the specialised tests cover elsewhere: dispersion phase on a
    non-power-of-two N, row-wise and whole-frame normalisation, pi and dark frames, more samples than FFT points
    (fractionalk is read past its end there: defined 0), float input, the transposed layout."""
    rng = np.random.default_rng(31)
    W, H = 640, 5
    yb = (synth.make_background(W).astype(np.float64) + 10.0) / 65535.0
    frames = synth.make_frames(12, 2, W, H)
    cases = [
        dict(cfgkw=dict(numfftpoints=1280, numdisplaypoints=700), kw=dict(phase=synth.dispersion_phase(1280))),
        dict(cfgkw=dict(numfftpoints=1280, numdisplaypoints=320, rowwisenormalize=1), kw={}),
        dict(cfgkw=dict(numfftpoints=1280, numdisplaypoints=320, donotnormalize=0), kw=dict(yp=0.01 * rng.random((H, W)), yd=20.0 * rng.random(W))),
        dict(cfgkw=dict(numfftpoints=1920, numdisplaypoints=320, increasefftpointsmultiplier=3, variant=VARIANT_SIM), kw={}),
        dict(cfgkw=dict(numfftpoints=320, numdisplaypoints=160), kw={}),                 # W > N
    ]
    for c in cases:
        cfg = Config(width=W, height=H, **c["cfgkw"])
        b = _parity_generic(cfg, frames, yb, "generic options %s" % (c["cfgkw"],), **c["kw"])
        # same call with float frames and the reference's D x H layout
        r = Reconstructor(cfg)
        r.set_background(yb)
        for k, fn in (("yp", r.set_pi_frame), ("yd", r.set_dark), ("phase", r.set_dispersion_phase)):
            if c["kw"].get(k) is not None:
                fn(c["kw"][k])
        bt, _ = r.process(frames.astype(np.float32), layout=LAYOUT_TRANSPOSED)
        r.close()
        np.testing.assert_array_equal(np.transpose(bt, (0, 2, 1)), b)


@pytest.mark.parametrize("N,D,phase_on", [(2048, 300, False), (2048, 70, False), (2048, 129, False), (2048, 700, True), (2048, 1500, True),
                                          (4096, 1000, False), (1024, 77, False)])
def test_cropped_depths_on_the_fast_path(N, D, phase_on):
    """numdisplaypoints below the full depth (any value, not a multiple of the wave width): the fast-path kernels'
    predicated store path, with the DC mask on bins 0/1, on the real and on the dispersion-phase path.  (The tolerance
    is relative to the maximum of the stored row, so the crops keep the reflector peaks in; D <= 4 is in the edge test.)"""
    W, H = N, 11
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D)
    frames, yb = synth.make_frames(40, 2, W, H), synth.make_background(W)
    kw = dict(phase=synth.dispersion_phase(N)) if phase_on else {}
    r = Reconstructor(cfg)
    r.set_background(yb)
    if phase_on:
        r.set_dispersion_phase(kw["phase"])
    b, d = r.process(frames)
    r.close()
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, **kw)
    helpers.check_mag(b, mag_o, "cropped D=%d" % D)
    helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, "cropped D=%d dB" % D)


def test_random_configurations():
    """A seeded sweep of random geometries (power-of-two and 2^a 3^b 5^c lengths, widths with and without the fast-path
    alignment, zero-pad multipliers, averaging) and option combinations (input type, 1-row / full-frame background, pi and
    dark frames, normalisations, dispersion phase, moving average, sim variant): every case against the oracle."""
    import fuzz_cases
    lines = []
    stats = {}
    fails = fuzz_cases.run_sweep(20261004, 60, log=lines.append, stats=stats)
    assert fails == 0, "\n".join(l for l in lines if l.startswith("FAIL"))
    assert sum(l.startswith("ok") for l in lines) >= 50
    # cases outside the tolerance against the f32 restatement that only the exact chain (helpers.oracle_truth) passes must stay a rarity
    assert stats["by_truth"] <= max(1, stats["ran"] // 30), "\n".join(l for l in lines if l.startswith("truth"))


def test_random_long_rows():
    """The same sweep drawn as long rows (4000 ... 65536 points, zero-pad up to x8): the 512- / 1024-thread workgroup-per-row
    kernels with two DFT buffers or one in place, and the long-row path, whichever the library takes -- each against the oracle."""
    import fuzz_cases
    from fdoct_amd import capi
    lines = []
    stats = {}
    fails = fuzz_cases.run_sweep(20261005, 40, log=lines.append, stats=stats, big_share=1.0)
    assert fails == 0, "\n".join(l for l in lines if l.startswith("FAIL"))
    assert sum(l.startswith("ok") for l in lines) >= 35
    fam = stats["families"]
    assert fam.get(capi.KERNEL_GENERIC, 0) >= 5 and fam.get(capi.KERNEL_LONG_ROWS, 0) >= 5, fam


def test_random_forced_routes():
    """The same sweep with most cases pushed off the route the library would take by itself: the fused any-option kernel, the
    workgroup-per-row kernel, run-time compilation off, launches of one to three workgroups, one image only."""
    import fuzz_cases
    lines = []
    stats = {}
    fails = fuzz_cases.run_sweep(20261006, 60, log=lines.append, stats=stats, jit_share=0.15, route_share=0.8)
    assert fails == 0, "\n".join(l for l in lines if l.startswith("FAIL"))
    assert sum(l.startswith("ok") for l in lines) >= 45
    assert len(stats["routes"]) >= 5, stats["routes"]


def test_random_weakly_modulated_frames():
    """The same sweep on what a sample arm returns -- fringes of 2 % or 0.1 % of the DC level, both words of the reciprocal on --
    over random geometries, options (normalisations, pi / dark frames, the moving average, averaging) and routes."""
    import fuzz_cases
    lines = []
    stats = {}
    fails = fuzz_cases.run_sweep(20261007, 50, log=lines.append, stats=stats, jit_share=0.2, route_share=0.3, weak_share=1.0)
    assert fails == 0, "\n".join(l for l in lines if l.startswith("FAIL"))
    assert sum(l.startswith("ok") and "weak=" in l for l in lines) >= 40
    assert stats["by_truth"] <= 2, "\n".join(l for l in lines if l.startswith("truth"))


def test_random_tall_frames_device_pointers_and_reused_handles():
    """The sweep's other dimensions: 60 ... 400 lines per frame and three or four B-scans per call (the pipelined host path, many
    rows per wave), the device-pointer entry point with padded row pitches and addresses off the 16-byte grid, and a second run on
    the same handle after a setter has changed its route."""
    import fuzz_cases
    lines = []
    stats = {}
    fails = fuzz_cases.run_sweep(20261008, 45, log=lines.append, stats=stats, jit_share=0.2, route_share=0.3, weak_share=0.2, tall_share=0.35,
                                 dev_share=0.5, reuse_share=0.5)
    assert fails == 0, "\n".join(l for l in lines if l.startswith("FAIL"))
    ok = [l for l in lines if l.startswith("ok")]
    assert len(ok) >= 38 and sum("device-api" in l for l in ok) >= 10 and sum("then:" in l for l in ok) >= 10, len(ok)


def test_fast_path_options_on_the_2048_point_plan():
    """Full-frame background and the two normalisations on the 2048-point row-swap plan (dispersion-phase rows of
    N = 2048, real rows of N = 4096 / W = 2048): oracle parity and agreement with the general kernel to a few f32 roundings."""
    rng = np.random.default_rng(41)
    W, H = 2048, 13
    for N, D, phase_on, cfgkw in ((2048, 1024, True, {}), (2048, 2048, True, dict(rowwisenormalize=1)),
                                  (2048, 1024, True, dict(donotnormalize=0)), (4096, 2048, False, dict(donotnormalize=0))):
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, **cfgkw)
        frames = np.tile(synth.make_frames(17, 4, W, H), (25, 1, 1))
        frames = (frames * rng.uniform(0.4, 1.0, (frames.shape[0], 1, 1))).astype(np.uint16)
        yb = (synth.make_background(W).astype(np.float64) + 10.0)[None, :] * (0.8 + 0.4 * rng.random((H, 1)))
        ph = synth.dispersion_phase(N) if phase_on else None
        r = Reconstructor(cfg)
        r.set_background(yb)
        if phase_on:
            r.set_dispersion_phase(ph)
        b, d = r.process(frames)
        r.set_plan(-1, True)
        bg, dg = r.process(frames)
        r.close()
        helpers.check_same(b, bg, "fast-path option vs the any-option kernel")
        helpers.check_db(d, dg, bg, "fast-path option vs the any-option kernel (dB)")
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames[:2], yb, phase=ph)
        helpers.check_mag(b[:2], mag_o, "2048-point plan options N=%d %s" % (N, cfgkw))


def test_bench_contract_line(monkeypatch, capsys):
    """bench.py prints ONE JSON line with the contract's keys (small batch, two timed steps), including the roofline
    and cpu_baseline objects, and its own parity spot check passes."""
    import importlib
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1", "--ramp-seconds", "0", "--frames-per-step", "6",
                                      "--cpu-seconds", "0.5", "--stage-steps", "3", "--half-chip-steps", "2", "--sustained-seconds", "0.2"])
    bench = importlib.import_module("bench")
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("C2") and d["unit"] == "A-scans/s" and d["value"] > 0
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert d["roofline"]["bound"] == "hbm" and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert "failed" not in d["parity"]
    assert 0.0 <= d["parity"]["flat_1e-3_dB_pass_rate"] <= 1.0
    # the per-stage roofline (north star: resample and FFT stages) rides in the default line
    assert [s["stage"] for s in d["stages"]] == ["resample", "fft_mag_log"]
    for s in d["stages"]:
        assert s["kernel_ms_avg"] > 0 and abs(s["frac"] - s["achieved"] / s["peak"]) < 1e-3
    assert "traffic_source" in d["roofline"] and "multi_core" in d["cpu_baseline"]
    # package power / clock of the timed region (None where the hwmon files are not readable; two steps may end before a sample)
    assert "power" in d
    # the per-clock rate on half the chip (untimed extra steps), next to the power-capped headline
    import torch
    assert d["half_chip"]["workgroups"] == torch.cuda.get_device_properties(0).multi_processor_count // 2
    assert d["half_chip"]["ascans_per_s"] > 0
    if d["power"] is not None:
        assert d["power"]["package_w_max"] > 0
        if d["power"]["cap_w"] is not None:   # power1_cap may be unreadable where power1_input is not
            assert d["power"]["package_w_max"] <= 1.1 * d["power"]["cap_w"]
    # the same full launch repeated untimed with the power sampler running, and what the process group was
    assert d["sustained"]["steps"] > 0 and d["sustained"]["ascans_per_s"] > 0
    assert d["process_group"]["ranks_seen"] == 1 and len(d["process_group"]["devices"]) == 1
    # the second ceilings: FP32 vector rate always, package power when the hwmon files are readable
    assert d["roofline_fp32"]["bound"] == "fp32_valu" and 0 < d["roofline_fp32"]["frac"] < 1
    if d.get("roofline_power"):
        rp = d["roofline_power"]
        assert rp["bound"] == "package_power" and abs(rp["frac"] - rp["achieved"] / rp["peak"]) < 1e-3 and rp["uj_per_ascan"] > 0


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """The N > 1 launch contract (torch.distributed.run, one rank per process, set-up broadcast of the state blob, barrier +
    MAX-over-ranks timing) rehearsed with two ranks sharing this GPU over gloo (RCCL refuses two ranks on one device)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
           "--steps", "3", "--warmup", "1", "--ramp-seconds", "0", "--frames-per-step", "6", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["config"]["parallelism"] == "frame-shard x2"
    assert "failed" not in d["parity"]


@pytest.mark.parametrize("what", ["fused u16 averages 2", "any-option kernel f32 averages 3", "f64 frames", "zero-pad 160 x 4 raw u8 frames binned 2 x 2",
                                  "workgroup-per-row 300 -> 600", "sim variant, last frame of 3"])
def test_host_pipeline_with_small_chunks_over_kernel_families(what, monkeypatch):
    """The host-buffer pipeline (fdoct_process: chunks of whole averaging groups over three streams, pageable buffers through the
    pinned staging slots, round 6) over the kernel families and input types, with the chunk size forced down to 1 MB so that a
    small batch is 5-20 chunks with a short last one: bit for bit the single shot (chunk size forced above the batch), in both
    layouts, staged by three copy threads, unstaged, and from pinned buffers."""
    from fdoct_amd import PinnedArray
    rng = np.random.default_rng(77)
    kw, fe, A = {}, None, 1
    if what.startswith("fused"):
        W, H, N, D, A = 1024, 60, 1024, 512, 2
        frames = synth.make_frames(5, 40, W, H)
    elif what.startswith("any-option"):
        W, H, N, D, A = 1024, 33, 1024, 300, 3
        frames = synth.make_frames(6, 30, W, H).astype(np.float32) + 0.25
    elif what.startswith("f64"):
        W, H, N, D = 512, 64, 512, 256
        frames = synth.make_frames(7, 20, W, H).astype(np.float64) + 0.125
    elif what.startswith("zero-pad"):
        W, H, N, D = 160, 120, 2560, 320
        kw = dict(increasefftpointsmultiplier=4)
        base = synth.make_frames(8, 60, 256, H, dtype=np.uint8)[:, :, :W]
        frames = np.repeat(np.repeat(base, 2, axis=1), 2, axis=2)          # raw 320 x 240 camera frames
        fe = (0, 2, 2)
    elif what.startswith("workgroup"):
        W, H, N, D, A = 300, 50, 600, 200, 4
        frames = synth.make_frames(9, 160, 512, H)[:, :, :W].copy()
    else:
        W, H, N, D = 1024, 40, 1024, 512
        kw = dict(averages=3, variant=VARIANT_SIM)
        frames = synth.make_frames(10, 180, W, H)
    if "averages" not in kw:
        kw["averages"] = A
    yb = synth.make_background(max(W, 64))[:W].astype(np.float64) + 2.0
    if what.startswith("sim"):
        yb = yb / 65535.0
    r = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, **kw))
    r.set_background(yb)
    if fe:
        r.set_frontend(*fe)
    assert frames.nbytes > 2 << 20, frames.nbytes
    for layout in (0, LAYOUT_TRANSPOSED):
        monkeypatch.setenv("FDOCT_HOST_CHUNK_MB", "100000")
        b, d = r.process(frames, layout=layout)
        assert r.timing()["kernel_ms"] > 0                       # the single shot brackets its kernels with events ...
        monkeypatch.setenv("FDOCT_HOST_CHUNK_MB", "1")
        for threads in (3, 0, -1):
            r.set_host_staging(threads)
            b1, d1 = r.process(frames, layout=layout)
            np.testing.assert_array_equal(b1, b, err_msg="%s, layout %d, staging %d" % (what, layout, threads))
            np.testing.assert_array_equal(d1, d, err_msg="%s, layout %d, staging %d" % (what, layout, threads))
            assert r.timing()["kernel_ms"] == 0 and r.timing()["process_ms"] > 0     # ... the pipeline reports its wall time only
        pin_in, pin_b, pin_d = PinnedArray(frames.shape, frames.dtype), PinnedArray(b.shape, np.float32), PinnedArray(d.shape, np.float32)
        pin_in.array[...] = frames
        r.process(pin_in.array, layout=layout, out_bscan=pin_b.array, out_db=pin_d.array)
        np.testing.assert_array_equal(pin_b.array, b)
        np.testing.assert_array_equal(pin_d.array, d)
        assert r.timing()["kernel_ms"] == 0
        for pa in (pin_in, pin_b, pin_d):
            pa.free()
    assert np.isfinite(d).all() and float(b.max()) > 0
    r.close()
