"""The C++ caller of the C ABI: host/bscanfft_sim, the headless counterpart of the reference's simulation harness
(BscanFFTsim.cpp:775-1131), built with plain g++ and run on the reference's saved frames (tests/golden/*.bin); its output
files -- raw f32, the `.ocv` Mat dump (BscanFFTspinj.cpp:672-715) and the Matlab text of savematasdata (main:333-339) -- are
read back with fdoct_amd/io.py and checked against the committed oracle outputs (tests/golden/oracle_outputs.npz) and a
fresh oracle run.  This is also the GPU-box test of the on-disk formats."""
import os
import subprocess

import numpy as np
import pytest

import helpers
from fdoct_amd import VARIANT_MAIN, VARIANT_SIM, Config, io, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
W, H, N, D = 128, 96, 1024, 512


@pytest.fixture(scope="module")
def harness():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(ROOT, "host", "bscanfft_sim")
    assert os.path.exists(exe)
    return exe


def _run(exe, tmp_path, frames_file, bg_file, bits, extra=()):
    prefix = str(tmp_path / "out")
    cmd = [exe, "--frames", frames_file, "--background", bg_file, "--width", str(W), "--height", str(H), "--bits", str(bits),
           "--numfftpoints", str(N), "--numdisplaypoints", str(D), "--out", prefix, *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    assert "A-scans/s" in out.stdout
    bscan = np.fromfile(prefix + "_bscan.f32", np.float32).reshape(-1, D, H)
    db = np.fromfile(prefix + "_bscandb.f32", np.float32).reshape(-1, D, H)
    return prefix, bscan, db


def test_main_variant_on_the_reference_frames_u16(harness, tmp_path):
    imgi = np.fromfile(os.path.join(GOLD, "imgi_u16_96x128.bin"), np.uint16).reshape(H, W)
    backg = np.fromfile(os.path.join(GOLD, "backg_u16_96x128.bin"), np.uint16).reshape(H, W)
    prefix, bscan, db = _run(harness, tmp_path, os.path.join(GOLD, "imgi_u16_96x128.bin"), os.path.join(GOLD, "backg_u16_96x128.bin"), 16)
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, variant=VARIANT_MAIN)
    mag_o, bscan_o, db_o = helpers.oracle_reference(cfg, imgi[None], backg.astype(np.float64))
    # the harness writes the reference's transposed D x H layout (main:1220)
    helpers.check_mag(np.transpose(bscan, (0, 2, 1)), mag_o, "C++ harness, main variant")
    helpers.check_db(np.transpose(db, (0, 2, 1)), np.transpose(db_o, (0, 2, 1)), mag_o, "C++ harness, main variant dB")
    # the committed golden outputs of the same case
    gold = np.load(os.path.join(GOLD, "oracle_outputs.npz"))
    helpers.check_mag(np.transpose(bscan, (0, 2, 1)), gold["fixture_main_u16__mag"], "C++ harness vs committed golden")
    helpers.check_db(db, gold["fixture_main_u16__db"], np.transpose(gold["fixture_main_u16__mag"], (0, 2, 1)), "C++ harness vs committed golden dB")
    # on-disk formats: the .ocv dump and the Matlab text hold the same B-scan, the text to the last bit of the f32 values
    ocv = io.read_ocv(prefix + "_bscan001.ocv")
    assert ocv.dtype == np.float32 and ocv.shape == (D, H)
    np.testing.assert_array_equal(ocv, bscan[0])
    txt = open(prefix + ".m").read()
    assert txt.startswith("bscan001=[") and txt.rstrip().endswith("];")
    m = io.read_matlab_text(txt, "bscan001")
    assert m.shape == (D, H)
    np.testing.assert_array_equal(m.astype(np.float32), bscan[0])
    # display images of main:1242-1255, 1284
    pgm = open(prefix + "_bscan001.pgm", "rb").read()
    assert pgm.startswith(b"P5\n%d %d\n255\n" % (H, D)) and len(pgm) == len(b"P5\n%d %d\n255\n" % (H, D)) + D * H


def test_sim_variant_8_bit_and_ocv_input(harness, tmp_path):
    """BscanFFTsim.cpp's own settings: 8-bit imread, whole-frame normalise (sim:845), eps 1e-6; frames handed over as the
    instrument programs' .ocv Mat dumps."""
    imgi = np.fromfile(os.path.join(GOLD, "imgi_u16_96x128.bin"), np.uint16).reshape(H, W)
    backg = np.fromfile(os.path.join(GOLD, "backg_u16_96x128.bin"), np.uint16).reshape(H, W)
    img8, bg8 = (imgi >> 8).astype(np.uint8), (backg >> 8).astype(np.uint8)
    f_ocv, b_ocv = str(tmp_path / "imgi8.ocv"), str(tmp_path / "backg8.ocv")
    io.write_ocv(f_ocv, np.concatenate([img8, img8[::-1]]).reshape(2 * H, W))   # two frames back to back
    io.write_ocv(b_ocv, bg8)
    prefix, bscan, db = _run(harness, tmp_path, f_ocv, b_ocv, 8, extra=("--sim",))
    assert bscan.shape[0] == 2
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, variant=VARIANT_SIM)
    frames = np.stack([img8, img8[::-1]])
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, bg8.astype(np.float64))
    helpers.check_mag(np.transpose(bscan, (0, 2, 1)), mag_o, "C++ harness, sim variant")
    helpers.check_db(np.transpose(db, (0, 2, 1)), np.transpose(db_o, (0, 2, 1)), mag_o, "C++ harness, sim variant dB")
    gold = np.load(os.path.join(GOLD, "oracle_outputs.npz"))
    helpers.check_mag(np.transpose(bscan[:1], (0, 2, 1)), gold["fixture_sim_u8__mag"], "C++ harness (sim) vs committed golden")


def test_clone_to_device_and_the_single_process_multi_handle_mode(harness, tmp_path):
    """SURVEY 8e's other option -- one process, one handle + host thread per GPU: fdoct_clone_to_device copies the
    configuration, constant state and settings; `bscanfft_sim --gpus 3` shards 7 frames 3/2/2 over three handles (all on
    this box's one GPU) and must write exactly what the one-handle run writes."""
    from fdoct_amd import Reconstructor
    rng = np.random.default_rng(3)
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, averages=1)
    frames = (synth.make_frames(9, 7, W, H) * rng.uniform(0.5, 1.0, (7, 1, 1))).astype(np.uint16)
    yb = synth.make_background(W).astype(np.float64)[None, :] * (0.8 + 0.4 * rng.random((H, 1)))
    r0 = Reconstructor(cfg)
    r0.set_background(yb)
    r0.set_window(synth.hann_window(W))
    r0.set_averages(1)
    r1 = r0.clone_to_device(0)
    b0, d0 = r0.process(frames)
    b1, d1 = r1.process(frames)
    np.testing.assert_array_equal(b0, b1)
    np.testing.assert_array_equal(d0, d1)
    np.testing.assert_array_equal(r1.get_window(), synth.hann_window(W))
    r0.set_window(None)                       # handles are independent after the clone
    b1b, _ = r1.process(frames)
    np.testing.assert_array_equal(b1b, b1)
    with pytest.raises(Exception):
        r0.clone_to_device(97)                # no such device
    r0.close()
    r1.close()
    ffile, bfile = str(tmp_path / "frames.bin"), str(tmp_path / "bg.bin")
    frames.tofile(ffile)
    synth.make_background(W).tofile(bfile)
    _, one, one_db = _run(harness, tmp_path, ffile, bfile, 16)
    _, three, three_db = _run(harness, tmp_path, ffile, bfile, 16, extra=("--gpus", "3", "--devices", "0,0,0"))
    assert one.shape == (7, D, H)
    np.testing.assert_array_equal(one, three)
    np.testing.assert_array_equal(one_db, three_db)


def test_precise_division_from_the_c_plus_plus_caller(harness, tmp_path):
    """`bscanfft_sim` as it comes (both words of 1/background: the library default since round 5) and with `--one-word-division`
    (fdoct_set_precise_division(h, 0) after fdoct_create, fdoct_prepare before the loop): a sample arm's weak fringes -- 0.1 % of
    the DC level -- on the fast path, the sim variant's whole-frame normalisation included, inside the tolerance from a C++ host by
    default; with the opt-out the main variant's frames are outside it."""
    w, h, n, d = 2048, 16, 2048, 1024
    frames, _ = synth.weak_fringe_frame(1e-3, w, h)
    yb = synth.make_background(w)
    (tmp_path / "fr.bin").write_bytes(frames.tobytes())
    (tmp_path / "bg.bin").write_bytes(yb.tobytes())          # one spectrum for every row: the fast path with both words
    for sim in (False, True):
        cfg = Config(width=w, height=h, numfftpoints=n, numdisplaypoints=d, variant=VARIANT_SIM if sim else VARIANT_MAIN)
        mag_o, _, _ = helpers.oracle_reference(cfg, frames, yb)
        worst = {}
        for flag in (("--one-word-division",), ()):
            prefix = str(tmp_path / ("out%d%d" % (sim, len(flag))))
            cmd = [harness, "--frames", str(tmp_path / "fr.bin"), "--background", str(tmp_path / "bg.bin"), "--width", str(w), "--height", str(h),
                   "--bits", "16", "--numfftpoints", str(n), "--numdisplaypoints", str(d), "--out", prefix, *flag] + (["--sim"] if sim else [])
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
            b = np.fromfile(prefix + "_bscan.f32", np.float32).reshape(-1, d, h)
            worst[1 - len(flag)] = float(helpers.mag_ratio(np.transpose(b, (0, 2, 1)), mag_o).max())   # [1]: default, [0]: opt-out
        assert worst[1] <= 1.0, (sim, worst)
        if not sim:   # (the main variant's plain set-up is the fast path: one word with the opt-out)
            assert worst[0] > 1.0, (sim, worst)
