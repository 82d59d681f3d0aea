"""Run-time specialisation of the wave-per-row kernel (fdoct_set_jit, on by default; fdoct_amd/csrc/fdoct_jit.cpp): a geometry that is not
among the library's compiled shapes -- another ROI width, zero-pad multiplier or numfftpoints than the shipped ini files
use (build/BscanFFT.ini:9-12, 25-26, 31-32, 51-52) -- gets wave_kernel<W, M, N, ..> compiled for itself by hipRTC instead of
the workgroup-per-row kernel.  Checked: the compiled kernel is the one that runs, against the oracle, against the
workgroup-per-row kernel, from the disk cache in a second process, and the fall-back when the template cannot take the shape."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from fdoct_amd import DTYPE_U16, LAYOUT_ROWMAJOR, LAYOUT_TRANSPOSED, Config, FdoctError, Reconstructor, capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAM = dict(lambdamin=840.5e-9, lambdamax=859.5e-9)

# (width, multiplier, numfftpoints, numdisplaypoints, sample type, averages): zero-pad x2, no zero-pad with a non-power-of-two
# numfftpoints, a longer transform than any shipped one, a neighbour shape (built in for 8/16-bit samples and <= 512 bins) with a
# deep display and with float samples, a short row
SHAPES = [(1280, 2, 2560, 400, np.uint16, 2), (320, 2, 1280, 200, np.uint8, 1), (960, 1, 1920, 300, np.uint16, 3),
          (640, 4, 5120, 512, np.uint16, 1), (192, 4, 2560, 1000, np.uint16, 2), (192, 4, 2560, 320, np.float32, 1),
          (96, 4, 768, 100, np.uint16, 1),
          # widths whose upsampled row does not split evenly over the 64 lanes (ROIs of 200 / 600 / 1000 / 100 columns): the last
          # lanes own fewer samples, or none
          (200, 4, 2560, 320, np.uint8, 2), (600, 4, 2560, 320, np.uint16, 1), (1000, 4, 2560, 320, np.uint16, 1),
          (100, 4, 2560, 320, np.uint8, 1), (250, 2, 1000, 250, np.uint16, 2), (1000, 1, 2000, 400, np.uint16, 1)]


def _case(W, M, N, D, dt, A, H=37, G=2):
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A, **LAM)
    src = np.uint8 if dt == np.uint8 else np.uint16
    frames = synth.make_frames(11, G * A, max(W, 64), H, dtype=src)[:, :, :W].copy()
    yb = synth.make_background(max(W, 64), dtype=src)[:W].astype(np.float64) + 3.0
    return cfg, frames, yb


@pytest.mark.parametrize("W,M,N,D,dt,A", SHAPES)
def test_run_time_compiled_wave_kernel_against_the_oracle_and_the_workgroup_kernel(W, M, N, D, dt, A, tmp_path, monkeypatch):
    monkeypatch.setenv("FDOCT_JIT_CACHE", str(tmp_path))
    cfg, frames, yb = _case(W, M, N, D, dt, A)
    given = frames.astype(np.float32) if dt == np.float32 else frames
    r = Reconstructor(cfg)
    r.set_background(yb)
    r.set_jit(False)
    bg, dg = r.process(given)
    assert r.last_kernel() == capi.KERNEL_GENERIC, "not a built-in shape, run-time compile switched off: the workgroup-per-row kernel"
    r.set_jit(True)
    r.set_launch(0, 2)           # two workgroups: every wave strides over several rows (the persistent loop and its prefetch)
    b, d = r.process(given)
    assert r.jit_note() == "", r.jit_note()
    assert r.last_kernel() == capi.KERNEL_WAVE_JIT
    r.set_launch(0, 0)
    b2, d2 = r.process(given)
    r.close()
    np.testing.assert_array_equal(b, b2)
    np.testing.assert_array_equal(d, d2)
    assert any(f.endswith(".co") for f in os.listdir(tmp_path)), "the compiled kernel was not written to the cache directory"
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb)
    what = "run-time compiled wave kernel %dx%d -> %d" % (W, M, N)
    helpers.check_mag(b, mag_o, what)
    helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)
    helpers.check_same(b, bg, what + " vs workgroup-per-row kernel", scale=0.5)   # two DFT factorisations, each within 1.0 of the oracle above
    assert np.abs(b - bg).max() > 0, "both runs took the same kernel?"


_CHILD = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from fdoct_amd import DTYPE_U16, LAYOUT_ROWMAJOR, LAYOUT_TRANSPOSED, Config, FdoctError, Reconstructor, capi, synth
W, M, N, D, H = 1280, 2, 2560, 400, 9
cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, lambdamin=840.5e-9, lambdamax=859.5e-9)
r = Reconstructor(cfg)
r.set_background(synth.make_background(W).astype(np.float64) + 3.0)
fr = synth.make_frames(3, 1, W, H)      # (run-time compilation is on by default)
t0 = time.perf_counter()
b, _ = r.process(fr)
dt = time.perf_counter() - t0
print(json.dumps({"kernel": r.last_kernel(), "note": r.jit_note(), "first_call_s": dt, "sum": float(np.float64(b).sum())}))
"""


def test_second_process_loads_the_kernel_from_the_disk_cache(tmp_path):
    env = dict(os.environ, FDOCT_JIT_CACHE=str(tmp_path))
    runs = []
    for _ in range(2):
        p = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        runs.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1]))
    files = [f for f in os.listdir(tmp_path) if f.endswith(".co")]
    assert len(files) == 1, files
    for rr in runs:
        assert rr["kernel"] == capi.KERNEL_WAVE_JIT and rr["note"] == "", rr
    assert runs[0]["sum"] == runs[1]["sum"]
    # (no timing assertion: the point is that the second process found the file, did not write another, and ran the same code)
    # a damaged cache file is recompiled, not trusted
    path = os.path.join(tmp_path, files[0])
    blob = open(path, "rb").read()
    open(path, "wb").write(blob[:len(blob) // 2])
    p = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rr = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert rr["kernel"] == capi.KERNEL_WAVE_JIT and rr["sum"] == runs[0]["sum"], rr
    assert open(path, "rb").read() == blob


_PREPARE_CHILD = r"""
import json, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
from fdoct_amd import Config, Reconstructor, synth, capi, DTYPE_U16
W, M, N, D, H = 400, 4, 1280, 320, 8      # not a built-in shape
cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M)
r = Reconstructor(cfg)
r.set_background(synth.make_background(W))
t0 = time.perf_counter()
fam = r.prepare(DTYPE_U16)
t_prepare = time.perf_counter() - t0
import torch
fr = torch.from_numpy(synth.make_frames(0, 1, W, H).view(np.int16)).cuda()
out = torch.empty((1, H, D), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
r.process_device(fr.data_ptr(), DTYPE_U16, 1, W * 2, None, out.data_ptr())
r.synchronize()
t_first = time.perf_counter() - t0
print(json.dumps({"prepared": fam, "ran": r.last_kernel(), "note": r.jit_note(), "prepare_s": t_prepare, "first_call_s": t_first}))
"""


def test_prepare_compiles_without_frames_and_the_first_frame_does_not_stall(tmp_path):
    """fdoct_prepare: an acquisition loop's first frame must not pay for the run-time compile.  On a cold cache
    key -- fresh process, empty cache directory -- fdoct_prepare resolves the kernel family and compiles the kernel for the
    handle's geometry; the first fdoct_process_async after it takes milliseconds and runs the family prepare() announced."""
    env = dict(os.environ, FDOCT_JIT_CACHE=str(tmp_path / "cold"))
    p = subprocess.run([sys.executable, "-c", _PREPARE_CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rr = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert rr["prepared"] == capi.KERNEL_WAVE_JIT and rr["ran"] == capi.KERNEL_WAVE_JIT and rr["note"] == "", rr
    assert rr["first_call_s"] < 0.05, rr                      # the compile (0.3-0.9 s) happened in prepare
    assert rr["prepare_s"] > 2 * rr["first_call_s"], rr


def test_prepare_names_the_family_of_every_kind_of_handle():
    for cfg_kw, setup, layout, want in (
            (dict(width=2048, height=16, numfftpoints=2048, numdisplaypoints=1024), None, LAYOUT_ROWMAJOR, capi.KERNEL_FUSED),
            (dict(width=2048, height=16, numfftpoints=2048, numdisplaypoints=1024), None, LAYOUT_TRANSPOSED, capi.KERNEL_FUSED_TRANSPOSED),
            (dict(width=2048, height=16, numfftpoints=2048, numdisplaypoints=1024), lambda r: r.set_staged(True), LAYOUT_ROWMAJOR, capi.KERNEL_FUSED_STAGED),
            (dict(width=160, height=16, numfftpoints=2560, numdisplaypoints=320, increasefftpointsmultiplier=4), None, LAYOUT_ROWMAJOR, capi.KERNEL_WAVE),
            (dict(width=2048, height=16, numfftpoints=2002, numdisplaypoints=1001), None, LAYOUT_ROWMAJOR, capi.KERNEL_GENERIC),
            (dict(width=2048, height=3, numfftpoints=32768, numdisplaypoints=2048, increasefftpointsmultiplier=8), None, LAYOUT_ROWMAJOR, capi.KERNEL_GENERIC),
            (dict(width=2048, height=3, numfftpoints=65536, numdisplaypoints=2048, increasefftpointsmultiplier=8), None, LAYOUT_ROWMAJOR, capi.KERNEL_LONG_ROWS)):
        cfg = Config(**cfg_kw)
        r = Reconstructor(cfg)
        with pytest.raises(FdoctError):
            r.prepare()                                   # needs the background, like process()
        r.set_background(synth.make_background(cfg.width))
        if setup:
            setup(r)
        fam = r.prepare(DTYPE_U16, layout)
        frames = synth.make_frames(0, 1, cfg.width, cfg.height)
        r.process(frames, layout=layout)
        assert fam == want == r.last_kernel(), (cfg_kw, fam, want, r.last_kernel())
        r.close()


@pytest.mark.parametrize("W,M,N,D,dt,A", [(160, 4, 2560, 320, np.uint8, 2), (160, 4, 2560, 2560, np.uint16, 1), (640, 4, 2560, 1280, np.uint16, 1),
                                           (640, 1, 640, 500, np.uint16, 2), (320, 2, 1280, 1000, np.uint16, 1), (200, 4, 2560, 320, np.uint16, 1),
                                           (720, 4, 2880, 360, np.uint16, 1)])
def test_dispersion_phase_on_the_wave_per_row_kernel(W, M, N, D, dt, A, tmp_path, monkeypatch):
    """Dispersion compensation (fdoct_set_dispersion_phase; wangOCTrec4.m:130-131, 169) on the shipped-ini geometries: complex
    rows are a compile-time option of the wave-per-row kernel (full-length final transform, phasor multiply in the gather, no
    untangle, any numdisplaypoints up to numfftpoints), so they no longer drop to the workgroup-per-row kernel.  Against the
    oracle and against the workgroup-per-row kernel."""
    monkeypatch.setenv("FDOCT_JIT_CACHE", str(tmp_path))
    cfg, frames, yb = _case(W, M, N, D, dt, A, H=9, G=2)
    phase = synth.dispersion_phase(N)
    r = Reconstructor(cfg)
    r.set_background(yb)
    r.set_dispersion_phase(phase)
    b, d = r.process(frames)
    assert r.last_kernel() == capi.KERNEL_WAVE_JIT, r.jit_note()
    r.set_plan(-2, False)
    bg, dg = r.process(frames)
    assert r.last_kernel() == capi.KERNEL_GENERIC
    r.close()
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, phase=phase)
    what = "complex rows %dx%d->%d D=%d" % (W, M, N, D)
    helpers.check_mag(b, mag_o, what)
    helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)
    helpers.check_same(b, bg, what + " vs the workgroup-per-row kernel")


@pytest.mark.parametrize("W,M,N,D,dt,A", [(160, 4, 2560, 2000, np.uint8, 2), (160, 4, 2560, 2560, np.uint16, 1), (640, 4, 2560, 1281, np.uint16, 1),
                                           (640, 1, 640, 640, np.uint16, 2), (320, 2, 1280, 700, np.float32, 1), (720, 4, 2880, 1441, np.uint16, 1)])
def test_display_beyond_half_of_numfftpoints_on_the_wave_per_row_kernel(W, M, N, D, dt, A, tmp_path, monkeypatch):
    """numdisplaypoints may be anything up to numfftpoints (the reference crops magI.colRange(0, numdisplaypoints), main:1193):
    beyond numfftpoints / 2 the spectrum of a real row mirrors, |X[b]| = |X[N - b]|, and bin N/2 comes from Z[0] alone.  A
    compile-time option of the wave-per-row kernel; against the oracle and the workgroup-per-row kernel."""
    monkeypatch.setenv("FDOCT_JIT_CACHE", str(tmp_path))
    cfg, frames, yb = _case(W, M, N, D, dt, A, H=7, G=2)
    r = Reconstructor(cfg)
    r.set_background(yb)
    b, d = r.process(frames)
    assert r.last_kernel() == capi.KERNEL_WAVE_JIT, r.jit_note()
    r.set_plan(-2, False)
    bg, dg = r.process(frames)
    assert r.last_kernel() == capi.KERNEL_GENERIC
    r.close()
    mag_o, _, db_o = helpers.oracle_reference(cfg, frames.astype(np.uint16) if dt == np.float32 else frames, yb)
    what = "deep display %dx%d->%d D=%d" % (W, M, N, D)
    helpers.check_mag(b, mag_o, what)
    helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)
    helpers.check_same(b, bg, what + " vs the workgroup-per-row kernel")
    if D > N // 2 + 1:   # the mirror itself
        np.testing.assert_array_equal(b[..., N // 2 + 1:D], b[..., N - (N // 2 + 1):N - D:-1][..., :D - N // 2 - 1])


def test_a_cache_directory_others_can_write_to_is_left_alone(tmp_path):
    """A code object read from the disk runs on the GPU with the caller's rights: the cache is used only where
    nobody else can have put it.  A directory that group or others may write to (or that is a symbolic link) is neither read
    nor written; the kernel is compiled and runs all the same, and fdoct_jit_note says why the disk cache was skipped.  A
    private directory is created 0700 and its files 0600."""
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    env = dict(os.environ, FDOCT_JIT_CACHE=str(shared))
    p = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rr = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert rr["kernel"] == capi.KERNEL_WAVE_JIT and "not used" in rr["note"], rr
    assert not os.listdir(shared)
    link = tmp_path / "link"
    private = tmp_path / "deep" / "private"
    os.symlink(tmp_path / "shared", link)
    env = dict(os.environ, FDOCT_JIT_CACHE=str(link))
    p = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rr = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert rr["kernel"] == capi.KERNEL_WAVE_JIT and "not used" in rr["note"], rr
    env = dict(os.environ, FDOCT_JIT_CACHE=str(private))
    p = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rr = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert rr["kernel"] == capi.KERNEL_WAVE_JIT and rr["note"] == "", rr
    assert (os.stat(private).st_mode & 0o777) == 0o700
    files = os.listdir(private)
    assert len(files) == 1 and (os.stat(private / files[0]).st_mode & 0o077) == 0


def test_shapes_the_template_cannot_take_fall_back_and_say_why(tmp_path, monkeypatch):
    monkeypatch.setenv("FDOCT_JIT_CACHE", str(tmp_path))
    # 208 x 4: the half-length transforms would be 104 = 8 * 13 points (a prime factor above 5): the zero-pad stage's full-length
    # form with Bluestein, inside the workgroup-per-row kernel's LDS buffers since round 6 (the long-row path until then);
    # 30 x 4: fewer than two upsampled samples per lane (the workgroup-per-row kernel)
    for W, fam in ((208, capi.KERNEL_GENERIC), (30, capi.KERNEL_GENERIC)):
        cfg, frames, yb = _case(W, 4, 2560, 320, np.uint16, 1, H=5, G=1)
        r = Reconstructor(cfg)
        r.set_background(yb)
        b, d = r.process(frames)
        assert r.last_kernel() == fam
        r.close()
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb)
        helpers.check_mag(b, mag_o, "W=%d" % W)
    # a built-in shape never compiles anything
    cfg, frames, yb = _case(160, 4, 2560, 320, np.uint16, 1, H=5, G=1)
    r = Reconstructor(cfg)
    r.set_background(yb)
    r.process(frames)
    assert r.last_kernel() == capi.KERNEL_WAVE
    r.close()
    assert not os.listdir(tmp_path)


@pytest.mark.parametrize("W,M,N,D,dt", [(160, 4, 2560, 320, np.uint8), (640, 4, 2560, 320, np.uint16), (1280, 2, 2560, 400, np.uint16)])
def test_pi_frame_dark_frame_and_band_pass_are_options_of_the_compiled_kernel(W, M, N, D, dt, tmp_path, monkeypatch):
    """BscanDark.cpp subtracts a dark frame from every frame (dark:1269) and can band-pass inside the zero-pad stage
    (dark:218-236); BscanFFT.cpp subtracts the pi-shifted / J0 frame (main:1132).  The wave-per-row kernel takes them as
    compile-time options: the library's own instantiations are the plain set-up, a handle with an option gets its kernel from
    the run-time compiler -- for the shipped shapes too.  Each option alone and all together, against the oracle and the
    workgroup-per-row kernel."""
    monkeypatch.setenv("FDOCT_JIT_CACHE", str(tmp_path))
    cfg, frames, yb = _case(W, M, N, D, dt, 2, H=11, G=2)
    H = cfg.height
    rng = np.random.default_rng(5)
    yp1, yp2 = 0.01 * float(frames.max()) * rng.random(W), 0.01 * float(frames.max()) * rng.random((H, W))
    yd1, yd2 = 0.02 * float(frames.max()) * rng.random(W), 0.02 * float(frames.max()) * rng.random((H, W))
    for name, yp, yd, bp in (("pi frame", yp1, None, 0), ("dark frame", None, yd2, 0), ("band-pass", None, None, 1),
                             ("pi + dark + band-pass", yp2, yd1, 1)):
        r = Reconstructor(cfg)
        r.set_background(yb)
        if yp is not None:
            r.set_pi_frame(yp)
        if yd is not None:
            r.set_dark(yd)
        r.set_bandpass(bool(bp))
        b, d = r.process(frames)
        assert r.jit_note() == "" and r.last_kernel() == capi.KERNEL_WAVE_JIT, (name, r.last_kernel(), r.jit_note())
        r.set_jit(False)
        bg, _ = r.process(frames)
        assert r.last_kernel() == capi.KERNEL_GENERIC
        r.close()
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, yp=yp, yd=yd, bandpass=bp)
        what = "%dx%d -> %d with %s" % (W, M, N, name)
        helpers.check_mag(b, mag_o, what)
        helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)
        # (two DFT factorisations, each within 1.0 of the oracle above; the band-passed rows are what is left of 3 % of the spectrum)
        helpers.check_same(b, bg, what + " vs workgroup-per-row kernel", scale=1.0 if bp else 0.5)


@pytest.mark.parametrize("W,M,N,D,dt", [(160, 4, 2560, 320, np.uint8), (640, 4, 2560, 320, np.uint16), (960, 1, 1920, 300, np.uint16)])
def test_normalisations_are_options_of_the_compiled_kernel(W, M, N, D, dt, tmp_path, monkeypatch):
    """BscanFFTsim.cpp always min-max normalises the frame to [0, 1] (sim:845), BscanFFT.cpp does so unless donotnormalize is
    set and normalises row by row with rowwisenormalize (main:1126-1129, 88-97).  Both on the wave-per-row kernel compiled for the
    handle: the sim variant (whole-frame), the main variant with row-wise normalisation, and the main variant with the
    whole-frame one plus a dark frame (min / max are those of the frame AFTER the dark subtraction, dark:1269 precedes it)."""
    from fdoct_amd import VARIANT_SIM
    monkeypatch.setenv("FDOCT_JIT_CACHE", str(tmp_path))
    H, G = 11, 3
    src = np.uint8 if dt == np.uint8 else np.uint16
    frames = synth.make_frames(21, G, max(W, 64), H, dtype=src)[:, :, :W].copy()
    frames[1] = frames[1] // 2                          # frames of different ranges: the whole-frame scale is per frame
    yb = synth.make_background(max(W, 64), dtype=src)[:W].astype(np.float64) / float(np.iinfo(src).max) + 0.01   # of the normalised scale
    yd = 0.02 * float(frames.max()) * np.random.default_rng(8).random((H, W))
    for name, ckw, use_yd in (("sim variant", dict(variant=VARIANT_SIM), False), ("row-wise", dict(rowwisenormalize=1), False),
                              ("whole-frame + dark", dict(donotnormalize=0), True)):
        cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, **LAM, **ckw)
        r = Reconstructor(cfg)
        r.set_background(yb)
        if use_yd:
            r.set_dark(yd)
        b, d = r.process(frames)
        assert r.jit_note() == "" and r.last_kernel() == capi.KERNEL_WAVE_JIT, (name, r.last_kernel(), r.jit_note())
        r.set_jit(False)
        bg, _ = r.process(frames)
        assert r.last_kernel() == capi.KERNEL_GENERIC
        r.close()
        mag_o, _, db_o = helpers.oracle_reference(cfg, frames, yb, yd=yd if use_yd else None)
        what = "%dx%d -> %d, %s" % (W, M, N, name)
        helpers.check_mag(b, mag_o, what)
        helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, what)
        helpers.check_same(b, bg, what + " vs workgroup-per-row kernel", scale=0.5)


@pytest.mark.parametrize("W,M,N,D,dt,H", [(160, 4, 2560, 320, np.uint8, 120), (640, 4, 2560, 320, np.uint16, 12), (640, 4, 2560, 320, np.uint8, 9),
                                           (1000, 4, 2560, 320, np.uint16, 7), (640, 1, 640, 320, np.uint8, 10)])
def test_two_by_two_binning_inside_the_compiled_kernel(W, M, N, D, dt, H, tmp_path, monkeypatch):
    """The shipped configurations bin the raw camera frame 2 x 2 in software before the block (main:958, binvalue 2 in every
    build/*.ini but two).  With nothing else in front of the chain the kernel compiled for the handle takes the RAW frames and
    bins in its own loads; the result is that of the binning pass followed by the library's built-in kernel, bit for bit
    (integer binning, same source and flags), and within the tolerance of binning on the CPU followed by the oracle."""
    import oracle_lib as orc
    monkeypatch.setenv("FDOCT_JIT_CACHE", str(tmp_path))
    A, G = 2, 2
    rng = np.random.default_rng(W + H)
    src = np.uint8 if dt == np.uint8 else np.uint16
    base = synth.make_frames(13, G * A, max(W, 64), H, dtype=src)[:, :, :W]
    raw = np.repeat(np.repeat(base, 2, axis=1), 2, axis=2).astype(np.int32) + rng.integers(-2, 3, (G * A, 2 * H, 2 * W))
    raw = np.clip(raw, 0, np.iinfo(src).max).astype(src)
    yb = synth.make_background(max(W, 64), dtype=src)[:W].astype(np.float64) + 3.0
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M, averages=A, **LAM)
    r = Reconstructor(cfg)
    r.set_background(yb)
    r.set_frontend(0, 2, 2)
    r.set_launch(0, 2)          # the persistent row loop and its prefetch of raw row pairs
    b, d = r.process(raw)
    # (8-bit rows of more than 320 samples keep the binning pass: measured slower with twenty 2-byte loads per lane)
    fused = dt == np.uint16 or W <= 320
    assert r.jit_note() == "" and r.last_kernel() == (capi.KERNEL_WAVE_JIT if fused else capi.KERNEL_WAVE), (r.last_kernel(), r.jit_note())
    r.set_launch(0, 0)
    b1, d1 = r.process(raw)
    r.set_jit(False)
    b0, d0 = r.process(raw)     # binning pass, then the built-in kernel (or the workgroup-per-row one off the built-in shapes)
    fam0 = r.last_kernel()
    binned_gpu = r.frontend(raw, 0, 2, 2)
    r.close()
    np.testing.assert_array_equal(b, b1)
    np.testing.assert_array_equal(d, d1)
    binned = np.stack([orc.resize_area(f.astype(np.uint16), 2, 2) for f in raw]).astype(src)
    np.testing.assert_array_equal(binned_gpu, binned)
    if fam0 == capi.KERNEL_WAVE:
        np.testing.assert_array_equal(b, b0)
        np.testing.assert_array_equal(d, d0)
    else:
        assert fam0 == capi.KERNEL_GENERIC
        helpers.check_same(b, b0, "binning in the loads vs binning pass + workgroup-per-row kernel", scale=0.5)
    mag_o, _, db_o = helpers.oracle_reference(cfg, binned, yb)
    helpers.check_mag(b, mag_o, "2 x 2 binning in the kernel's loads, %dx%d -> %d" % (W, M, N))
    helpers.check_db(d, np.transpose(db_o, (0, 2, 1)), mag_o, "2 x 2 binning in the kernel's loads")
    # together with the other options and the reference's D x H layout: full-frame background, dark frame, band-pass
    from fdoct_amd import LAYOUT_TRANSPOSED
    yb2 = yb[None, :] * (0.8 + 0.4 * rng.random((H, 1)))
    yd2 = 0.02 * float(raw.max()) * rng.random((H, W))
    r = Reconstructor(cfg)
    r.set_background(yb2)
    r.set_dark(yd2)
    r.set_bandpass(M > 1)
    r.set_frontend(0, 2, 2)
    bt, dtt = r.process(raw, layout=LAYOUT_TRANSPOSED)
    assert r.jit_note() == "" and r.last_kernel() == capi.KERNEL_WAVE_JIT, (r.last_kernel(), r.jit_note())
    r.close()
    mag_o2, _, db_o2 = helpers.oracle_reference(cfg, binned, yb2, yd=yd2, bandpass=int(M > 1))
    helpers.check_mag(np.transpose(bt, (0, 2, 1)), mag_o2, "binning in the loads + 2-D background + dark + band-pass, D x H")
    helpers.check_db(np.transpose(dtt, (0, 2, 1)), np.transpose(db_o2, (0, 2, 1)), mag_o2, "binning in the loads + options, D x H")
    # a median in front of the binning, or another bin factor, keeps the separate pass (and the built-in kernel)
    r = Reconstructor(cfg)
    r.set_background(yb)
    r.set_frontend(3, 2, 2)
    r.process(raw)
    assert r.last_kernel() in (capi.KERNEL_WAVE, capi.KERNEL_WAVE_JIT) and r.jit_note() == ""
    r.close()


def test_last_kernel_names_the_family_that_ran():
    from fdoct_amd import LAYOUT_TRANSPOSED
    W, H = 2048, 64
    cfg = Config(width=W, height=H, numfftpoints=2048, numdisplaypoints=1024)
    r = Reconstructor(cfg)
    assert r.last_kernel() == capi.KERNEL_NONE
    r.set_background(synth.make_background(W))
    fr = synth.make_frames(1, 1, W, H)
    r.process(fr)
    assert r.last_kernel() == capi.KERNEL_FUSED
    r.process(fr, layout=LAYOUT_TRANSPOSED)
    assert r.last_kernel() == capi.KERNEL_FUSED_TRANSPOSED
    r.set_staged(True)
    r.process(fr)
    assert r.last_kernel() == capi.KERNEL_FUSED_STAGED
    r.close()
    for (W, M, N, D), fam in (((8192, 4, 16384, 1024), capi.KERNEL_LONG_ROWS),    # 32768 upsampled samples: beyond any LDS buffer
                              ((300, 1, 1002, 400), capi.KERNEL_GENERIC), ((160, 4, 2560, 320), capi.KERNEL_WAVE)):
        r = Reconstructor(Config(width=W, height=4, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M))
        r.set_background(synth.make_background(W))
        r.process(synth.make_frames(1, 1, W, 4))
        assert r.last_kernel() == fam, (W, M, N, D, r.last_kernel())
        r.close()


def test_compile_time_pruning_changes_no_bit():
    """Round 5: the wave-per-row kernel leaves out, at compile time, the blocks of the final transform's last pass that no depth
    bin it serves reads (wave_depth_bound) and the input blocks of the zero-pad stage's inverse transform that are zero by
    construction (wave_zero_block; the fused radix 20 runs its first stage on two inputs there).  None of it may change a value:
    nine shapes -- the shipped ones sent to the run-time compiler by a pi frame, heavy down-sampling, a x 8 zero-pad -- compiled
    with the pruning and with -DFDOCT_WAVE_PRUNE=0 -DFDOCT_WAVE_ZPRUNE=0 (tools/prune_bitcheck.py: two child processes, a
    compile cache each), outputs compared byte for byte."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "prune_bitcheck.py")], capture_output=True, text=True, timeout=600)
    print(p.stdout[-2000:], p.stderr[-2000:])
    assert p.returncode == 0 and p.stdout.count("bit-identical") == 9 and "DIFFERENT" not in p.stdout
