"""Deterministic synthetic FD-OCT interferograms (numpy).

Generalises the reference's own generator, ``Matlab files/wangOCTimg.m:12-56``
(the script that produced the shipped imgi.png / backg.png fixtures), to
arbitrary frame sizes -- SURVEY.md section 8(d).  Gaussian source 850 nm /
20 nm FWHM sampled at lambda0 + sigma*linspace(-2,2,W), two reflectors per
row, n = 1.38, reflectivities 0.5 / 0.25.
"""
import numpy as np

LAMBDA0 = 850e-9
DLAMBDA = 20e-9
NS = 1.38
RS1, RS2 = 0.5, 0.25
LAMBDAMIN, LAMBDAMAX = 816e-9, 884e-9  # BscanFFTsim.cpp:276-277


def lambdas(W):
    sigma = DLAMBDA / np.sqrt(2 * np.log(2))
    return LAMBDA0 + sigma * np.linspace(-2, 2, W)


def source_spectrum(W):
    sigma = DLAMBDA / np.sqrt(2 * np.log(2))
    lam = lambdas(W)
    return np.exp(-0.5 * (lam - LAMBDA0) ** 2 / sigma ** 2)


def reference_fixture_rows(W=128, H=96):
    """wangOCTimg.m:41-49 exactly: row ii (1-based) has reflectors at ii um and ii+50 um; no noise.
    Returns (imgi, backg) as float64 in [0,1]."""
    lam = lambdas(W)
    S = source_spectrum(W)
    ii = np.arange(1, H + 1)[:, None]
    ls1 = ii * 1e-6
    ls2 = (ii + 50) * 1e-6
    E1 = RS1 * np.exp(1j * 2 * 2 * np.pi * NS * ls1 / lam[None, :])
    E2 = RS2 * np.exp(1j * 2 * 2 * np.pi * NS * ls2 / lam[None, :])
    I = S[None, :] * np.abs(1 + E1 + E2) ** 2
    imgi = I / I.max(axis=1, keepdims=True)
    backg = np.broadcast_to(S / S.max(), (H, W)).copy()
    return imgi, backg


def frame_depths_um(f, H):
    """Reflector depths of row r of frame f: ls1 = 10 + 2*((r + 7f) mod 500) um, ls2 = ls1 + 150."""
    r = np.arange(H)
    ls1 = 10.0 + 2.0 * ((r + 7 * f) % 500)
    return ls1, ls1 + 150.0


def make_frame(f, W, H, noise=0.005, dtype=np.uint16):
    """Frame f as camera counts (u16 at 0.9 full scale; u8 = the same >> 8)."""
    lam = lambdas(W)
    S = source_spectrum(W)
    ls1, ls2 = frame_depths_um(f, H)
    E1 = RS1 * np.exp(1j * 4 * np.pi * NS * (ls1[:, None] * 1e-6) / lam[None, :])
    E2 = RS2 * np.exp(1j * 4 * np.pi * NS * (ls2[:, None] * 1e-6) / lam[None, :])
    I = S[None, :] * np.abs(1 + E1 + E2) ** 2
    I = I / I.max(axis=1, keepdims=True)
    rng = np.random.Generator(np.random.PCG64(20260000 + int(f)))
    I = I + noise * rng.standard_normal(I.shape)
    q = np.clip(np.rint(I * 0.9 * 65535.0), 0, 65535).astype(np.uint16)
    if dtype == np.uint8:
        return (q >> 8).astype(np.uint8)
    return q.astype(dtype)


def make_frames(f0, n, W, H, **kw):
    return np.stack([make_frame(f0 + i, W, H, **kw) for i in range(n)])


def make_background(W, dtype=np.uint16):
    """1-row background S(lambda)/max as backg.png stores it (full-scale u16)."""
    S = source_spectrum(W)
    q = np.clip(np.rint(S / S.max() * 65535.0), 0, 65535).astype(np.uint16)
    if dtype == np.uint8:
        return (q >> 8).astype(np.uint8)
    return q.astype(dtype)


def weak_fringe_frame(amp, W, H, seed=5, dtype=np.uint16):
    """What a sample arm returns: one reflector per row (40 + 6 r um deep) whose fringes are `amp` of the DC level,
    I = S(lambda) (1 + amp cos(4 pi n z / lambda)) at 0.9 full scale, with the camera's quantisation as the only noise.
    Returns (frame[1, H, W], depths_um[H])."""
    lam = lambdas(W)
    S = source_spectrum(W)
    depth = 40.0 + 6.0 * np.arange(H)
    fringe = amp * np.cos(4 * np.pi * NS * (depth[:, None] * 1e-6) / lam[None, :])
    rng = np.random.default_rng(seed)
    I = S[None, :] * (1.0 + fringe)
    full = 65535.0 if dtype == np.uint16 else 255.0
    q = np.clip(np.rint(I * 0.9 * full + rng.uniform(-0.5, 0.5, I.shape)), 0, full).astype(dtype)
    return q[None], depth


def hann_window(W):
    """C3's window: 0.5 - 0.5 cos(2 pi p / (W-1))."""
    p = np.arange(W)
    return 0.5 - 0.5 * np.cos(2 * np.pi * p / (W - 1))


def dispersion_phase(N, a2=20.0, a3=5.0):
    """C3's phase: phi(q) = a2 x^2 + a3 x^3, x = (q - N/2)/(N/2); returns (N,2) float32 (cos, sin)."""
    x = (np.arange(N) - N / 2) / (N / 2)
    phi = a2 * x ** 2 + a3 * x ** 3
    return np.stack([np.cos(phi), np.sin(phi)], axis=1).astype(np.float32)


def expected_peak_bin(depth_um, W, lmin=LAMBDAMIN, lmax=LAMBDAMAX):
    """Analytic KAT (wangOCTrec4.m:200-202): depth bin pitch deltax = pi/(kmax-kmin)."""
    dl = (lmax - lmin) / W
    kmin = 2 * np.pi / (lmax - dl)
    kmax = 2 * np.pi / lmin
    deltax = np.pi / (kmax - kmin)
    return NS * depth_um * 1e-6 / deltax
