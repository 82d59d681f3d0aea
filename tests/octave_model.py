"""numpy restatement of the reference's Octave prototype -- the one independent second implementation of the
reconstruction the reference tree holds ("to check the reconstruction done in C", Matlab files/wangOCTrec4.m:1-3).

Followed lines: wangOCTrec4.m:99-104 (lambdamin/max, deltalambda, kmin, kmax), :113-115 (lambdas = linspace(lambdamin,
lambdamax - deltalambda, W); k = 2*pi ./ lambdas; klinear = linspace(kmin, kmax, numfftpoints)), :146
(plinear = interp1(k, row, klinear, 'linear')), :164 (bscan = abs(ifft(plinear))), :200-202 (deltax = pi/(kmax-kmin)).

Test infrastructure, like oracle/: it is NOT the parity oracle (the C++ block differs from it on purpose-built quirks,
SURVEY.md 8a A5) but an independent physics check: same k grid, a TRUE linear interpolation and numpy's FFT.  What
differs from the C++ block:
  * interp1 is a true lerp between the bracketing samples; the C++ steps `+ fractionalk[nearestkindex[q]] * slope` away
    from y[nearestkindex[q]] with a weight indexed by the SAMPLE (A5 quirks i, ii);
  * klinear = linspace(kmin, kmax, N) (pitch (kmax-kmin)/(N-1)); the C++ uses kmin + (f+1)*(kmax-kmin)/N;
  * ifft carries 1/N (undone here so magnitudes compare); no division by the background, DC removal or window in the
    prototype's active code (`apodi = resizedim`) -- `reconstruct(..., cxx_preprocess=True)` applies the C++ block's
    three steps first (main:1132-1142) so that ONLY the resampling differs.
"""
import numpy as np


def k_grids(W, N, lambdamin, lambdamax):
    """wangOCTrec4.m:99-104, 113-115."""
    deltalambda = (lambdamax - lambdamin) / W
    kmax = 2 * np.pi / lambdamin
    kmin = 2 * np.pi / (lambdamax - deltalambda)
    lambdas = np.linspace(lambdamin, lambdamax - deltalambda, W)
    k = 2 * np.pi / lambdas                      # decreasing
    klinear = np.linspace(kmin, kmax, N)
    return k, klinear, kmin, kmax


def reconstruct(rows, lambdamin, lambdamax, N, background=None, window=None, cxx_preprocess=False):
    """rows: (H, W) spectra.  Returns abs(ifft(interp1(k, row, klinear))) * N, shape (H, N) (wangOCTrec4.m:146, 164)."""
    y = np.asarray(rows, np.float64)
    H, W = y.shape
    if cxx_preprocess:
        if background is not None:
            bg = np.asarray(background, np.float64)
            y = np.divide(y, bg, out=np.zeros_like(y), where=bg != 0)     # main:1132, x/0 = 0
        y = y - y.mean(axis=1, keepdims=True)                             # main:1138-1139
        if window is not None:
            y = y * np.asarray(window, np.float64)[None, :]               # main:1142
    k, klinear, _, _ = k_grids(W, N, lambdamin, lambdamax)
    # np.interp wants increasing abscissae: k decreases with the sample index
    plinear = np.stack([np.interp(klinear, k[::-1], r[::-1]) for r in y])
    return np.abs(np.fft.ifft(plinear, axis=1)) * N


def depth_bin(depth_m, n_refr, W, lambdamin, lambdamax):
    """wangOCTrec4.m:200-202: one depth bin is deltax = pi/(kmax-kmin) of optical path."""
    _, _, kmin, kmax = k_grids(W, 8, lambdamin, lambdamax)
    return n_refr * depth_m / (np.pi / (kmax - kmin))
