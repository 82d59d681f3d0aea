"""CPU checks of the kernel's lane/register scheme through its numpy model (tests/kernel_model.py):
Stockham passes over T lanes with P points each, the padded LDS layout's bank behaviour, the
real-input untangle and its partner mapping, and the one-exchange 1024-point row-swap plan."""
import numpy as np
import pytest

import kernel_model as km

PLANS = [(256, 16, (16, 16)), (512, 16, (32, 16)), (1024, 64, (16, 16, 4)), (1024, 32, (32, 32)),
         (2048, 64, (32, 8, 8))]  # the compiled kind-0 plans of fdoct_kernels.hip


@pytest.mark.parametrize("NC,T,radices", PLANS)
def test_stockham_lane_scheme(NC, T, radices):
    rng = np.random.default_rng(NC + T)
    z = rng.standard_normal(NC) + 1j * rng.standard_normal(NC)
    Z = km.stockham_lanes(z, T, radices)
    ref = np.fft.ifft(z) * NC
    assert np.abs(Z - ref).max() <= 1e-12 * np.abs(ref).max()


def test_rowswap_plan():
    rng = np.random.default_rng(5)
    x = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
    assert np.abs(km.fft1024_rowswap_model(x) - np.fft.ifft(x) * 1024).max() <= 1e-11 * 1024


def test_untangle_and_partner_mapping():
    rng = np.random.default_rng(2)
    N = 2048
    x = rng.standard_normal(N)
    Z = np.fft.ifft(x[0::2] + 1j * x[1::2]) * (N // 2)
    X = km.untangle_real(Z, N)
    assert np.abs(X - (np.fft.ifft(x) * N)[:N // 2]).max() <= 1e-10
    for T, P in ((64, 16), (32, 32), (16, 16)):
        NC = T * P
        for l in range(T):
            for m in range(P):
                pl, pm = km.untangle_partner(l, m, T, P)
                assert pl + T * pm == (NC - (l + T * m)) % NC


@pytest.mark.parametrize("NC,T,radices", PLANS)
def test_padded_layout_bank_conflicts(NC, T, radices):
    """Exchange layout e -> e + (e >> log2(R1)): writes conflict free for every plan, read-backs
    conflict free when R1 == 32 and at most 2-way otherwise."""
    P = NC // T
    LP = int(np.log2(radices[0]))
    pad = lambda e: e + (e >> LP)
    stride = NC + (NC >> LP) + 2
    NS = 1
    for R in radices[:-1]:
        for t in range(P // R):
            for r in range(R):
                acc = []
                for lane in range(64):
                    l, sub = lane % T, lane // T
                    j = l + T * t
                    e = (j // NS) * NS * R + (j % NS) + r * NS
                    acc.append((lane, pad(e) + sub * stride))
                assert km.bank_conflicts(acc, 8, 16, 32) == 1
        worst = 1
        for m in range(P):
            acc = [(lane, pad(lane % T + T * m) + (lane // T) * stride) for lane in range(64)]
            worst = max(worst, km.bank_conflicts(acc, 8, 32, 64))
        assert worst <= (1 if radices[0] == 32 and T >= 32 else 2)
        NS *= R


def test_rowswap_exchange_banks():
    """Slot 65*b + l': writes by 16-lane groups and reads by 32-lane groups are conflict free."""
    for c in range(4):
        for k2 in range(4):
            acc = [(lane, 65 * (lane & 15) + (lane >> 4) + 4 * c + 16 * k2) for lane in range(64)]
            assert km.bank_conflicts(acc, 8, 16, 32) == 1
    for bb in range(16):
        acc = [(lane, 65 * bb + lane) for lane in range(64)]
        assert km.bank_conflicts(acc, 8, 32, 64) == 1


def test_rowswap_plan_2048():
    rng = np.random.default_rng(4)
    x = rng.standard_normal(2048) + 1j * rng.standard_normal(2048)
    assert np.abs(km.fft2048_rowswap_model(x) - np.fft.ifft(x) * 2048).max() <= 1e-11 * 2048


@pytest.mark.parametrize("n", [80, 320, 1280, 360, 1440, 160, 640])
def test_wave_per_row_plan_tables_and_inplace_passes(n):
    """fdoct_wave.hip's transforms (the lengths of the shipped shapes: W/2, M*W/2, N/2): radix plan incl. the fused 20,
    per-pass twiddle tables, in-place passes with clamped partial rounds, both directions; and the last pass's store filter
    keeps exactly what the real-input untangle reads."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    plan = km.wave_plan(n)
    assert plan is not None and int(np.prod([r for r, _ in plan])) == n
    assert all(ns == int(np.prod([r for r, _ in plan[:i]])) for i, (_, ns) in enumerate(plan))
    assert (plan[0][0] == 20) == (n % 1280 == 0)
    inv = km.wave_fft_inplace(x, inverse=True)
    assert np.abs(inv - np.fft.ifft(x) * n).max() <= 1e-11 * n
    fwd = km.wave_fft_inplace(x, inverse=False)
    assert np.abs(fwd - np.fft.fft(x)).max() <= 1e-11 * n
    D = n // 4
    kept = km.wave_fft_inplace(x, inverse=True, keep=lambda e: e < D or e > n - D)
    need = [b for b in range(D)] + [n - b for b in range(1, D)]
    assert np.abs(kept[need] - inv[need]).max() == 0
    assert km.wave_plan(2 * 7 * 64) is None


@pytest.mark.parametrize("n, D", [(1280, 320), (1280, 300), (1280, 161), (1440, 360), (1440, 100), (320, 80), (320, 41), (640, 320), (160, 33)])
def test_wave_per_row_pruned_last_pass_keeps_every_bin_a_depth_reads(n, D):
    """Round 5 (fdoct_wave.h::wave_depth_bound, fdoct_wave_dev.h::wave_pass NEVER): with the depth bound DK a kernel is compiled
    for, the blocks of the last pass the rule declares dead are exactly blocks no depth <= DK reads -- Z[k] and Z[n - k], k < D
    (real rows) -- so leaving their stores out changes nothing the untangle sees, for D at the bound and below it."""
    DK = km.wave_depth_bound(n, D)
    assert D <= DK <= n and DK % km.wave_plan(n)[-1][1] == 0
    dead = km.wave_dead_output_blocks(n, DK, n - DK)
    R, Ns = km.wave_plan(n)[-1]
    for d in (DK, D, max(1, D // 2)):
        need = set(range(d)) | {n - b for b in range(1, d)}
        assert not any(e // Ns in dead for e in need), (n, d, dead)
    if (n, D) in ((1280, 320), (1280, 300)):
        assert dead == [2, 3, 4, 5]          # BscanFFT.ini: half of the last radix-8 pass
    rng = np.random.default_rng(n + D)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    full = km.wave_fft_inplace(x, inverse=True)
    pruned = km.wave_fft_inplace(x, inverse=True, keep=lambda e: e < D or e > n - D, dead_out=dead)
    need = list(range(D)) + [n - b for b in range(1, D)]
    assert np.abs(pruned[need] - full[need]).max() == 0
    # complex rows / deep display: only e < D is kept, the bound's upper end is the transform length
    dead_c = km.wave_dead_output_blocks(n, DK, n)
    assert all(r * Ns >= DK for r in dead_c) and set(dead) <= set(dead_c)


@pytest.mark.parametrize("W, M", [(640, 4), (160, 4), (720, 4), (320, 4), (300, 8), (512, 2), (96, 3)])
def test_wave_per_row_zero_input_blocks_of_the_zero_pad_inverse(W, M):
    """Round 5 (fdoct_wave_dev.h::wave_zero_block): the inverse transform of the zero-pad stage has M W / 2 inputs of which
    elements W/2 .. M W / 2 - W/2 are zero by construction.  The first-pass blocks the rule picks lie inside that band, so the
    transform is the same whether they are read or taken as zero -- and whatever the buffer holds there (the re-packing no
    longer writes them)."""
    LH, WH = M * W // 2, W // 2
    if km.wave_plan(LH) is None:
        pytest.skip("no plan")
    zb = km.wave_zero_input_blocks(LH, WH, LH - WH)
    R, nb = km.wave_plan(LH)[0][0], LH // km.wave_plan(LH)[0][0]
    for r in zb:
        assert WH <= r * nb and (r + 1) * nb - 1 <= LH - WH
    if (W, M) == (640, 4):
        assert zb == list(range(5, 15))      # the fused radix 20: ten of twenty inputs
    rng = np.random.default_rng(W * M)
    x = rng.standard_normal(LH) + 1j * rng.standard_normal(LH)
    x[WH:LH - WH + 1] = 0
    full = km.wave_fft_inplace(x, inverse=True)
    skipped = km.wave_fft_inplace(x, inverse=True, zero_in=zb, poison=1e30)
    assert np.abs(skipped - full).max() == 0
