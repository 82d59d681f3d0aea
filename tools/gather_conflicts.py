"""LDS bank conflicts of the lambda->k gather of the fused kernels, per staging layout (local, no GPU).

The gather is `ds_read_b32` with one address per lane: two 32-lane groups per instruction, bank = (byte address / 4) mod 32,
distinct addresses on one bank serialise (MI355X_MICROARCH.md, LDS).  For each BASELINE shape this counts the LDS cycles of
all the gather reads of one row against the conflict-free count, for the shipped even/odd split layout and for skewed
variants of it.  Result (round 2): the plain split is the best of them -- over 32 lanes the k-linear grid spans 59..69
samples (local slope 0.92..1.08), i.e. 33..35 slots of a plane for 32 banks, and a skew only moves the collisions.
usage: python tools/gather_conflicts.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402  (test infrastructure: the k tables)
from fdoct_amd import synth  # noqa: E402


def cycles(W, N, cplx, skew, oddshift, lmin=synth.LAMBDAMIN, lmax=synth.LAMBDAMAX):
    NC = N if cplx else N // 2
    idx, _ = orc.tables(W, 1, N, lmin, lmax)

    def off(i):
        v = i >> 1
        return 4 * (v + skew(v) + ((W // 2 + skew(W // 2 - 1) + oddshift) if (i & 1) else 0))
    tot = 0
    for m in range(NC // 64):
        for comp in range(1 if cplx else 2):
            addrs = []
            for lane in range(64):
                n = lane + 64 * m
                q = n if cplx else 2 * n + comp
                addrs.append(-4 if (q <= 0 or q >= N - 1) else off(int(idx[q])))
            for g in range(2):
                banks = {}
                for a in addrs[32 * g:32 * g + 32]:
                    banks.setdefault((a // 4) % 32, set()).add(a)
                tot += max(len(v) for v in banks.values())
    return tot, 2 * (NC // 64) * (1 if cplx else 2)


if __name__ == "__main__":
    layouts = [("even/odd split (shipped)", lambda v: 0), ("+4 slots per 32", lambda v: 4 * (v >> 5)), ("+4 slots per 64", lambda v: 4 * (v >> 6)),
               ("+8 slots per 32", lambda v: 8 * (v >> 5)), ("+16 slots per 32", lambda v: 16 * (v >> 5))]
    print("LDS cycles of one row's gather reads (conflict-free count in brackets)")
    for name, skew in layouts:
        for osh in (0, 4, 8, 16):
            c2, c4, c1 = cycles(2048, 2048, False, skew, osh), cycles(4096, 4096, False, skew, osh), cycles(1024, 1024, False, skew, osh)
            # (C3's complex rows gather one sample per FFT point, stride 1: they use the natural layout, conflict-free)
            print("%-26s odd plane +%2d:  C2 %3d (%d)  C4 %3d (%d)  C1 %3d (%d)" % (name, osh, *c2, *c4, *c1))
