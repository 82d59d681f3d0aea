"""numpy model of the data movement of the fused HIP kernel (fdoct_amd/csrc/fdoct_kernels.hip).

It mirrors, index for index, how a group of T lanes holding P = NC/T complex
points each runs the Stockham passes (registers -> twiddle -> radix butterfly ->
swizzled LDS -> natural read-back), the real-input untangle and its partner
mapping, and the LDS bank behaviour of every access pattern.  The GPU tests
check the kernel itself; this model lets the CPU-only suite check the scheme
(and was used to pick the swizzle shifts hard-coded in the kernel).
"""
import numpy as np


def swz(e, S):
    """LDS element swizzle used for an exchange: e ^ ((e >> S) & 15)."""
    return e ^ ((e >> S) & 15)


def stockham_lanes(z, T, radices, inverse=True, swizzles=None, record=None):
    """z: complex array of NC points in natural order.  Returns Z (natural order).

    Lane l holds elements e = l + T*m in register m (the 'natural' layout) at
    the start of every pass and at the end of the last one.
    """
    NC = z.shape[-1]
    P = NC // T
    assert P * T == NC and int(np.prod(radices)) == NC
    sgn = 1.0 if inverse else -1.0
    regs = np.zeros((T, P), complex)
    for l in range(T):
        for m in range(P):
            regs[l, m] = z[l + T * m]
    NS = 1
    for pi, R in enumerate(radices):
        assert P % R == 0
        last = pi == len(radices) - 1
        S = None if (last or swizzles is None) else swizzles[pi]
        lds = np.zeros(NC, complex)
        new = np.zeros_like(regs)
        for l in range(T):
            for t in range(P // R):
                j = l + T * t
                k = j % NS
                v = np.array([regs[l, t + r * (P // R)] for r in range(R)])
                tw = np.exp(sgn * 2j * np.pi * np.arange(R) * k / (NS * R))
                v = v * tw
                # R-point DFT
                V = np.array([np.sum(v * np.exp(sgn * 2j * np.pi * np.arange(R) * q / R)) for q in range(R)])
                if last:
                    for r in range(R):
                        new[l, t + r * (P // R)] = V[r]
                else:
                    base = (j // NS) * NS * R + k
                    for r in range(R):
                        e = base + r * NS
                        pe = e if S is None else swz(e, S)
                        if record is not None:
                            record.setdefault(("w", pi, t, r), []).append((l, pe))
                        lds[pe] = V[r]
        if not last:
            for l in range(T):
                for m in range(P):
                    e = l + T * m
                    pe = e if S is None else swz(e, S)
                    if record is not None:
                        record.setdefault(("r", pi, m), []).append((l, pe))
                    new[l, m] = lds[pe]
        regs = new
        NS *= R
    Z = np.zeros(NC, complex)
    for l in range(T):
        for m in range(P):
            Z[l + T * m] = regs[l, m]
    return Z


def untangle_partner(l, m, T, P):
    """(lane, reg) holding Z[(NC - e) mod NC] for e = l + T*m."""
    if l == 0:
        return 0, (P - m) % P
    return T - l, P - 1 - m


def untangle_real(Z, N):
    """X[e], e < N/2, of the unscaled inverse DFT of the real sequence x of
    length N, from Z = IDFT_{N/2}(x[2n] + i x[2n+1])."""
    NC = N // 2
    e = np.arange(NC)
    Zp = np.conj(Z[(NC - e) % NC])
    A = Z + Zp
    B = Z - Zp
    O = B / (2j)
    return 0.5 * A + np.exp(2j * np.pi * e / N) * O


def bank_conflicts(accesses, bytes_per_lane, group, nbanks):
    """accesses: list of (lane, element_index) for one wave instruction, element
    size = bytes_per_lane.  Lanes are serviced in `group`-lane contiguous
    groups; returns the worst-case conflict degree (1 = conflict free)."""
    worst = 1
    accesses = sorted(accesses)
    for g0 in range(0, len(accesses), group):
        grp = accesses[g0:g0 + group]
        per_bank = {}
        for _, e in grp:
            a = e * bytes_per_lane
            for d in range(bytes_per_lane // 4):
                b = ((a // 4) + d) % nbanks
                per_bank.setdefault(b, set()).add(a)
        worst = max(worst, max(len(s) for s in per_bank.values()))
    return worst


def fft1024_rowswap_model(x):
    """numpy model of fft1024_rowswap (fdoct_kernels.hip): unscaled inverse DFT of 1024 complex points
    with index split n = 64*m + 16*a + b and output k = k1 + 16*k2 + 64*k3.  Mirrors the kernel's
    register/lane placement: reg[lane][r]."""
    N = 1024
    W = lambda n, e: np.exp(2j * np.pi * e / n)
    reg = np.zeros((64, 16), complex)
    for lane in range(64):
        for m in range(16):
            reg[lane, m] = x[lane + 64 * m]
    # 1. radix-16 over the register index
    F16 = np.array([[W(16, m * k1) for m in range(16)] for k1 in range(16)])
    reg = reg @ F16.T
    # 2. 4x4 transpose (lane row a) <-> (k1 & 3) inside each register quad
    t = np.zeros_like(reg)
    for lane in range(64):
        a, b = lane >> 4, lane & 15
        for c in range(4):
            for d in range(4):
                # new register 4c+d at row a takes old register 4c+a at row d
                t[lane, 4 * c + d] = reg[16 * d + b, 4 * c + a]
    reg = t
    # 3. register 4c+i of lane (j,b) = A[k1=4c+j][a=i][b]; twiddle W_64^(i*(4c+j)), radix-4 over i
    lds = np.zeros(65 * 16 + 2, complex)
    for lane in range(64):
        j, b = lane >> 4, lane & 15
        for c in range(4):
            v = np.array([reg[lane, 4 * c + i] * W(64, i * (4 * c + j)) for i in range(4)])
            out = np.array([sum(v[i] * W(4, i * k2) for i in range(4)) for k2 in range(4)])
            for k2 in range(4):
                lds[65 * b + j + 4 * c + 16 * k2] = out[k2]
    # 4./5. read back, twiddle W_1024^(b*l'), radix-16 over b
    X = np.zeros(N, complex)
    for lane in range(64):
        v = np.array([lds[65 * bb + lane] * W(1024, bb * lane) for bb in range(16)])
        for k3 in range(16):
            X[lane + 64 * k3] = sum(v[bb] * W(16, bb * k3) for bb in range(16))
    return X


def fft2048_rowswap_model(x):
    """numpy model of fft2048_rowswap (fdoct_kernels.hip): unscaled inverse DFT of 2048 complex points, index split
    n = 64*m + 16*a + b (m: 32 registers, a: 16-lane row, b: lane in row), output k = k1 + 32*k2 + 128*k3.
    Final register s + 2*k3 of lane l holds bin l + 64*(s + 2*k3) (natural slot order)."""
    N = 2048
    W = lambda n, e: np.exp(2j * np.pi * e / n)
    reg = np.zeros((64, 32), complex)
    for lane in range(64):
        for m in range(32):
            reg[lane, m] = x[lane + 64 * m]
    F32 = np.array([[W(32, m * k1) for m in range(32)] for k1 in range(32)])
    reg = reg @ F32.T                                            # 1. radix-32 over the register index
    t = np.zeros_like(reg)
    for lane in range(64):                                       # 2. (row a) <-> (k1 & 3) in each of the 8 register quads
        a, b = lane >> 4, lane & 15
        for c in range(8):
            for d in range(4):
                t[lane, 4 * c + d] = reg[16 * d + b, 4 * c + a]
    reg = t
    S = 129
    lds = np.zeros(S * 16 + 2, complex)
    for lane in range(64):                                       # 3. twiddle W_128^(i*k1), radix-4 over i; 4. exchange
        j, b = lane >> 4, lane & 15
        for c in range(8):
            v = np.array([reg[lane, 4 * c + i] * W(128, i * (4 * c + j)) for i in range(4)])
            for k2 in range(4):
                lds[S * b + 4 * c + j + 32 * k2] = sum(v[i] * W(4, i * k2) for i in range(4))
    X = np.zeros(N, complex)
    for lane in range(64):                                       # 5. twiddle W_2048^(b*l'), radix-16 over b, two l' per lane
        for s in range(2):
            lp = lane + 64 * s
            v = np.array([lds[S * bb + lp] * W(2048, bb * lp) for bb in range(16)])
            for k3 in range(16):
                X[lane + 64 * (s + 2 * k3)] = sum(v[bb] * W(16, bb * k3) for bb in range(16))
    return X


# ---------------------------------------------------------------------------------------------------------------
# Wave-per-row kernels (fdoct_amd/csrc/fdoct_wave.hip): the radix plan, the per-pass twiddle tables the host builds
# (fdoct_capi.cpp::rebuild_wave_state) and the in-place Stockham passes with clamped partial rounds.
def wave_plan(n):
    """fdoct_wave.h::wave_plan: [(R, Ns)], or None when n has a prime factor above 5."""
    plan, ns = [], 1

    def push(r):
        nonlocal ns
        plan.append((r, ns))
        ns *= r
    n_all = n
    if n % 1280 == 0:
        push(20)
        n //= 20
    while n % 15 == 0 and n % 9 != 0 and n_all >= 240:   # one in-register 3 x 5 pass (FDOCT_WAVE_R15_MIN)
        push(15)
        n //= 15
    while n % 5 == 0:
        push(5)
        n //= 5
    while n % 9 == 0:                                     # one in-register 3 x 3 pass
        push(9)
        n //= 9
    while n % 3 == 0:
        push(3)
        n //= 3
    # the power of two that is left: as many radix-16 passes as make the pass count smaller (not below 512 points), then 8s
    bits, m = 0, n
    while m > 1 and m % 2 == 0:
        bits, m = bits + 1, m // 2
    n16, fewest, a16 = 0, 1 << 20, 0
    while 4 * a16 <= bits and (a16 == 0 or n_all >= 512):
        passes = a16 + (bits - 4 * a16 + 2) // 3
        if passes < fewest:
            fewest, n16 = passes, a16
        a16 += 1
    for _ in range(n16):
        push(16)
        n //= 16
    while n % 8 == 0:
        push(8)
        n //= 8
    if n % 4 == 0:
        push(4)
        n //= 4
    if n % 2 == 0:
        push(2)
        n //= 2
    return plan if n == 1 else None


def wave_pass_tables(n):
    """One table per pass with Ns > 1: exp(+2 pi i k / (Ns R)), k < Ns, concatenated as the host uploads them."""
    out = []
    for R, Ns in wave_plan(n):
        if Ns > 1:
            out.append(np.exp(2j * np.pi * np.arange(Ns) / (Ns * R)))
    return np.concatenate(out) if out else np.zeros(0, complex)


def wave_depth_bound(n, D):
    """fdoct_wave.h::wave_depth_bound for a transform of n complex points: D rounded up to a whole output block of the last pass."""
    R, Ns = wave_plan(n)[-1]
    if D >= n:
        return n
    return min(n, -(-D // Ns) * Ns)


def wave_dead_output_blocks(n, DK, DKH):
    """fdoct_wave_dev.h::wave_pass, NEVER: blocks r of the last pass (outputs r Ns .. (r + 1) Ns - 1) that no depth <= DK reads
    when the store filter keeps e < D or e > n - D (real rows: DKH = n - DK) or e < D (DKH = n)."""
    R, Ns = wave_plan(n)[-1]
    return [r for r in range(R) if r * Ns >= DK and (r + 1) * Ns - 1 <= DKH]


def wave_zero_input_blocks(n, zlo, zhi):
    """fdoct_wave_dev.h::wave_zero_block: blocks r of the FIRST pass (inputs r nb .. (r + 1) nb - 1) that lie inside zlo .. zhi."""
    R, _ = wave_plan(n)[0]
    nb = n // R
    return [r for r in range(R) if zhi >= zlo and r * nb >= zlo and (r + 1) * nb - 1 <= zhi]


def wave_fft_inplace(x, inverse=True, keep=None, dead_out=(), zero_in=(), poison=None):
    """The n-point transform as ONE wave runs it: per pass every lane first reads its butterflies' inputs (rounds of 64
    butterflies, the last round's idle lanes repeat butterfly nb-1 when n >= 640), then writes the outputs back into the
    same buffer; twiddles = one table entry per butterfly, powers by products.  keep(e) filters the last pass's stores.
    dead_out: blocks of the last pass that are not stored at all (compile-time pruning); zero_in: blocks of the first pass
    that are not read (taken as zero) -- poison, if given, is what the buffer holds there instead (the re-packing does not write them)."""
    n = len(x)
    buf = np.array(x, complex)
    if poison is not None and zero_in:
        nb0 = n // wave_plan(n)[0][0]
        for r in zero_in:
            buf[r * nb0:(r + 1) * nb0] = poison
    tables = wave_pass_tables(n)
    plan = wave_plan(n)
    toff = 0
    for p, (R, Ns) in enumerate(plan):
        nb = n // R
        rounds = (nb + 63) // 64
        clamp = n >= 640
        loaded = {}
        for t in range(rounds):
            for lane in range(64):
                j = lane + 64 * t
                if j >= nb and not clamp:
                    continue
                jc = min(j, nb - 1)
                loaded[(t, lane)] = (j, jc, np.array([0.0 if (p == 0 and r in zero_in) else buf[jc + r * nb] for r in range(R)]))
        new = buf.copy()
        for (t, lane), (j, jc, v) in loaded.items():
            q, k = (jc // Ns, jc % Ns) if Ns > 1 else (jc, 0)
            if Ns > 1:
                w1 = tables[toff + k]
                if not inverse:
                    w1 = np.conj(w1)
                v = v * w1 ** np.arange(R)
            sgn = 1.0 if inverse else -1.0
            V = np.array([np.sum(v * np.exp(sgn * 2j * np.pi * np.arange(R) * s / R)) for s in range(R)])
            if j < nb:   # idle lanes of a partial round computed a duplicate: their stores are masked off
                for r in range(R):
                    e = q * Ns * R + k + r * Ns
                    if p == len(plan) - 1 and r in dead_out:
                        continue
                    if p < len(plan) - 1 or keep is None or keep(e):
                        new[e] = V[r]
        buf = new
        if Ns > 1:
            toff += Ns
    return buf
