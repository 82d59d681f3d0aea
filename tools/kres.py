#!/usr/bin/env python3
"""Registers / scratch / occupancy of every fused_kernel instantiation of one translation unit, with the template arguments
decoded.  usage: tools/kres.py [hipcc -D flags ...]   e.g.  tools/kres.py -DFDOCT_ONLY_PLAN=5 -DFDOCT_ONLY_PRECT=1"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "fdoct_amd", "csrc", "fdoct_kernels.hip")
os.makedirs("/tmp/kres", exist_ok=True)
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *sys.argv[1:], "-save-temps", "-c", src,
       "-o", "/tmp/kres/k.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp/kres")
txt = out.stderr
names = ["LOG2NC", "T", "R1", "R2", "R3", "KIND", "WCH", "IN", "CPLX", "LEAN", "STAGE", "AVG", "IB2D", "NORM", "TRO", "PRECT"]
cur = None
rows = []
for ln in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, ln)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    n = r["name"]
    if "fused_kernel" not in n:
        continue
    args = re.findall(r"L[ib](\d+)E|([thf])(?=L|E)", n.split("fused_kernelI")[1].split("EEv")[0] + "E")
    vals = [a or b for a, b in args]
    desc = " ".join("%s=%s" % (k, v) for k, v in zip(names, vals) if k in ("KIND", "WCH", "IN", "CPLX", "LEAN", "STAGE", "AVG", "IB2D", "NORM", "TRO", "PRECT") and v not in ("0",) or k in ("KIND", "WCH"))
    print("%4d VGPR %5d B scratch  %s" % (r.get("vgpr", -1), r.get("scratch", -1), desc))
if out.returncode:
    print(txt[-3000:])
    sys.exit(out.returncode)
