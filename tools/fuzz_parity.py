"""Randomised parity sweep (seeded): random geometries and option combinations through whatever kernel the library
selects, each against the oracle.  python tools/fuzz_parity.py [seed] [count] [jit_share] [big_share] [route_share] [weak_share] [tall_share] [dev_share] [reuse_share]
(jit_share: fraction of cases drawn as geometries for the run-time compiled wave-per-row kernel, default 0;
big_share: fraction drawn as long rows -- 4000 ... 65536 points: the 512- / 1024-thread workgroup-per-row kernels and the
long-row path --, default 0; route_share: fraction pushed onto another route than the library's own choice -- staged mode, the
fused any-option kernel, the workgroup-per-row kernel, no run-time compilation, a launch of one to three workgroups --, default 0; weak_share: fraction run on weakly modulated frames, fringes of 2 % or 0.1 % of the DC
level, default 0; tall_share: fraction with 60 ... 400 lines per frame and three or four B-scans per call, default 0; dev_share: fraction through the device-pointer entry point with padded row pitches and
addresses off the 16-byte grid, default 0; reuse_share: fraction whose handle runs a second time after a setter changed its route, default 0)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
jit_share = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
big_share = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
route_share = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
weak_share = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
tall_share = float(sys.argv[7]) if len(sys.argv) > 7 else 0.0
dev_share = float(sys.argv[8]) if len(sys.argv) > 8 else 0.0
reuse_share = float(sys.argv[9]) if len(sys.argv) > 9 else 0.0
stats = {}
fails = fuzz_cases.run_sweep(seed, count, stats=stats, jit_share=jit_share, big_share=big_share, route_share=route_share, weak_share=weak_share,
                             tall_share=tall_share, dev_share=dev_share, reuse_share=reuse_share)
print("failures: %d; adjudicated by the exact chain (outside the tolerance against the f32 restatement, no farther from truth than it): %d of %d cases; "
      "worst |gpu - truth| / tol %.3f, worst |f32 oracle - truth| / tol %.3f, cases with gpu beyond 0.5: %d; %d ran a run-time compiled kernel%s" % (
    fails, stats.get("by_truth", 0), stats.get("ran", 0), stats.get("worst_gpu_truth", 0.0), stats.get("worst_f32_truth", 0.0), stats.get("over_half", 0), stats.get("jit", 0),
    (("; long rows by kernel family (fdoct_kernel): %s" % dict(sorted(stats.get("families", {}).items()))) if big_share else "") +
    (("; forced routes: %s" % dict(sorted(stats.get("routes", {}).items()))) if route_share else "")))
# (exit code: a failure, or more than 2 % of the cases with the HIP result itself beyond 0.5 x the tolerance from the chain in double --
# adjudications where the f32 restatement is the one that is far from it, as under BscanDark's band-pass, do not count)
sys.exit(1 if fails or stats.get("over_half", 0) > max(2, stats.get("ran", 0) // 50) else 0)
