import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# kernels compiled at run time (fdoct_set_jit) go to a directory of this test session (removed when the interpreter exits),
# not into the home directory
if "FDOCT_JIT_CACHE" not in os.environ:
    _jit_cache_dir = tempfile.TemporaryDirectory(prefix="fdoct_jit_tests_")
    os.environ["FDOCT_JIT_CACHE"] = _jit_cache_dir.name


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """The adjudication table of the run: for every comparison against an oracle output, the HIP result's and the f32
    restatement's distance from the fp64 evaluation of the chain, in units of the tolerance (helpers.check_mag / check_truth)."""
    try:
        import helpers
    except Exception:
        return
    if not helpers.TRUTH_LOG:
        return
    out = os.environ.get("FDOCT_TRUTH_TABLE", os.path.join(ROOT, "gpurun_out", "truth_table.txt"))
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        rows = helpers.TRUTH_LOG
        with open(out, "w") as f:
            f.write("# |x - truth| / tol per comparison against an oracle output; truth = the chain in double (oracle_truth)\n")
            f.write("# %d comparisons; worst gpu %.3f, worst f32 oracle %.3f; gpu beyond %.2g: %d\n" % (
                len(rows), max(r[2] for r in rows), max(r[3] for r in rows), helpers.TRUTH_LIMIT, sum(r[2] > helpers.TRUTH_LIMIT for r in rows)))
            f.write("# gpu      f32-oracle  test :: what\n")
            for t, what, g, o in rows:
                f.write("%8.4f  %8.4f    %s :: %s\n" % (g, o, t, what))
    except OSError:
        pass
