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
