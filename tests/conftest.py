import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# kernels compiled at run time (fdoct_set_jit) go to a directory of this test session, not into the home directory
os.environ.setdefault("FDOCT_JIT_CACHE", tempfile.mkdtemp(prefix="fdoct_jit_tests_"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
