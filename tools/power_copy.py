"""Package power while the GPU streams HBM (device-to-device copy of 1 GiB, a read-only reduction, a fill): the energy of a
byte moved, next to tools/power_probe.sh's energy of an instruction.  Run on the GPU box; prints GB/s and rocm-smi power."""
import subprocess
import sys
import threading
import time

import torch

n = 1 << 30
src = torch.empty(n, dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)
src4 = src.view(torch.float32)


def sample(out):
    time.sleep(2.5)
    out.append(subprocess.run("rocm-smi --showpower --showclocks | grep -E 'sclk|Package Power' | sed 's/GPU\\[0\\]//; s/\\t//g' | tr '\\n' ' '",
                              shell=True, capture_output=True, text=True).stdout)


for name, fn, nbytes in (("copy 1 GiB (read + write)", lambda: dst.copy_(src), 2 * n), ("sum 1 GiB (read only)", lambda: src4.sum(), n),
                         ("fill 1 GiB (write only)", lambda: dst.fill_(3), n)):
    out = []
    th = threading.Thread(target=sample, args=(out,))
    th.start()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 5.0:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        k += 50
    dt = time.perf_counter() - t0
    th.join()
    print("%-28s %7.0f GB/s | %s" % (name, nbytes * k / dt / 1e9, out[0].strip()))
    sys.stdout.flush()
