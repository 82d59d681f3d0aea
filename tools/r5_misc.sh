#!/bin/bash
# round 5 evidence, part 2: phase probes, transposed traffic, soaks, odd-width rate
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
bash tools/fused_probe.sh > /dev/null 2>&1
FDOCT_LIB=$PWD/fdoct_amd/libfdoct_hip_probe5.so python3 bench.py --layout transposed --steps 300 --warmup 20 --no-cpu-baseline --half-chip-steps 0 --sustained-seconds 0 --stage-steps 0 --precise-steps 0 2> gpurun_out/fp_t.err > /dev/null
{ echo "== C2, transposed store (probe5) both words (default)"; grep "fused probe" gpurun_out/fp_t.err | tail -1 | tr '|' '\n' | sed 's/^ */   /'; } >> gpurun_out/fused_probe.txt
echo "probe done"
bash tools/prof_traffic.sh r05t --layout transposed > gpurun_out/r5_traffic_t.log 2>&1; tail -3 gpurun_out/r5_traffic_t.log
python3 bench.py --layout transposed --steps 200 --sustained-seconds 60 --no-cpu-baseline --half-chip-steps 0 --stage-steps 0 > gpurun_out/r5_soak_transposed.json 2> gpurun_out/r5_soak.err; echo "soak1 $?"
python3 bench.py --layout transposed --lines-per-frame 500 --display-points 512 --steps 200 --sustained-seconds 45 --no-cpu-baseline --half-chip-steps 0 --stage-steps 0 > gpurun_out/r5_soak_transposed_h500_d512.json 2>> gpurun_out/r5_soak.err; echo "soak2 $?"
python3 bench.py --layout transposed --background-2d --steps 200 --sustained-seconds 30 --no-cpu-baseline --half-chip-steps 0 --stage-steps 0 > gpurun_out/r5_soak_transposed_bg2d.json 2>> gpurun_out/r5_soak.err; echo "soak3 $?"
python3 - <<'PY' > gpurun_out/r5_odd_width.txt 2>&1
import time, numpy as np, sys
sys.path.insert(0, "tests")
from fdoct_amd import Config, Reconstructor, synth, capi
for (W, M, N, D, H) in [(321, 4, 1284, 320, 240), (161, 4, 2560, 320, 240), (225, 3, 1024, 300, 240), (320, 4, 1280, 320, 240)]:
    cfg = Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, increasefftpointsmultiplier=M)
    r = Reconstructor(cfg); r.set_background(synth.make_background(max(W, 64))[:W].astype(np.float64) + 10)
    fr = synth.make_frames(0, 8, max(W, 64), H)[:, :, :W].copy()
    r.process(fr)
    t0 = time.perf_counter()
    for _ in range(5): r.process(fr)
    dt = (time.perf_counter() - t0) / 5
    print("W=%d M=%d N=%d D=%d: kernel family %d, %.3g A-scans/s through fdoct_process (host buffers, %d A-scans per call)" % (W, M, N, D, r.last_kernel(), fr.shape[0] * H / dt, fr.shape[0] * H))
    r.close()
PY
cat gpurun_out/r5_odd_width.txt
cat gpurun_out/fused_probe.txt | head -80
