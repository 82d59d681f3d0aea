#!/bin/bash
# One script for a round's GPU calls (it replaces round 5's r5_*.sh one-offs).  usage, always through gpurun:
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh suite'          the whole GPU suite as the driver runs it + the default and
#                                                                     the driver-sized bench line; writes gpurun_out/truth_table.txt
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh fuzz [set]'     the seeded sweeps of one set (first | other | final | fresh | fresh2 | fresh3 | bandpass), FIVE processes side by side
#                                                                     (a sweep is bound by the CPU oracle, not by the GPU)
#   gpurun --timeout 900  -- 'bash tools/gpu_round.sh tests <pytest args>'   a part of the suite
#   gpurun --timeout 900  -- 'bash tools/gpu_round.sh wl C1 C3 C4 INI'       bench lines of other workloads (both layouts)
cd "$(dirname "${BASH_SOURCE[0]}")/.." || exit 1
mkdir -p gpurun_out
tag=${TAG:-r6}
what=${1:-suite}; shift
line() { python3 - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
r = d['roofline']
print(d['config']['workload'][:60], '| value %.4g' % d['value'], '| ms/step %.4f' % d['ms_per_step'], '| frac %.4f' % r['frac'],
      '| achievable %s' % r.get('frac_of_achievable'), '| parity', json.dumps(d.get('parity', {}))[:600])
PY
}
case $what in
  suite)
    FDOCT_TRUTH_TABLE=gpurun_out/${tag}_truth_table.txt timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
    rc=$?; echo "pytest exit $rc" >> gpurun_out/${tag}_tests.log; tail -8 gpurun_out/${tag}_tests.log
    head -3 gpurun_out/${tag}_truth_table.txt
    [ $rc -eq 0 ] || exit $rc
    python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && line gpurun_out/${tag}_bench.json &&
    python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver.json 2>/dev/null && line gpurun_out/${tag}_bench_driver.json ;;
  tests)
    FDOCT_TRUTH_TABLE=gpurun_out/${tag}_truth_table_part.txt timeout -k 10 800 python3 -m pytest -x -q -m gpu "$@" > gpurun_out/${tag}_tests_part.log 2>&1
    rc=$?; tail -15 gpurun_out/${tag}_tests_part.log; exit $rc ;;
  wl)
    for w in "$@"; do
      for lay in rowmajor transposed; do
        python3 bench.py --workload $w --layout $lay --no-cpu-baseline > gpurun_out/${tag}_wl_${w}_${lay}.json 2>/dev/null && line gpurun_out/${tag}_wl_${w}_${lay}.json || echo "$w $lay failed"
      done
    done ;;
  fuzz)
    set_=${1:-first}
    out=gpurun_out/${tag}_fuzz_${set_}_summary.txt
    # <seed> <cases> <shares: jit, long rows, forced routes, weak frames, tall frames, device pointers, reuse>
    case $set_ in
      first) runs=("6101 260 0.2 0.1 0.2 0.3 0.2 0.2 0.2" "6102 260 0.2 0.1 0.2 0.3 0.2 0.2 0.2" "5501 350 0.2 0.1 0.2 0.3 0.2 0.2 0.2" "5401 300 0.7 0 0.2 0.3 0.2 0.2 0.2" "5601 300 0.2 0.1 0.2 0.3 0.2 0.2 0.2") ;;
      other) runs=("6201 300 0.5 0 0 0.3 0 0 0" "6202 120 0 0.6 0 0.2 0 0 0" "6203 300 0 0 0.5 0.5 0.3 0.3 0.3" "6204 300 0.3 0 0.5 0.3 0.5 0.3 0.3" "6205 300 0.2 0.1 0.3 0.3 0.2 0.2 0.2") ;;
      final) runs=("6301 280 0.2 0.1 0.2 0.3 0.2 0.2 0.2" "6302 280 0.2 0.1 0.2 0.3 0.2 0.2 0.2" "6303 280 0.5 0 0.3 0.3 0.2 0.2 0.2" "6304 280 0.1 0.2 0.3 0.3 0.3 0.3 0.3" "6305 280 0.3 0 0.5 0.5 0.3 0.3 0.3") ;;
      fresh) runs=("6501 280 0.3 0.1 0.3 0.3 0.2 0.2 0.2" "6502 280 0.5 0 0.4 0.4 0.2 0.3 0.2" "6503 280 0.1 0.3 0.3 0.2 0.3 0.2 0.3" "6504 280 0.3 0 0.6 0.5 0.3 0.3 0.3" "6505 280 0.2 0.1 0.2 0.6 0.4 0.2 0.2") ;;
      fresh2) runs=("6601 280 0.3 0.1 0.3 0.4 0.4 0.2 0.2" "6602 280 0.4 0 0.5 0.5 0.3 0.3 0.2" "6603 280 0.1 0.3 0.3 0.3 0.4 0.2 0.3" "6604 280 0.2 0 0.6 0.6 0.4 0.3 0.3" "6605 280 0.2 0.1 0.3 0.7 0.5 0.2 0.2") ;;
      fresh3) runs=("6701 280 0.2 0.1 0.4 0.7 0.6 0.2 0.2" "6702 280 0.3 0 0.5 0.7 0.6 0.3 0.2" "6703 280 0.1 0.2 0.3 0.5 0.6 0.2 0.3" "6704 280 0.6 0 0.3 0.6 0.3 0.3 0.3" "6705 280 0.2 0.1 0.6 0.4 0.5 0.4 0.4") ;;
      bandpass) export FDOCT_FUZZ_ROUTE=band-pass   # every routed case gets BscanDark's band-pass (it takes effect under a zero-pad multiplier)
                runs=("6401 260 0.4 0 1 0.4 0.1 0.2 0.1" "6402 260 0.6 0 1 0.5 0.1 0.2 0.1" "6403 260 0.2 0.1 1 0.3 0.2 0.2 0.2" "6404 260 0.5 0 1 0.5 0 0 0" "6405 260 0.3 0 1 0.2 0.3 0.3 0.3") ;;
      *) echo "unknown set $set_"; exit 2 ;;
    esac
    : > $out
    pids=()
    for r in "${runs[@]}"; do
      seed=${r%% *}
      ( OMP_NUM_THREADS=1 timeout -k 10 ${LIMIT:-1050} python3 tools/fuzz_parity.py $r > gpurun_out/${tag}_fuzz_$seed.log 2>&1; echo "exit $?" >> gpurun_out/${tag}_fuzz_$seed.log ) &
      pids+=($!)
    done
    # progress lines so that the call is not taken for hung
    while :; do
      alive=0; for p in "${pids[@]}"; do kill -0 $p 2>/dev/null && alive=1; done
      [ $alive -eq 0 ] && break
      sleep 60; wc -l gpurun_out/${tag}_fuzz_*.log | tail -1
    done
    for r in "${runs[@]}"; do
      seed=${r%% *}
      echo "== $r" >> $out
      grep -E "^FAIL|^truth |failures:|^exit" gpurun_out/${tag}_fuzz_$seed.log | tail -14 >> $out
    done
    cat $out ;;
  *) echo "unknown target $what"; exit 2 ;;
esac
