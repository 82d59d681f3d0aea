cd "$GRAFT_REPO_ROOT"
for v in o3 o4 o5 o6 r16; do cp fdoct_amd/libfdoct_hip_$v.so fdoct_amd/libfdoct_hip.so; echo $v; timeout -k 10 100 python tools/bench_generic.py 2>&1 | tail -1; done
