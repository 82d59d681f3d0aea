cd /root/repo
timeout -k 10 400 bash tools/gpu_round.sh tests tests/test_gpu_parity.py tests/test_gpu_jit.py -k "generic or long_rows or any_numfft or fall_back or workgroup or weak_fringes_on" || exit 1
for t in 0 1 0 1; do echo "== FDOCT_GENERIC_TICKETS=$t"; FDOCT_GENERIC_TICKETS=$t python3 tools/odd_width.py 2>/dev/null | grep "^W="; done
