set -e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_fwd_dgrad_wgrad" > gpurun_out/r7_tests.log 2>&1 || { tail -40 gpurun_out/r7_tests.log; exit 1; }
tail -2 gpurun_out/r7_tests.log
for f in "stem 7x7" "out 7x7"; do
  echo "== old $f"; S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_old.so timeout -k 10 120 python tools/bench_conv.py "$f" 2>&1 | grep "|"
  echo "== new $f"; timeout -k 10 120 python tools/bench_conv.py "$f" 2>&1 | grep "|"
done
