set -e
python -m pytest tests/test_kernels_gpu.py -x -q -k "head or planeg or conv_fwd or conv_dgrad" > gpurun_out/r5_t4.log 2>&1 || { tail -40 gpurun_out/r5_t4.log; exit 1; }
tail -2 gpurun_out/r5_t4.log
D=s2p_amd/csrc/libs2p_hip_diag.so
echo "== new forms"; S2P_LIB=$D python tools/bench_dhead.py 2>&1 | grep "^N"
