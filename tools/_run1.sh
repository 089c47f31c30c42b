set -e
python -m pytest tests/test_kernels_gpu.py -x -q -k "plane_kernels_production_shape or conv_plane_kernels_epilogues or plane_pair" > gpurun_out/r5_gst_tests.log 2>&1 || { tail -30 gpurun_out/r5_gst_tests.log; exit 1; }
tail -3 gpurun_out/r5_gst_tests.log
D=s2p_amd/csrc/libs2p_hip_diag.so
for v in 1 0 1 0; do
  echo "== switch7=$v (1 = no gamma|beta staging)"
  S2P_LIB=$D S2P_DIAG_SET="7=$v" python tools/bench_fused.py 2>&1 | tail -2
  S2P_LIB=$D S2P_DIAG_SET="7=$v" python tools/bench_gb_locality.py 2>&1 | grep -E "^N \(norm"
done
