set -e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > gpurun_out/dma_tests.log 2>&1 || { tail -30 gpurun_out/dma_tests.log; exit 1; }
tail -3 gpurun_out/dma_tests.log
for f in convT down "D0" "D1 " "D2 " "shared"; do
  echo "== old $f"; S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_old.so timeout -k 10 120 python tools/bench_conv.py "$f" 2>&1 | grep "|" 
  echo "== new $f"; timeout -k 10 120 python tools/bench_conv.py "$f" 2>&1 | grep "|"
done
echo "== step old"; S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_old.so timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -2
echo "== step new"; timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -2
echo "== step old"; S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_old.so timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -2
echo "== step new"; timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -2
