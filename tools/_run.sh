set -e
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q > gpurun_out/diet_tests.log 2>&1 || { tail -40 gpurun_out/diet_tests.log; exit 1; }
tail -3 gpurun_out/diet_tests.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -k "not rollout" > gpurun_out/diet_model_tests.log 2>&1 || { tail -40 gpurun_out/diet_model_tests.log; exit 1; }
tail -3 gpurun_out/diet_model_tests.log
echo "== step old"; S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_old.so timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -1
echo "== step new"; timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -1
echo "== step old"; S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_old.so timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -1
echo "== step new"; timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -1
bash tools/pmc_insts.sh r5c > gpurun_out/pmc_insts_c.log 2>&1
head -3 gpurun_out/pmc_insts_c.log
