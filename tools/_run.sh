set -e
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q > gpurun_out/diet_tests.log 2>&1 || { tail -40 gpurun_out/diet_tests.log; exit 1; }
tail -3 gpurun_out/diet_tests.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -k "train_step or parity or whole or production" > gpurun_out/diet_model_tests.log 2>&1 || { tail -40 gpurun_out/diet_model_tests.log; exit 1; }
tail -3 gpurun_out/diet_model_tests.log
echo "== step old"; S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_old.so timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -1
echo "== step new"; timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -1
echo "== step old"; S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_old.so timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -1
echo "== step new"; timeout -k 10 200 python tools/ab_step.py autograd_nodes.OVERLAP_VGG 3 True True 2>&1 | tail -1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5e_serial -- python3 $R/bench.py --no-cpu-baseline --serial-streams > $R/gpurun_out/prof_r5e_serial_bench.json 2> $R/gpurun_out/prof_r5e_serial.err
python3 $R/tools/kernel_summary.py $R/gpurun_out/prof_r5e_serial $R/gpurun_out/prof_r5e_serial_kernel_summary.csv 63
