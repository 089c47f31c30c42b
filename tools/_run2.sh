set -e
python -m pytest tests/test_kernels_gpu.py -x -q -k "head" > gpurun_out/r5_head_tests.log 2>&1 || { tail -40 gpurun_out/r5_head_tests.log; exit 1; }
tail -3 gpurun_out/r5_head_tests.log
python -m pytest tests/test_model_gpu.py -x -q -k "discriminator or train_step_losses_and_grads" > gpurun_out/r5_head_model_tests.log 2>&1 || { tail -40 gpurun_out/r5_head_model_tests.log; exit 1; }
tail -3 gpurun_out/r5_head_model_tests.log
D=s2p_amd/csrc/libs2p_hip_diag.so
S2P_LIB=$D python tools/ab_step.py lib:11 3 0 1 2>&1 | tail -3
S2P_LIB=$D python tools/ab_step.py lib:12 3 0 1 2>&1 | tail -3
