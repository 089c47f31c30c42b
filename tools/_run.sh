set -e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad or conv_fwd_dgrad" > gpurun_out/slab_tests.log 2>&1 || { tail -40 gpurun_out/slab_tests.log; exit 1; }
tail -2 gpurun_out/slab_tests.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -k "discriminator or train_step" > gpurun_out/slab_mtests.log 2>&1 || { tail -40 gpurun_out/slab_mtests.log; exit 1; }
tail -2 gpurun_out/slab_mtests.log
export S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_diag.so
echo "== table"; timeout -k 10 200 python tools/bench_wgrad_strided.py 2>&1 | tail -14
echo "== state (switch 18)"; S2P_DIAG_SET="18=1" timeout -k 10 200 python tools/bench_wgrad_strided.py 2>&1 | tail -14
echo "== step A/B lib:18"; timeout -k 10 300 python tools/ab_step.py lib:18 3 1 0 2>&1 | tail -2
