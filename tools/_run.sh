set -e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad or conv_fwd_dgrad or groups" > gpurun_out/wd_tests.log 2>&1 || { tail -40 gpurun_out/wd_tests.log; exit 1; }
tail -2 gpurun_out/wd_tests.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -k "generator_forward_backward or edge_shapes" > gpurun_out/wd_mtests.log 2>&1 || { tail -40 gpurun_out/wd_mtests.log; exit 1; }
tail -2 gpurun_out/wd_mtests.log
