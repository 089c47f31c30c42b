#!/bin/bash
# per-dispatch kernel summary of the default bench command with every launch on ONE stream (kernel-alone durations)
R=$GRAFT_REPO_ROOT; TAG=${1:-r3a}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_serial -- python3 $R/bench.py --no-cpu-baseline --serial-streams > $R/gpurun_out/prof_${TAG}_serial_bench.json 2> $R/gpurun_out/prof_${TAG}_serial.err
python3 $R/tools/kernel_summary.py $R/gpurun_out/prof_${TAG}_serial $R/gpurun_out/prof_${TAG}_serial_kernel_summary.csv 63
