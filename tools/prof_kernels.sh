#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python command + per-dispatch summary.  Usage: tools/prof_kernels.sh TAG STEPS cmd...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; STEPS=$2; shift 2
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG -- "$@" > $R/gpurun_out/prof_$TAG.out 2> $R/gpurun_out/prof_$TAG.err
cd $R && python3 tools/kernel_summary.py gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_summary.csv $STEPS
