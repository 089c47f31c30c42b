#!/bin/bash
# The linear kernels' LDS read forms (s2p_amd/csrc/linear_small.hip, S2P_LIN_LDS_MODE builds) beside LDS-DMA conv launches
for m in 0 1 2 3 4; do
  if [ $m = 0 ]; then lib=$GRAFT_REPO_ROOT/s2p_amd/csrc/libs2p_hip.so; else lib=$GRAFT_REPO_ROOT/s2p_amd/csrc/libs2p_hip_lin$m.so; fi
  echo "== S2P_LIN_LDS_MODE=$m"
  S2P_LIB=$lib REPRO_ITERS=12 REPRO_ONLY="none,ResBlk fwd,down0 fwd,slab,IN fused" timeout -k 10 300 python tests/tools/repro_lds.py 2>/dev/null
done
