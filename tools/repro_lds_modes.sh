#!/bin/bash
# The LDS-staged linear kernels' read forms (s2p_amd/csrc/linear_small.hip, diagnostics builds with -DS2P_LIN_LDS_MODE=m: the product
# kernels use no LDS since round 3; S2P_LIN_LDS=1 selects the LDS-staged ones) beside LDS-DMA conv launches.  Build the variants first:
#   for m in 1 2 3 4; do (cd s2p_amd/csrc && FORCE=1 bash build.sh diag -DS2P_LIN_LDS_MODE=$m && cp libs2p_hip_diag.so libs2p_hip_lin$m.so); done; (cd s2p_amd/csrc && FORCE=1 bash build.sh diag)
for m in 0 1 2 3 4; do
  if [ $m = 0 ]; then lib=$GRAFT_REPO_ROOT/s2p_amd/csrc/libs2p_hip_diag.so; else lib=$GRAFT_REPO_ROOT/s2p_amd/csrc/libs2p_hip_lin$m.so; fi
  echo "== S2P_LIN_LDS_MODE=$m"
  S2P_LIN_LDS=1 S2P_LIB=$lib REPRO_ITERS=12 REPRO_ONLY="none,ResBlk fwd,down0 fwd,slab,IN fused" timeout -k 10 300 python tests/tools/repro_lds.py 2>/dev/null
done
