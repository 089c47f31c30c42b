#!/bin/bash
# Build libs2p_hip.so for gfx950 (cross-compiles without a GPU).
#   build.sh                 product library (no diagnostics, reads no environment variables)
#   build.sh diag            libs2p_hip_diag.so with -DS2P_DIAG_BUILD (timing ablations + A/B switches; select it with
#                            S2P_LIB=.../libs2p_hip_diag.so -- only tools/ do)
#   FORCE=1 build.sh         rebuild even when the library is newer than every source
set -e
cd "$(dirname "$0")"
OUT=libs2p_hip.so; SUF=""; EXTRA=""
if [ "$1" = "diag" ]; then OUT=libs2p_hip_diag.so; SUF=".diag"; EXTRA="-DS2P_DIAG_BUILD"; shift; fi
SRCS="conv_igemm.hip conv_plane.hip wgrad_igemm.hip wgrad_slab.hip wgrad_head.hip linear_small.hip norm.hip misc.hip thin_conv.hip thin_rows.hip metrics.hip"
newest=$(ls -t $SRCS s2p_common.h conv_plane.h ../../include/s2p_hip.h build.sh | head -1)
if [ -z "$FORCE" ] && [ -f "$OUT" ] && [ "$OUT" -nt "$newest" ]; then echo "up to date: $(pwd)/$OUT"; exit 0; fi
pids=(); objs=""
for s in $SRCS; do
  o="${s%.hip}${SUF}.o"; objs="$objs $o"
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $EXTRA -c "$s" -o "$o" "$@" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $objs
echo "built $(pwd)/$OUT"
