#!/bin/bash
# Build libs2p_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=libs2p_hip.so
SRCS="conv_igemm.hip wgrad_igemm.hip norm.hip misc.hip thin_conv.hip metrics.hip"
newest=$(ls -t $SRCS s2p_common.h ../../include/s2p_hip.h build.sh | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$newest" ]; then exit 0; fi
pids=()
for s in $SRCS; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -c "$s" -o "${s%.hip}.o" "$@" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" conv_igemm.o wgrad_igemm.o norm.o misc.o thin_conv.o metrics.o
echo "built $(pwd)/$OUT"
