#!/bin/bash
# Build libs2p_hip.so for gfx950 (cross-compiles without a GPU).
#   build.sh                 product library (no diagnostics, reads no environment variables)
#   build.sh diag            libs2p_hip_diag.so with -DS2P_DIAG_BUILD (timing ablations + A/B switches; select it with
#                            S2P_LIB=.../libs2p_hip_diag.so -- only tools/ do)
#   FORCE=1 build.sh         rebuild even when the library is newer than every source
# -fvisibility=hidden: only the C entry points declared in include/s2p_hip.h (visibility pragma there) are exported.
# -target-feature -packed-fp32-ops: NO packed fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any kernel.  Round 4
#   found that such an instruction with an op_sel operand swizzle (hipcc forms them when it SLP-vectorises fp32 arithmetic) returns
#   wrong results in lanes 48..63 while the wave shares a SIMD with the slab weight-gradient or the LDS-DMA conv kernel (DESIGN.md
#   section 4; tests/tools/repro_valu_probe.py).  tests/test_host_logic.py::test_no_packed_fp32_instructions audits the ISA.
# -Wno-inline-asm: the LDS-DMA helpers declare M0 clobbered, which hipcc reports as "reserved register on the clobber list".
set -e
cd "$(dirname "$0")"
OUT=libs2p_hip.so; SUF=""; EXTRA=""
DIAG_SRCS=""
if [ "$1" = "diag" ]; then OUT=libs2p_hip_diag.so; SUF=".diag"; EXTRA="-DS2P_DIAG_BUILD"; DIAG_SRCS="diag_probe.hip"; shift; fi
SRCS="conv_igemm.hip conv_plane.hip conv_planeg.hip wgrad_igemm.hip wgrad_slab.hip wgrad_slabg.hip wgrad_head.hip linear_small.hip norm.hip misc.hip thin_conv.hip thin_rows.hip metrics.hip"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
newest=$(ls -t $SRCS $DIAG_SRCS s2p_common.h conv_plane.h conv_planeg.h ../../include/s2p_hip.h build.sh | head -1)
if [ -z "$FORCE" ] && [ -f "$OUT" ] && [ "$OUT" -nt "$newest" ]; then echo "up to date: $(pwd)/$OUT"; exit 0; fi
pids=(); objs=""
for s in $SRCS $DIAG_SRCS; do
  o="${s%.hip}${SUF}.o"; objs="$objs $o"
  pk="$NOPK"; if [ "$s" = "diag_probe.hip" ]; then pk=""; fi       # the probe file executes packed fp32 instructions on purpose
  rm -f "$o"
  hipcc --offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -std=c++17 -Wno-unused-value -Wno-inline-asm $pk $EXTRA -c "$s" -o "$o" "$@" \
    2> >(grep -v "is not a recognized feature for this target" >&2) &      # (the host half of the compile does not know the device feature)
  pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
for o in $objs; do [ -f "$o" ] || fail=1; done
if [ $fail -ne 0 ]; then echo "build.sh: a compile failed" >&2; exit 1; fi
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $objs
echo "built $(pwd)/$OUT"
