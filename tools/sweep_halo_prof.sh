#!/bin/bash
# kernel-only durations (rocprofv3) of the halo conv sweep for a few S2P_DIAG ablations
cd /tmp && export TMPDIR=/tmp
for d in 0 3 5 4; do
  S2P_LIB=$GRAFT_REPO_ROOT/s2p_amd/csrc/libs2p_hip_diag.so S2P_DIAG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sw_$d -- python3 $GRAFT_REPO_ROOT/tools/sweep_halo.py > /dev/null 2>&1
  echo "DIAG=$d"; python3 - <<PY
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/sw_$d/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "conv_halo" in r["Kernel_Name"]]
import collections
agg = collections.OrderedDict()
for r in rows:
    k = (r["Grid_Size_X"],)
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
# launches come in groups of 35 per (N, cin) combination in sweep order
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
for i in range(0, len(durs), 35):
    seg = sorted(durs[i:i + 35]); print("  combo %d: median %.1f us  min %.1f" % (i // 35, seg[len(seg) // 2], seg[0]))
PY
done
