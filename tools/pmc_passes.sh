#!/bin/bash
# Separate rocprofv3 --pmc passes over tools/pmc_resblk.py (counters only with --kernel-trace, never with other traces).
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$i -- python3 $GRAFT_REPO_ROOT/tools/pmc_resblk.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
fs = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$i/*/*counter_collection.csv")
if not fs: print("pass $i: no counter file"); raise SystemExit
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(fs[0])):
    k = (r["Kernel_Name"][:40], r["Counter_Name"])
    agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for (kn, cn), (v, n) in sorted(agg.items()):
    if "conv_halo" in kn or "wgrad" in kn: print("pass $i  %-42s %-32s avg/launch %.4g  (n=%d)" % (kn, cn, v / n, n))
PY
done
