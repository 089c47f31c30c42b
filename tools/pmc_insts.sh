#!/bin/bash
# Instruction mix per (kernel, grid): VALU / SALU / MFMA / LDS / VMEM instructions and waves per launch (one --pmc pass over a short
# eager, serial-stream bench).  Which kernels spend their issue slots outside the MFMA loop?  -> gpurun_out/pmc_insts_TAG.json
TAG=${1:-r5}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc_avail.txt 2>&1
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmci_${TAG}_$i -- python3 $R/bench.py --steps 2 --warmup 1 \
    --no-graph --no-cpu-baseline --no-roofline --serial-streams > /dev/null 2> $R/gpurun_out/pmci_${TAG}_$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, json, collections
R = "$R"; TAG = "$TAG"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sorted(glob.glob(R + "/gpurun_out/pmci_%s_*/" % TAG)):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]
            key = name + " | grid " + r["Grid_Size"]
            a = acc[key][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for k, cs in acc.items():
    e = {"launches": max(v[1] for v in cs.values())}
    for c, (v, n) in cs.items():
        e[c] = v / n
    out[k] = e
json.dump(out, open(R + "/gpurun_out/pmc_insts_%s.json" % TAG, "w"), indent=1, sort_keys=True)
rows = sorted(out.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0) * kv[1]["launches"])
print("%-70s %8s %10s %10s %9s %9s" % ("kernel | grid", "launches", "VALU/launch", "SALU", "MFMA", "waves"))
for k, e in rows[:45]:
    print("%-70s %8d %10.0f %10.0f %9.0f %9.0f" % (k[:70], e["launches"], e.get("SQ_INSTS_VALU", 0), e.get("SQ_INSTS_SALU", 0), e.get("SQ_INSTS_MFMA", 0), e.get("SQ_WAVES", 0)))
PY
