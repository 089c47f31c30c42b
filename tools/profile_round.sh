#!/bin/bash
# Evidence for one round (run on the GPU box): tools/profile_round.sh TAG
#   1. rocprofv3 --kernel-trace --stats over the default bench.py command -> per-dispatch (grid-keyed) summary + stats CSV
#   2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + GRBM) over a short eager bench -> per-(kernel, grid)
#      HBM traffic per launch (FETCH_SIZE x2 per MI355X_MICROARCH.md section HBM, KB -> bytes)
# Everything lands in gpurun_out/; copy the summaries into profiles/ afterwards.
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline \
  > $R/gpurun_out/prof_${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_${TAG}.err
python3 $R/tools/kernel_summary.py $R/gpurun_out/prof_$TAG $R/gpurun_out/prof_${TAG}_kernel_summary.csv 63
cp $(ls $R/gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1) $R/gpurun_out/prof_${TAG}_kernel_stats.csv 2>/dev/null
# same command with every launch on one stream: per-kernel durations without contention from overlapping chains (what
# bench.py's roofline leg measures live)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_serial -- python3 $R/bench.py --no-cpu-baseline \
  --serial-streams > $R/gpurun_out/prof_${TAG}_serial_bench_under_rocprof.json 2> $R/gpurun_out/prof_${TAG}_serial.err
python3 $R/tools/kernel_summary.py $R/gpurun_out/prof_${TAG}_serial $R/gpurun_out/prof_${TAG}_serial_kernel_summary.csv 63
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --steps 2 --warmup 1 \
    --no-graph --no-cpu-baseline --no-roofline --serial-streams > /dev/null 2> $R/gpurun_out/pmc_${TAG}_$i.err
done
python3 - <<PY
import csv, glob, json, collections, os
R = "$R"; TAG = "$TAG"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sorted(glob.glob(R + "/gpurun_out/pmc_%s_*/" % TAG)):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]
            key = name + " | grid " + r["Grid_Size"] + " | lds " + r.get("LDS_Block_Size", "")
            a = acc[key][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for k, cs in acc.items():
    e = {"launches": max(v[1] for v in cs.values())}
    for c, (v, n) in cs.items():
        e[c + "_avg"] = v / n
    if "FETCH_SIZE" in cs: e["hbm_read_bytes"] = cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1] * 1024 * 2      # KB, x2 (gfx950)
    if "WRITE_SIZE" in cs: e["hbm_write_bytes"] = cs["WRITE_SIZE"][0] / cs["WRITE_SIZE"][1] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
        e["mfma_util"] = (cs["SQ_VALU_MFMA_BUSY_CYCLES"][0] / cs["SQ_VALU_MFMA_BUSY_CYCLES"][1]) / (cs["GRBM_GUI_ACTIVE"][0] / cs["GRBM_GUI_ACTIVE"][1] / 8 * 1024)
    out[k] = e
# provenance: the library version the counters were taken on (bench.py reports traffic only while it matches the loaded
# library); the commit is added when the file is copied into profiles/ (the GPU box has no .git)
import ctypes
out["_meta"] = {"s2p_version": ctypes.CDLL(R + "/s2p_amd/csrc/libs2p_hip.so").s2p_version(), "commit": None, "tag": TAG}
json.dump(out, open(R + "/gpurun_out/pmc_%s_traffic.json" % TAG, "w"), indent=1, sort_keys=True)
print("pmc: %d (kernel, grid) groups -> gpurun_out/pmc_%s_traffic.json" % (len(out), TAG))
PY
