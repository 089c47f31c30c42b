#!/bin/bash
# PMC counters of the slab wgrad kernel (S=1 and S=4), one pass per counter group (rocprofv3, no tracing domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export S2P_LIB=$R/s2p_amd/csrc/libs2p_hip_diag.so
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM SQ_WAVES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_slab_$tag -- python3 $R/tools/bench_wgrad_slab.py resblk 1,4 > $R/gpurun_out/pmc_slab_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for d in sorted(glob.glob(R + "/gpurun_out/pmc_slab_*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            if "wgrad_slab_kernel" not in row["Kernel_Name"]: continue
            acc[row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for g, cs in acc.items():
            print(os.path.basename(os.path.dirname(d)), "grid", g, {k: round(sum(v) / len(v)) for k, v in cs.items()})
PY
