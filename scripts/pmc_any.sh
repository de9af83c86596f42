#!/bin/bash
# PMC passes for one script: scripts/pmc_any.sh <tag> <kernel-substring> <script> [args...]
tag=$1; shift; kname=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}/p$i -- python3 "$@" > gpurun_out/pmc_${tag}_p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_${tag}/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "${kname}" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v = v[2:] if len(v) > 2 else v
    print(f"{k:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
