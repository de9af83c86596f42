#!/bin/bash
# PMC passes over whole training steps (eager launches), one rocprofv3 run per counter set (guide: FETCH_SIZE and WRITE_SIZE
# do not fit one pass; no trace domain other than --kernel-trace next to --pmc).
#   scripts/pmc_family.sh <config B|C|...> <out.json>
# Per kernel family: fabric-side bytes per launch -- what the L2s fetched from / wrote to the memory fabric, i.e. HBM or the
# Infinity Cache in front of it (FETCH_SIZE x2: gfx950 counts 64 B per 128 B request, MI355X_MICROARCH.md "HBM"; WRITE_SIZE as is).  Whole step: MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
cfg=${1:-B}; out=${2:-gpurun_out/pmc_family_cfg$cfg.json}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_family_$cfg/p$i -- python3 bench.py --config $cfg --graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/pmc_family_${cfg}_p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$cfg" "$out" <<'PY'
import csv, glob, json, collections, sys
cfg, out_path = sys.argv[1], sys.argv[2]
def fam_of(k):
    if "gemm_nt" in k or "gemm_ln" in k or "ffn_chain" in k or "gemm_big" in k: return "gemm_nt"
    if "wgrad" in k: return "wgrad"
    if "attn_bwd" in k or "attn_frame_bwd" in k: return "attn_bwd"
    if "attn_fwd" in k or "attn_frame_fwd" in k: return "attn_fwd"
    if "ln_bwd" in k: return "ln_bwd"
    if "ln_fwd" in k: return "ln_fwd"
    return "misc"
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"gpurun_out/pmc_family_{cfg}/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "at::native" in k or "rocclr" in k: continue
        fam = fam_of(k); c = r["Counter_Name"]
        tot[fam][c] += float(r["Counter_Value"])
        if "wgrad_reduce" not in k: cnt[fam][c] += 1     # a weight-gradient call = its GEMM launch + its slab reduce
out = {}
allc = collections.defaultdict(float)
for fam in tot:
    for c, v in tot[fam].items(): allc[c] += v
    if fam == "misc": continue
    n = max(cnt[fam]["FETCH_SIZE"], 1)
    fetch_kb = tot[fam]["FETCH_SIZE"] / n * 2.0      # gfx950: FETCH_SIZE under-counts wide streaming reads by 2
    write_kb = tot[fam]["WRITE_SIZE"] / max(cnt[fam]["WRITE_SIZE"], 1)
    e = {"dispatches": cnt[fam]["FETCH_SIZE"], "fetch_bytes_per_launch": fetch_kb * 1024, "write_bytes_per_launch": write_kb * 1024,
         "hbm_bytes_per_launch": (fetch_kb + write_kb) * 1024}
    if tot[fam].get("GRBM_GUI_ACTIVE"):
        e["mfma_util"] = round(tot[fam]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * tot[fam]["GRBM_GUI_ACTIVE"] / 8.0), 4)
    out[fam] = e
if allc.get("GRBM_GUI_ACTIVE"):
    out["mfma_util"] = round(allc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * allc["GRBM_GUI_ACTIVE"] / 8.0), 4)
    out["mfma_util_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), all native kernels of 4 training steps"
    out["sq_insts_mfma_per_step"] = allc["SQ_INSTS_MFMA"] / 4.0
print(json.dumps(out, indent=1))
json.dump(out, open(out_path, "w"), indent=1)
PY
