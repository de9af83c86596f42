#!/bin/bash
# HBM traffic of the gemm_nt kernel family over whole training steps (cfg B, eager launches): two separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE; guide: no trace domains combined with --pmc), summed per dispatch and averaged.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_family/$ctr -- python3 bench.py --graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_family_$ctr.log 2>&1 || echo "pass $ctr failed"
done
python3 - <<'PY'
import csv, glob, json, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_family/{ctr}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr: continue
            k = r["Kernel_Name"]
            fam = "gemm_nt" if "gemm_nt" in k else "wgrad" if "wgrad_pw" in k or "wgrad_kernel" in k else "attn_bwd" if "attn_bwd" in k else "attn_fwd" if "attn_fwd" in k else "ln_bwd" if "ln_bwd_kernel" in k else "ln_fwd" if "ln_fwd" in k else None
            if fam: tot[fam][ctr] += float(r["Counter_Value"]); cnt[fam][ctr] += 1
out = {}
for fam in tot:
    n = cnt[fam]["FETCH_SIZE"]
    fetch_kb = tot[fam]["FETCH_SIZE"] / max(n, 1) * 2.0      # gfx950: FETCH_SIZE under-counts by 2 (MI355X_MICROARCH guide)
    write_kb = tot[fam]["WRITE_SIZE"] / max(cnt[fam]["WRITE_SIZE"], 1)
    out[fam] = {"dispatches": n, "fetch_bytes_per_launch": fetch_kb * 1024, "write_bytes_per_launch": write_kb * 1024,
                "hbm_bytes_per_launch": (fetch_kb + write_kb) * 1024}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_family.json", "w"), indent=1)
PY
