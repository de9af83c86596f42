#!/bin/bash
# Round-3 measurement pass on the GPU box: default bench line, the other named configurations, rocprofv3 kernel stats
# (cfg B, C and D, eager launches) and the PMC passes (cfg B and C).  Everything lands under gpurun_out/; the summaries to keep
# are copied into profiles/ afterwards (bench.py reads profiles/r03_pmc_family_cfg<id>.json for roofline.traffic).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
scripts/pmc_family.sh B gpurun_out/r03_pmc_family_cfgB.json > gpurun_out/pmcB.log 2>&1
scripts/pmc_family.sh C gpurun_out/r03_pmc_family_cfgC.json > gpurun_out/pmcC.log 2>&1
mkdir -p profiles && cp gpurun_out/r03_pmc_family_cfgB.json gpurun_out/r03_pmc_family_cfgC.json profiles/ 2>/dev/null
scripts/prof_step.sh r03B > /dev/null 2>&1
scripts/prof_step.sh r03C --config C > /dev/null 2>&1
scripts/prof_step.sh r03D --config D > /dev/null 2>&1
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || echo "default bench failed"
for c in A C Cp D ref; do
  python bench.py --config $c --steps 20 --warmup 5 --cpu-budget 8 --no-accuracy > gpurun_out/r03_bench_cfg$c.json 2> gpurun_out/r03_bench_cfg$c.err || echo "bench $c failed"
done
ls -la gpurun_out/r03_* gpurun_out/prof_r03*_stats.csv
