#!/bin/bash
# Round-2 measurement pass on the GPU box: default bench line, the other named configurations, rocprofv3 kernel stats
# (cfg B and C, eager launches) and the PMC passes.  Everything lands under gpurun_out/; the summaries to keep are
# copied into profiles/ afterwards.
cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err || echo "default bench failed"
for c in A C Cp D ref; do
  python bench.py --config $c --steps 20 --warmup 5 --cpu-budget 8 > gpurun_out/r02_bench_cfg$c.json 2> gpurun_out/r02_bench_cfg$c.err || echo "bench $c failed"
done
scripts/prof_step.sh r02B > /dev/null 2>&1
scripts/prof_step.sh r02C --config C > /dev/null 2>&1
scripts/pmc_family.sh B gpurun_out/r02_pmc_family_cfgB.json > gpurun_out/pmcB.log 2>&1
scripts/pmc_family.sh C gpurun_out/r02_pmc_family_cfgC.json > gpurun_out/pmcC.log 2>&1
ls -la gpurun_out/r02_* gpurun_out/prof_r02*_stats.csv
