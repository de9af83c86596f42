#!/bin/bash
# rocprofv3 kernel trace + stats of the bench workload (eager launches so every kernel is a named dispatch).
# usage: scripts/prof_step.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/ and gpurun_out/prof_<tag>_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --graph 0 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary "$@" > gpurun_out/prof_$tag.log 2>&1 || echo "rocprofv3 failed"
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/prof_${tag}_stats.csv; cut -c1-150 "$f" | head -24; fi
