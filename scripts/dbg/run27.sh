mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r3/t27.log 2>&1; rc=$?; echo exit $rc >> gpurun_out/r3/t27.log; tail -14 gpurun_out/r3/t27.log
[ $rc -eq 0 ] || exit 1
bash scripts/r03_collect.sh > gpurun_out/r3/collect.log 2>&1; tail -30 gpurun_out/r3/collect.log
