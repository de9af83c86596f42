mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t39.log 2>&1; rc=$?; echo exit $rc >> gpurun_out/r3/t39.log; tail -4 gpurun_out/r3/t39.log
[ $rc -eq 0 ] || exit 1
python bench.py --no-cpu-baseline --steps 40 --warmup 8 > gpurun_out/r3/b39.json 2> gpurun_out/r3/b39.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b39.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["secondary"]["value"], j["secondary"]["ms_per_step"])
print([(k["kernel"], k["avg_us"]) for k in j["roofline"]["kernels"]])
PY
