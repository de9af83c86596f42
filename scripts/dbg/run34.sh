mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t34.log 2>&1; rc=$?; echo exit $rc >> gpurun_out/r3/t34.log; tail -4 gpurun_out/r3/t34.log
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r3/b34_default.json 2> gpurun_out/r3/b34_default.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b34_default.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["secondary"]["value"], j["roofline"]["frac"], j["cpu_baseline"]["value"], j["cpu_baseline"]["batch"])
print([(k["kernel"], k["avg_us"]) for k in j["roofline"]["kernels"]])
PY
