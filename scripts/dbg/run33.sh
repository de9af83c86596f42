mkdir -p gpurun_out/r3
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attn or attention" > gpurun_out/r3/t33.log 2>&1; rc=$?; echo exit $rc >> gpurun_out/r3/t33.log; tail -3 gpurun_out/r3/t33.log
[ $rc -eq 0 ] || exit 1
for sp in 0 1 0 1; do IQ_TUNE_ATTN_SPLIT=$sp python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 8 > gpurun_out/r3/b33_s$sp.json 2>gpurun_out/r3/b33.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b33_s$sp.json").read().strip().splitlines()[-1])
k=[x for x in j["roofline"]["kernels"] if "attn_bwd" in x["kernel"]]
print("split$sp B", j["value"], j["ms_per_step"], k[0]["kernel"] if k else "", k[0]["avg_us"] if k else "")
PY
done
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r3/t33m.log 2>&1; echo exit $? >> gpurun_out/r3/t33m.log; tail -3 gpurun_out/r3/t33m.log
