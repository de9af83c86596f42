mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -v -k "qkv_dgrad_ffn_chain" > gpurun_out/r3/t26a.log 2>&1; rc=$?; echo exit $rc >> gpurun_out/r3/t26a.log; grep -E "PASSED|FAILED|exit|Abort" gpurun_out/r3/t26a.log | tail -12
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "chain" > gpurun_out/r3/t26.log 2>&1; rc=$?; echo exit $rc >> gpurun_out/r3/t26.log; tail -3 gpurun_out/r3/t26.log
[ $rc -eq 0 ] || exit 1
for pre in 2 3; do IQ_TUNE_CHAIN_PRE=$pre python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r3/b26_pre$pre.json 2>gpurun_out/r3/b26_pre$pre.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b26_pre$pre.json").read().strip().splitlines()[-1])
print("pre$pre", j["value"], j["ms_per_step"])
for k in j["roofline"]["kernels"]: print("   ", k["kernel"], k["launches_per_step"], k["avg_us"])
PY
done
python -m pytest tests/test_gpu_model.py tests/test_gpu_trainer.py -m gpu -x -q > gpurun_out/r3/t26m.log 2>&1; echo exit $? >> gpurun_out/r3/t26m.log; tail -5 gpurun_out/r3/t26m.log
