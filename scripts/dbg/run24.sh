mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "chain" > gpurun_out/r3/t24.log 2>&1; echo exit $? >> gpurun_out/r3/t24.log; tail -5 gpurun_out/r3/t24.log
python scripts/dbg/chain_bwd_diff.py 256 197 192 768 0.1 2>&1 | tail -6
python scripts/dbg/chain_bwd_diff.py 768 65 128 1024 0.2 2>&1 | tail -2
for pre in 1 2; do IQ_TUNE_CHAIN_PRE=$pre python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r3/b24_pre$pre.json 2>gpurun_out/r3/b24_pre$pre.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b24_pre$pre.json").read().strip().splitlines()[-1])
print("pre$pre", j["value"], j["ms_per_step"])
PY
done
python -m pytest tests/test_gpu_model.py tests/test_gpu_trainer.py -m gpu -x -q > gpurun_out/r3/t24m.log 2>&1; echo exit $? >> gpurun_out/r3/t24m.log; tail -5 gpurun_out/r3/t24m.log
