mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "chain" > gpurun_out/r3/t23.log 2>&1; echo exit $? >> gpurun_out/r3/t23.log; tail -5 gpurun_out/r3/t23.log
python scripts/layer_kernels.py 192 3 768 197 256 0.1 > gpurun_out/r3/lk23_B.log 2>&1 && grep -E "chain|out-proj|ffn|qkv|q,k,v" gpurun_out/r3/lk23_B.log
python scripts/layer_kernels.py 128 8 1024 65 768 0.2 > gpurun_out/r3/lk23_C768.log 2>&1 && grep -E "chain|out-proj|ffn|qkv|q,k,v" gpurun_out/r3/lk23_C768.log
for pre in 0 1 2; do IQ_TUNE_CHAIN_PRE=$pre python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r3/b23_pre$pre.json 2>gpurun_out/r3/b23_pre$pre.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b23_pre$pre.json").read().strip().splitlines()[-1])
print("pre$pre", j["value"], j["ms_per_step"])
PY
done
python -m pytest tests/test_gpu_model.py tests/test_gpu_trainer.py -m gpu -x -q > gpurun_out/r3/t23m.log 2>&1; echo exit $? >> gpurun_out/r3/t23m.log; tail -5 gpurun_out/r3/t23m.log
