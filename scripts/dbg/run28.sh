mkdir -p gpurun_out/r3
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "chain" > gpurun_out/r3/t28.log 2>&1; rc=$?; echo exit $rc >> gpurun_out/r3/t28.log; tail -4 gpurun_out/r3/t28.log
[ $rc -eq 0 ] || exit 1
python scripts/layer_kernels.py 128 8 1024 65 256 0.2 > gpurun_out/r3/lk28_C.log 2>&1 && cat gpurun_out/r3/lk28_C.log | grep -v amdgpu.ids
python scripts/dbg/chain_bwd_diff.py 256 65 128 1024 0.2 2>&1 | tail -1
for ch in 0 1; do IQ_TUNE_FFN_CHAIN=$ch python bench.py --config C --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r3/b28_C_ch$ch.json 2>gpurun_out/r3/b28_C_ch$ch.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b28_C_ch$ch.json").read().strip().splitlines()[-1])
print("C chain$ch", j["value"], j["ms_per_step"], j["roofline"]["frac"])
for k in j["roofline"]["kernels"]: print("   ", k["kernel"], k["launches_per_step"], k["avg_us"])
PY
done
for ch in 0 1; do IQ_TUNE_FFN_CHAIN=$ch python bench.py --config A --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r3/b28_A_ch$ch.json 2>gpurun_out/r3/b28_A_ch$ch.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b28_A_ch$ch.json").read().strip().splitlines()[-1])
print("A chain$ch", j["value"], j["ms_per_step"])
PY
done
