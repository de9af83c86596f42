mkdir -p gpurun_out/r3
for pre in 2 3; do IQ_TUNE_CHAIN_PRE=$pre python bench.py --config C --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r3/b30_C_pre$pre.json 2>gpurun_out/r3/b30.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b30_C_pre$pre.json").read().strip().splitlines()[-1])
print("C pre$pre", j["value"], j["ms_per_step"])
PY
done
for pre in 2 3; do IQ_TUNE_CHAIN_PRE=$pre python bench.py --config ref --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r3/b30_ref_pre$pre.json 2>gpurun_out/r3/b30.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b30_ref_pre$pre.json").read().strip().splitlines()[-1])
print("ref pre$pre", j["value"], j["ms_per_step"])
PY
done
