mkdir -p gpurun_out/r3
for b in 64 128; do for ch in 0 1; do IQ_TUNE_FFN_CHAIN=$ch python bench.py --batch $b --no-cpu-baseline --no-secondary --steps 50 --warmup 10 > gpurun_out/r3/b29_B${b}_ch$ch.json 2>gpurun_out/r3/b29.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b29_B${b}_ch$ch.json").read().strip().splitlines()[-1])
print("B batch $b chain$ch", j["value"], j["ms_per_step"])
PY
done; done
for c in ref Cp; do python bench.py --config $c --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r3/b29_$c.json 2>gpurun_out/r3/b29.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b29_$c.json").read().strip().splitlines()[-1])
print("$c", j["value"], j["ms_per_step"])
PY
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t29.log 2>&1; echo exit $? >> gpurun_out/r3/t29.log; tail -4 gpurun_out/r3/t29.log
