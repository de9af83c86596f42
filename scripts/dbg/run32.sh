mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_trainer.py tests/test_gpu_ddp.py -m gpu -x -q > gpurun_out/r3/t32.log 2>&1; rc=$?; echo exit $rc >> gpurun_out/r3/t32.log; tail -4 gpurun_out/r3/t32.log
[ $rc -eq 0 ] || exit 1
for f in 0 1 0 1; do IQ_TUNE_REDUCE_FORK=$f python bench.py --no-cpu-baseline --steps 40 --warmup 8 > gpurun_out/r3/b32_f$f.json 2>gpurun_out/r3/b32.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b32_f$f.json").read().strip().splitlines()[-1])
print("fork$f B", j["value"], j["ms_per_step"], " C", j["secondary"]["value"], j["secondary"]["ms_per_step"])
PY
done
for f in 0 1; do IQ_TUNE_REDUCE_FORK=$f python bench.py --config D --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r3/b32_D_f$f.json 2>gpurun_out/r3/b32.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r3/b32_D_f$f.json").read().strip().splitlines()[-1])
print("fork$f D", j["value"], j["ms_per_step"])
PY
done
