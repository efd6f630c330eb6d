cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backward_kernels.py -x -q -k "h2 or einit" 2>&1 | tail -15
python -m pytest tests/test_gpu_kernels.py -x -q -k "einit" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -x -q -k "gradient or train or captured" 2>&1 | tail -8
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --train-steps 3 > gpurun_out/b.json 2> gpurun_out/b.err; python - <<PY
import json
r=json.load(open("gpurun_out/b.json")); print("fwd ms", r["ms_per_step"], "train", r["train"], {k:v["avg_us"] for k,v in r["kernels_us"].items()})
PY
