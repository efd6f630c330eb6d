cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_split_kernels.py -x -q 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_anchors.py -x -q -k "not bf16" 2>&1 | tail -3
for w in c2 c4; do
timeout 120 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --train-steps 0 > gpurun_out/b.json 2> gpurun_out/b.err || tail -3 gpurun_out/b.err; python - <<PY
import json
r=json.load(open("gpurun_out/b.json")); print("$w ms", r["ms_per_step"], {k:v["avg_us"] for k,v in r["kernels_us"].items() if v["n"]>=8})
PY
done
