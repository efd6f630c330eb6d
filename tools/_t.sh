cd $GRAFT_REPO_ROOT
for w in c2 c4 c5; do
python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline --train-steps 0 > gpurun_out/b.json 2> gpurun_out/b.err || tail -3 gpurun_out/b.err; python - <<PY
import json
r=json.load(open("gpurun_out/b.json")); print("$w ms", r["ms_per_step"], {k:v["avg_us"] for k,v in r["kernels_us"].items() if v["n"]>=8})
PY
done
python -m pytest tests/test_gpu_split_kernels.py tests/test_gpu_bf16_kernels.py -x -q 2>&1 | tail -3
