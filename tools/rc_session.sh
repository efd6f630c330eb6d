#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5t; mkdir -p $O; cd $R
timeout 300 python tools/wgrad_bench.py c5 2>&1 | grep -v amdgpu | tee $O/wgrad_c5.txt
timeout 1200 python -m pytest tests -m gpu -q -x -k "bf16 or wgrad" > $O/t_bf16.log 2>&1; echo "bf16 tests rc=$?"
tail -4 $O/t_bf16.log
for rep in 1 2; do
    timeout 900 python bench.py --workload c5 --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/c5train_$rep.json 2> $O/c5train_$rep.err
    python - <<PY
import json
j = json.loads(open("$O/c5train_$rep.json").read().strip().splitlines()[-1])
print("c5 train rep=$rep ms_per_step", j.get("ms_per_step"), "loss", j.get("loss"))
PY
done
