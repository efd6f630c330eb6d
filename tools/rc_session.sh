#!/bin/bash
# GPU session: f16x2 taped MLP backward at d=128 / bf16 tapes: kernel + bf16 model tests, C5 shard training step A/B.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rc; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_backward_kernels.py -q -k "taped_h2" > $O/t_kernel.log 2>&1; echo "kernel rc=$?" | tee $O/status.txt
timeout 1200 python -m pytest tests -m gpu -q -x -k "bf16" > $O/t_model.log 2>&1; echo "bf16 tests rc=$?" | tee -a $O/status.txt
for rep in 1 2; do
  for v in 0 1; do
    TSPGNN_MLP_BWD_H2=$v timeout 900 python bench.py --workload c5 --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/c5train_h2${v}_$rep.json 2> $O/c5train_h2${v}_$rep.err
    python - <<PY
import json
try:
    j = json.loads(open("$O/c5train_h2${v}_$rep.json").read().strip().splitlines()[-1])
    print("c5 mlp_bwd_h2=$v rep=$rep ms_per_step", j.get("ms_per_step"), "loss", j.get("loss"))
except Exception as e:
    print("h2=$v rep=$rep FAILED", e)
PY
  done
done | tee -a $O/status.txt
tail -5 $O/t_kernel.log; tail -8 $O/t_model.log
