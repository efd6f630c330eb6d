#!/bin/bash
# GPU session for the recomputing message-MLP backward: kernel + model tests, kernel timing, C2 training step A/B
# (TSPGNN_RECOMPUTE=1 = recompute + weight gradients in the launch).  Outputs in gpurun_out/rc/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rc; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_backward_kernels.py -q -k "taped_h2 or recompute" > $O/t_kernel.log 2>&1; echo "kernel rc=$?" | tee $O/status.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "recomputed or pushed_training or gradient_parity or fused_messages or chunks" > $O/t_model.log 2>&1; echo "model rc=$?" | tee -a $O/status.txt
{ RC_DW=1 timeout 120 python tools/rc_bench.py 50; RC_DW=0 timeout 120 python tools/rc_bench.py 50; RC_DW=1 timeout 120 python tools/rc_bench.py 50; } 2>&1 | grep -v amdgpu.ids | tee $O/rc_bench.txt
for rep in 1 2 3; do
  for rc in 0 1; do
    TSPGNN_RECOMPUTE=$rc timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/train_rc${rc}_$rep.json 2> $O/train_rc${rc}_$rep.err
    python - <<PY
import json
try:
    j = json.loads(open("$O/train_rc${rc}_$rep.json").read().strip().splitlines()[-1])
    print("rc=$rc rep=$rep ms_per_step", j.get("ms_per_step"), "loss", j.get("loss"))
except Exception as e:
    print("rc=$rc rep=$rep FAILED", e)
PY
  done
done | tee -a $O/status.txt
tail -5 $O/t_kernel.log; tail -8 $O/t_model.log
