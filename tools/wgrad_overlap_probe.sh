#!/bin/bash
# C2 training step (HIP-graph replay) with the weight gradients of every SUB finished steps on a second stream beside the
# backward chain (TSPGNN_WGRAD_SUB; 0 = one reduction over all steps after the chain)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for s in ${SUBS:-0 2 4 8 16}; do
  TSPGNN_WGRAD_SUB=$s python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/wg_$s.err | tail -1 | \
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('TSPGNN_WGRAD_SUB=$s: %.4f ms per C2 training step' % j['ms_per_step'])" || tail -5 gpurun_out/wg_$s.err
done; done
