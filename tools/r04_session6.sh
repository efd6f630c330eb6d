#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "train_steps_follow or gradient_parity" 2>&1 | grep -a "train steps\|worst per-variable\|passed\|failed" > $O/s6_bars.txt; cat $O/s6_bars.txt
for lt in 12 11 10; do TSPGNN_H2_LOCK_TILES=$lt python bench.py --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 --serve-batches 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('lock_tiles $lt', j['ms_per_step'], j['kernels_us']['tspgnn_lnlstm_mlp_fwd_multi_h2'])"; done > $O/s6_lock.txt 2>&1; cat $O/s6_lock.txt
for lt in 12 11 10; do TSPGNN_H2_LOCK_TILES=$lt python bench.py --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 --serve-batches 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('lock_tiles $lt', j['ms_per_step'], j['kernels_us']['tspgnn_lnlstm_mlp_fwd_multi_h2'])"; done >> $O/s6_lock.txt 2>&1; tail -3 $O/s6_lock.txt
