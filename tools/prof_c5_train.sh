#!/bin/bash
# C5-shard training step under rocprofv3 --kernel-trace --stats (GPU box): TAG=r03 tools/prof_c5_train.sh
R=$GRAFT_REPO_ROOT; TAG=${TAG:-r03}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --workload c5 --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $O/c5_train_bench.json 2> $O/c5_train_bench.err
rm -rf $O/kt5
rocprofv3 --kernel-trace --stats -d $O/kt5 -o k -- python $R/bench.py --workload c5 --mode train --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $O/kt5.log 2>&1
f=$(find $O/kt5 -name "*.db" | head -1)
python $R/profiles/summarize_rocpd.py $f "round 3 ($TAG): python bench.py --workload c5 --mode train --steps 1 --warmup 1 --no-cpu-baseline --no-graph (C5 shard training step, bf16 storage)" > $O/c5_train_kernel_stats.txt
rm -rf $O/kt5
python -c "import json; j=json.load(open('$O/c5_train_bench.json')); print('c5 train ms_per_step', j['ms_per_step'])"
head -24 $O/c5_train_kernel_stats.txt | cut -c1-150
