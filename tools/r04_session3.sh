#!/bin/bash
# round 4, GPU session 3: suite with the new kernels, row-sum column split A/B at C4 / C5, stability of the adopted stores
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/s3_pytest.log 2>&1; echo "pytest rc $?" >> $O/s3_pytest.log; tail -4 $O/s3_pytest.log
one() {  # workload split -> "ms_per_step rowsum_us cell_us"
  TSPGNN_ROWSUM_SPLIT=$2 python bench.py --workload $1 --steps ${3:-10} --warmup 3 --no-cpu-baseline --train-steps 0 --serve-batches 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); k = j['kernels_us']
rs = [v['avg_us'] for n, v in k.items() if 'rowsum' in n]
print('$1 split=$2', j['ms_per_step'], 'rowsum', rs, 'marginal', j['roofline'].get('marginal_cost_us'))"
}
for i in 1 2; do for s in 0 4 8 -1; do one c4 $s; done; done > $O/s3_split_c4.txt 2>&1; cat $O/s3_split_c4.txt
for i in 1 2; do for s in 0 4 8 -1; do one c5 $s 6; done; done > $O/s3_split_c5.txt 2>&1; cat $O/s3_split_c5.txt
for i in 1 2 3 4 5 6; do one c2 -1 30; done > $O/s3_c2_repeat.txt 2>&1; cat $O/s3_c2_repeat.txt
