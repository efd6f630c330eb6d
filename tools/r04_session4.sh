#!/bin/bash
# round 4, GPU session 4: store-flavour variants with the store-data hazard closed (correctness first, then time),
# row-sum splits with >= 128-byte parts and a residency cap, suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for v in cur st1 st2 st3; do
  echo "== $v"; TSPGNN_LIB=$R/tools/variants/$v.so timeout 600 python -m pytest tests/test_gpu_anchors.py tests/test_gpu_model.py -m gpu -q -x -k "anchor or forward_parity or captured_graph" 2>&1 | tail -2
done > $O/s4_variants_correct.txt 2>&1; cat $O/s4_variants_correct.txt
W=c2 ROUNDS=5 timeout 1200 tools/abn.sh cur st1 st2 st3 > $O/s4_ab_c2.txt 2>&1; cat $O/s4_ab_c2.txt
W=c4 ROUNDS=3 timeout 900 tools/abn.sh cur st1 st3 > $O/s4_ab_c4.txt 2>&1; cat $O/s4_ab_c4.txt
one() {  # workload split lds_kb steps
  TSPGNN_ROWSUM_SPLIT=$2 TSPGNN_ROWSUM_LDS_KB=$3 python bench.py --workload $1 --steps ${4:-10} --warmup 3 --no-cpu-baseline --train-steps 0 --serve-batches 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); k = j['kernels_us']
rs = [round(v['avg_us'], 1) for n, v in k.items() if 'rowsum' in n]
print('$1 split=$2 lds=$3', j['ms_per_step'], 'rowsum', rs)"
}
for i in 1 2; do one c4 0 0; for l in 0 16 32 48; do one c4 2 $l; done; done > $O/s4_split_c4.txt 2>&1; cat $O/s4_split_c4.txt
for i in 1 2; do one c5 0 0 6; for l in 0 16 32 48 64; do one c5 2 $l 6; done; done > $O/s4_split_c5.txt 2>&1; cat $O/s4_split_c5.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/s4_pytest.log 2>&1; echo "pytest rc $?" >> $O/s4_pytest.log; tail -4 $O/s4_pytest.log
