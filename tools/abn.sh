#!/bin/bash
# Multi-way A/B on one box: ROUNDS alternating bench runs of each variant in tools/variants (names as arguments).
#   W=c2 ROUNDS=2 tools/abn.sh o_base o_nts ...   -> "name ms_per_step cell_us rowsum_us"
R=$GRAFT_REPO_ROOT
for i in $(seq ${ROUNDS:-2}); do
  for V in "$@"; do
    TSPGNN_LIB=$R/tools/variants/$V.so python $R/bench.py --workload ${W:-c2} --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); k = j.get('kernels_us', {})
print('%-10s %.4f  cell %.2f  rowsum %.2f' % ('$V', j['ms_per_step'], k.get('tspgnn_lnlstm_mlp_fwd_multi_h2', {}).get('avg_us', 0), k.get('tspgnn_csr_rowsum_f32', {}).get('avg_us', 0)))
"
  done
done
