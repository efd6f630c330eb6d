#!/bin/bash
# round 4, GPU session 5: the adopted store variant end to end (suite, C2 / C4 / C1 against round 3's library), training step
# with write-through stores in the forward tape / the backward's dz
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/s5_pytest.log 2>&1; echo "pytest rc $?" >> $O/s5_pytest.log; tail -3 $O/s5_pytest.log
W=c2 ROUNDS=4 timeout 900 tools/abn.sh r03base cur > $O/s5_ab_c2.txt 2>&1; cat $O/s5_ab_c2.txt
W=c4 ROUNDS=2 timeout 900 tools/abn.sh r03base cur > $O/s5_ab_c4.txt 2>&1; cat $O/s5_ab_c4.txt
W=c1 ROUNDS=2 timeout 900 tools/abn.sh r03base cur > $O/s5_ab_c1.txt 2>&1; cat $O/s5_ab_c1.txt
tr() {  # lib wt -> training ms
  TSPGNN_LIB=$R/tools/variants/$1.so TSPGNN_H2_WT=$2 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('train $1 wt=$2', j['ms_per_step'])"
}
for i in 1 2 3; do tr r03base -1; tr cur 0; tr cur 1; tr bwt 0; tr bwt 1; done > $O/s5_train.txt 2>&1; cat $O/s5_train.txt
