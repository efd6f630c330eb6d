#!/bin/bash
# A/B on one box: alternating bench runs of the in-tree library and tools/variants/$1.so.
#   tools/ab.sh VARIANT [workload] [rounds]  -> lines "A|B <ms_per_step> <cell launch us>"
V=$1; W=${2:-c2}; N=${3:-3}
R=$GRAFT_REPO_ROOT
for i in $(seq $N); do
  for side in A B; do
    if [ $side = A ]; then unset TSPGNN_LIB; else export TSPGNN_LIB=$R/tools/variants/$V.so; fi
    python $R/bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --train-steps ${TRAIN:-0} 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1])
k = j.get('kernels_us', {})
t = j.get('train')
print('$side', j['ms_per_step'], json.dumps(k)[:300], t.get('ms_per_step') if isinstance(t, dict) else '')
"
  done
done
