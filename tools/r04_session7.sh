#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/s7_pytest.log 2>&1; echo "pytest rc $?" >> $O/s7_pytest.log; tail -4 $O/s7_pytest.log
python bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline --train-steps 0 --serve-batches 0 > $O/s7_c5_bench.json 2> $O/s7_c5_bench.err; python -c "
import json; j=json.loads(open('$O/s7_c5_bench.json').read().strip().splitlines()[-1]); print('c5', j['ms_per_step'], {k:v for k,v in j['kernels_us'].items()})"
cd /tmp; export TMPDIR=/tmp
rm -rf $O/kt_c5b; rocprofv3 --kernel-trace --stats -d $O/kt_c5b -o k -- python $R/tools/forward_graph.py c5 20 > $O/kt_c5b.log 2>&1
f=$(find $O/kt_c5b -name "*.db" | head -1)
python $R/profiles/summarize_rocpd.py $f "round 4: python tools/forward_graph.py c5 20 (HIP-graph replays of the forward pass, nothing else)" > $O/c5_forward_kernel_stats.txt
python $R/tools/timeline_rocpd.py $f > $O/c5_forward_timeline.txt 2>&1
rm -rf $O/kt_c5b; head -22 $O/c5_forward_kernel_stats.txt | cut -c1-160
