#!/bin/bash
# Round-4 profile set (run on the GPU box through gpurun; outputs in gpurun_out/r04/, summaries copied to profiles/ by hand):
#   bench lines (c2, c4, c5; c2 / c5 training), rocprofv3 kernel stats + forward timeline, PMC traffic of the SpMM kernels
#   (micro-loop and in the forward), SQ / LDS / memory-path counters of the cell launch.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in ${WORKLOADS:-c2 c4 c5}; do
  steps=20; [ $w != c2 ] && steps=6
  python $R/bench.py --workload $w --steps $steps --warmup 3 --train-steps $([ $w = c5 ] && echo 0 || echo 2) > $O/${w}_bench.json 2> $O/${w}_bench.err
  rm -rf $O/kt_$w
  rocprofv3 --kernel-trace --stats -d $O/kt_$w -o k -- python $R/tools/forward_graph.py $w 20 > $O/kt_$w.log 2>&1
  f=$(find $O/kt_$w -name "*.db" | head -1)
  python $R/profiles/summarize_rocpd.py $f "round 4: python tools/forward_graph.py $w 20 (HIP-graph replays of the forward pass, nothing else)" > $O/${w}_forward_kernel_stats.txt
  python $R/tools/timeline_rocpd.py $f > $O/${w}_forward_timeline.txt 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${w}_$c
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${w}_$c -o p -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 --no-graph > $O/pmc_${w}_$c.log 2>&1
    f=$(find $O/pmc_${w}_$c -name "*.db" | head -1)
    python $R/profiles/summarize_pmc.py $f > $O/${w}_pmc_$c.txt
    rm -rf $O/pmcf_${w}_$c
    rocprofv3 --kernel-trace --pmc $c -d $O/pmcf_${w}_$c -o p -- python $R/tools/forward_only.py $w 3 > $O/pmcf_${w}_$c.log 2>&1
    f=$(find $O/pmcf_${w}_$c -name "*.db" | head -1)
    python $R/profiles/summarize_pmc.py $f > $O/${w}_forward_pmc_$c.txt
  done
  find $O -name "*.db" -delete
done
# training steps
python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/c2_train_bench.json 2> $O/c2_train_bench.err
rm -rf $O/kt_c2t
rocprofv3 --kernel-trace --stats -d $O/kt_c2t -o k -- python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_c2t.log 2>&1
f=$(find $O/kt_c2t -name "*.db" | head -1)
python $R/profiles/summarize_rocpd.py $f "round 4: python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline (C2 training step, HIP-graph replay)" > $O/c2_train_kernel_stats.txt
python $R/bench.py --workload c5 --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/c5_train_bench.json 2> $O/c5_train_bench.err
rm -rf $O/kt_c5t
rocprofv3 --kernel-trace --stats -d $O/kt_c5t -o k -- python $R/bench.py --workload c5 --mode train --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $O/kt_c5t.log 2>&1
f=$(find $O/kt_c5t -name "*.db" | head -1)
python $R/profiles/summarize_rocpd.py $f "round 4: python bench.py --workload c5 --mode train --steps 1 --warmup 1 --no-cpu-baseline --no-graph (C5 shard training step, bf16 storage)" > $O/c5_train_kernel_stats.txt
find $O -name "*.db" -delete
# counters of the cell launch (C2 forward)
TAG=r04cell WORKLOAD=c2 SKIP_TRAFFIC=1 $R/tools/profile_r04.sh > $O/profile_r04.log 2>&1
TAG=r04cell WORKLOAD=c2 $R/tools/pmc_mem.sh > $O/pmc_mem.log 2>&1
ls $O | head -80
