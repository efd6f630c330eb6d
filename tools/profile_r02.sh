#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats and PMC traffic of the SpMM
# kernels for the workloads c2 (headline), c4 (ragged, operands > Infinity Cache) and c5 (one GPU's shard, bf16 storage).
# Outputs land in gpurun_out/r02/; the summaries are then copied to profiles/ (tracked).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in ${WORKLOADS:-c2 c4 c5}; do
  steps=20; [ $w != c2 ] && steps=6
  python $R/bench.py --workload $w --steps $steps --warmup 3 --train-steps $([ $w = c5 ] && echo 0 || echo 2) > $O/${w}_bench.json 2> $O/${w}_bench.err
  rm -rf $O/kt_$w
  rocprofv3 --kernel-trace --stats -d $O/kt_$w -o k -- python $R/bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline --train-steps 0 > $O/kt_$w.log 2>&1
  f=$(find $O/kt_$w -name "*.db" | head -1)
  python $R/profiles/summarize_rocpd.py $f "round 2: python bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline --train-steps 0 (forward, HIP-graph replay; includes the pre-warm replays, the alternative-arithmetic legs and the SpMM micro-loops)" > $O/${w}_forward_kernel_stats.txt
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
ls -la $O | head -40
