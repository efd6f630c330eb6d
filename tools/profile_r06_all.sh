#!/bin/bash
# Round-6 profile set (run on the GPU box through gpurun; outputs in gpurun_out/r06/; the summaries are copied to
# profiles/r06_* by `python tools/collect_r06.py`).  Every rocprofv3 kernel summary starts with `# csrc_sha16: <hash>` = the
# fingerprint of the kernel sources it was taken from (bench.csrc_fingerprint): bench.py refuses to lead with a summary
# whose fingerprint is not the tree's.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
SHA=$(cd $R && python -c "import bench; print(bench.csrc_fingerprint())")
cd /tmp; export TMPDIR=/tmp
stats() {  # db, title, out
  { echo "# csrc_sha16: $SHA"; python $R/profiles/summarize_rocpd.py "$1" "$2"; } > "$3"
}
for w in ${WORKLOADS:-c2 c4 c5 c1}; do
  steps=20; [ $w != c2 ] && [ $w != c1 ] && steps=6
  python $R/bench.py --workload $w --steps $steps --warmup 3 --train-steps $([ $w = c5 ] && echo 0 || echo 2) > $O/${w}_bench.json 2> $O/${w}_bench.err
  rm -rf $O/kt_$w
  rocprofv3 --kernel-trace --stats -d $O/kt_$w -o k -- python $R/tools/forward_graph.py $w 20 > $O/kt_$w.log 2>&1
  f=$(find $O/kt_$w -name "*.db" | head -1)
  stats $f "round 6: python tools/forward_graph.py $w 20 (HIP-graph replays of the forward pass, nothing else)" $O/${w}_forward_kernel_stats.txt
  python $R/tools/timeline_rocpd.py $f > $O/${w}_forward_timeline.txt 2>&1
  if [ $w != c1 ]; then
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
  fi
  find $O -name "*.db" -delete
done
# the one-launch loops: register-resident where the selector takes it (C1) and forced at C2; memory-resident forced at C2
# and at 192 instances (inside the selector's window): kernel stats of the replayed forward
for spec in "c1 loop 3" "c2 loop 4" "c2 resident 4"; do
  set -- $spec
  rm -rf $O/kt_$2_$1
  TSPGNN_LOOP_KIND=$2 TSPGNN_LOOP_MAX_TILES=$3 rocprofv3 --kernel-trace --stats -d $O/kt_$2_$1 -o k -- python $R/tools/forward_graph.py $1 20 > $O/kt_$2_$1.log 2>&1
  f=$(find $O/kt_$2_$1 -name "*.db" | head -1)
  stats $f "round 6: TSPGNN_LOOP_KIND=$2 TSPGNN_LOOP_MAX_TILES=$3 python tools/forward_graph.py $1 20" $O/${1}_$2_kernel_stats.txt
  find $O -name "*.db" -delete
done
# training steps
python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/c2_train_bench.json 2> $O/c2_train_bench.err
rm -rf $O/kt_c2t
rocprofv3 --kernel-trace --stats -d $O/kt_c2t -o k -- python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_c2t.log 2>&1
f=$(find $O/kt_c2t -name "*.db" | head -1)
stats $f "round 6: python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline (C2 training step, HIP-graph replay)" $O/c2_train_kernel_stats.txt
python $R/bench.py --workload c5 --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/c5_train_bench.json 2> $O/c5_train_bench.err
rm -rf $O/kt_c5t
rocprofv3 --kernel-trace --stats -d $O/kt_c5t -o k -- python $R/bench.py --workload c5 --mode train --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $O/kt_c5t.log 2>&1
f=$(find $O/kt_c5t -name "*.db" | head -1)
stats $f "round 6: python bench.py --workload c5 --mode train --steps 1 --warmup 1 --no-cpu-baseline --no-graph (C5 shard training step, bf16 storage; 1 + 1 steps)" $O/c5_train_kernel_stats.txt
# the recomputing message-MLP backward (opt-in) against the default, alternating; the kernels alone; the weight-gradient
# reduction at the C5 shapes
{
  for rep in 1 2; do for rc in 0 1; do
    echo -n "TSPGNN_RECOMPUTE=$rc (weight gradients in the launch): "
    TSPGNN_RECOMPUTE=$rc python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'], 'ms per C2 training step')"
  done; done
  for v in 0 1; do
    echo -n "TSPGNN_MLP_BWD_H2=$v: "
    TSPGNN_MLP_BWD_H2=$v python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'], 'ms per C2 training step')"
  done
  RC_DW=1 python $R/tools/rc_bench.py 50
  python $R/tools/wgrad_bench.py c5; python $R/tools/wgrad_bench.py c2
} 2>&1 | grep -v amdgpu > $O/train_variants.txt
find $O -name "*.db" -delete
# one-launch forms vs step-by-step over batch sizes (what loop_plan.max_edge_tiles and resident_plan.in_auto_window were set
# from), the memory-resident loop's phase trace, the bound probe's numbers are in profiles/r06_loop_bound.txt
{
  for spec in "32 20 8" "32 40 32" "64 40 32" "96 40 32" "128 40 32"; do
    echo "## TSPGNN_LOOP_KIND=loop  B n T = $spec"; TSPGNN_LOOP_KIND=loop TSPGNN_LOOP_MAX_TILES=4 timeout 300 python $R/tools/loop_bench.py $spec 2 2>&1 | grep -v amdgpu | tail -6
  done
} > $O/loop_vs_steps.txt
{
  for spec in "32 40 32" "64 40 32" "96 40 32" "128 40 32" "160 40 32" "192 40 32" "224 40 32" "256 40 32"; do
    echo "## TSPGNN_LOOP_KIND=resident  B n T = $spec"; TSPGNN_LOOP_KIND=resident timeout 300 python $R/tools/loop_bench.py $spec 3 2>&1 | grep -v amdgpu | tail -8
  done
  echo "## TSPGNN_RES_SAFE=1 (agent-scope acquire instead of the L1-only invalidate)  128 40 32"; TSPGNN_RES_SAFE=1 TSPGNN_LOOP_KIND=resident LOOP_BENCH_MODES=loop timeout 300 python $R/tools/loop_bench.py 128 40 32 2 2>&1 | grep -v amdgpu | tail -2
  for c in 1 2 3; do echo "## TSPGNN_RES_CLASSES=$c  128 40 32"; TSPGNN_RES_CLASSES=$c TSPGNN_LOOP_KIND=resident LOOP_BENCH_MODES=loop timeout 300 python $R/tools/loop_bench.py 128 40 32 2 2>&1 | grep -v amdgpu | tail -2; done
} > $O/resident_vs_steps.txt
RES_TRACE_XCD=0 timeout 300 python $R/tools/resident_trace.py 128 40 32 > $O/resident_trace.txt 2>&1
timeout 300 python $R/tools/stager_breakdown.py 2>&1 | grep -v amdgpu > $O/stager_breakdown.txt
ANCHORS=c2,c2t8,c1 GEMMS=f16x2,bf16x3,f32 TOP=5 timeout 900 python $R/tools/grad_anchor_report.py 2>&1 | grep -v amdgpu > $O/grad_anchor_report.txt
{ echo "## GRAPHS=192 (default selector: tspgnn_mp_resident_h2)"; GRAPHS=192 NFWD=3000 NTRAIN=40 timeout 600 python $R/tools/soak.py 2>&1 | grep -v amdgpu; echo "## TSPGNN_LOOP_KIND=resident GRAPHS=128"; TSPGNN_LOOP_KIND=resident NFWD=5000 NTRAIN=40 timeout 600 python $R/tools/soak.py 2>&1 | grep -v amdgpu; echo "## GRAPHS=32 (register-resident loop)"; GRAPHS=32 NFWD=5000 NTRAIN=40 timeout 600 python $R/tools/soak.py 2>&1 | grep -v amdgpu; echo "## default C2"; NFWD=3000 NTRAIN=300 timeout 900 python $R/tools/soak.py 2>&1 | grep -v amdgpu; } > $O/soak.txt 2>&1
timeout 400 python $R/tools/rowsum_once_bound.py 2>&1 | grep -v amdgpu > $O/rowsum_once_bound.txt
# randomised parity sweeps: default path, the opt-in recomputing backward (both forms), bf16 storage, determinism
{
echo "## python tests/fuzz_parity.py 150 5"; timeout 900 python $R/tests/fuzz_parity.py 150 5 2>&1 | grep -v amdgpu | tail -12
echo "## TSPGNN_RECOMPUTE=1 python tests/fuzz_parity.py 90 7   (recomputing message-MLP backward, weight gradients in the launch)"; TSPGNN_RECOMPUTE=1 timeout 900 python $R/tests/fuzz_parity.py 90 7 2>&1 | grep -v amdgpu | tail -8
echo "## TSPGNN_LOOP_KIND=resident python tests/fuzz_parity.py 100 17   (every batch through the memory-resident one-launch loop)"; TSPGNN_LOOP_KIND=resident timeout 900 python $R/tests/fuzz_parity.py 100 17 2>&1 | grep -v amdgpu | tail -6
echo "## TSPGNN_LOOP_KIND=loop TSPGNN_LOOP_MAX_TILES=4 python tests/fuzz_parity.py 100 19   (every batch through the register-resident loop)"; TSPGNN_LOOP_KIND=loop TSPGNN_LOOP_MAX_TILES=4 timeout 900 python $R/tests/fuzz_parity.py 100 19 2>&1 | grep -v amdgpu | tail -6
echo "## BF16=1 python tests/fuzz_parity.py 60 11"; BF16=1 timeout 900 python $R/tests/fuzz_parity.py 60 11 2>&1 | grep -v amdgpu | tail -8
echo "## DET=1 python tests/fuzz_parity.py 40 13"; DET=1 timeout 900 python $R/tests/fuzz_parity.py 40 13 2>&1 | grep -v amdgpu | tail -6
echo "## DET=1 TSPGNN_RECOMPUTE=1 python tests/fuzz_parity.py 30 15"; DET=1 TSPGNN_RECOMPUTE=1 timeout 900 python $R/tests/fuzz_parity.py 30 15 2>&1 | grep -v amdgpu | tail -6
} > $O/fuzz_parity.txt 2>&1
# counters of the cell launch (C2 forward)
TAG=r06cell WORKLOAD=c2 SKIP_TRAFFIC=1 $R/tools/profile_r04.sh > $O/profile_cell.log 2>&1
ls $O | head -80
