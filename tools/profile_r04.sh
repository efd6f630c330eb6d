#!/bin/bash
# Round-4 counter passes for the cell + message launch (and everything else) of an eager forward, on the GPU box:
#   TAG=r04a WORKLOAD=c2 tools/profile_r04.sh      -> gpurun_out/$TAG/{pmc_mfma,pmc_lds,pmc_FETCH_SIZE,pmc_WRITE_SIZE}.txt, kernel stats, timeline
# PMC passes carry --kernel-trace only (no other trace domain).  Summaries are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r04}
W=${WORKLOAD:-c2}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
pass() {   # name, counters...
  local name=$1; shift
  rm -rf $O/raw_$name
  rocprofv3 --kernel-trace --pmc "$@" -d $O/raw_$name -o p -- python $R/tools/forward_only.py $W 3 > $O/raw_$name.log 2>&1
  local f=$(find $O/raw_$name -name "*.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --pmc $* -- python tools/forward_only.py $W 3   (eager forward passes, per-dispatch means)"; python $R/profiles/summarize_pmc.py $f; } > $O/${W}_pmc_$name.txt
  rm -rf $O/raw_$name
}
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU
pass lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM
if [ -z "$SKIP_TRAFFIC" ]; then
  pass FETCH_SIZE FETCH_SIZE
  pass WRITE_SIZE WRITE_SIZE
fi
rm -rf $O/kt
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --train-steps 0 > $O/kt.log 2>&1
f=$(find $O/kt -name "*.db" | head -1)
python $R/profiles/summarize_rocpd.py $f "round 4 ($TAG): python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --train-steps 0" > $O/${W}_forward_kernel_stats.txt
python $R/tools/timeline_rocpd.py $f > $O/${W}_forward_timeline.txt 2>&1
rm -rf $O/kt
ls -la $O
