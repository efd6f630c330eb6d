#!/bin/bash
# Memory-path counters of an eager forward (TA / L1 / L2 / fabric): TAG=... WORKLOAD=c2 tools/pmc_mem.sh
R=$GRAFT_REPO_ROOT; TAG=${TAG:-mem}; W=${WORKLOAD:-c2}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
pass() { local name=$1; shift; rm -rf $O/raw_$name
  rocprofv3 --kernel-trace --pmc "$@" -d $O/raw_$name -o p -- python $R/tools/forward_only.py $W 3 > $O/raw_$name.log 2>&1
  local f=$(find $O/raw_$name -name "*.db" | head -1)
  { echo "# --pmc $*"; python $R/profiles/summarize_pmc.py $f ${FILTER:-lnlstm_mlp_fwd_h2}; } > $O/${W}_$name.txt; rm -rf $O/raw_$name; cat $O/${W}_$name.txt; }
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pass tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum
pass tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum
pass tcc3 TCC_TAG_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_CYCLE_sum
pass grbm GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
