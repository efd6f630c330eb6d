cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export TSPGNN_H2_WAVES=16
CMD="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --train-steps 0 --no-graph"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU -d $R/gpurun_out/pmc_h2a -o p -- $CMD > $R/gpurun_out/pmc_h2a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES -d $R/gpurun_out/pmc_h2b -o p -- $CMD > $R/gpurun_out/pmc_h2b.log 2>&1
cd $R
for d in pmc_h2a pmc_h2b; do f=$(find gpurun_out/$d -name "*.db" | head -1); python profiles/summarize_pmc.py $f lnlstm_mlp_fwd_h2 > gpurun_out/$d.txt; cat gpurun_out/$d.txt; done
find gpurun_out/pmc_h2a gpurun_out/pmc_h2b -name "*.db" -delete
