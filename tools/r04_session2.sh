#!/bin/bash
# round 4, GPU session 2: low-end test printouts, store-flavour A/B (5 rounds), packer ranks on the GPU box's host
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -x -q -s -k "small_activations or variance_floor or stale_range or bucket" > $O/s2_lowend.log 2>&1
grep -a "low end\|variance floor\|passed\|failed" $O/s2_lowend.log
W=c2 ROUNDS=5 timeout 1200 tools/abn.sh cur wt1 wt2 wt3 > $O/s2_ab_c2.txt 2>&1; cat $O/s2_ab_c2.txt
W=c4 ROUNDS=3 timeout 900 tools/abn.sh cur wt1 wt2 wt3 > $O/s2_ab_c4.txt 2>&1; cat $O/s2_ab_c4.txt
timeout 300 python tools/pack_ranks.py 8 20 > $O/s2_pack_ranks.json 2>&1; cat $O/s2_pack_ranks.json
timeout 300 python tools/pack_ranks.py 16 20 >> $O/s2_pack_ranks.json 2>&1; tail -1 $O/s2_pack_ranks.json
