#!/bin/bash
# round 4, GPU session 1: suite, hand-off probe, A/B of the store flavour, two passes in flight, first bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/s1_pytest.log 2>&1; echo "pytest rc $?" >> $O/s1_pytest.log
tail -5 $O/s1_pytest.log
timeout 300 tools/per_graph_handoff_probe.bin 30 > $O/handoff_probe.txt 2>&1; cat $O/handoff_probe.txt
W=c2 ROUNDS=3 timeout 900 tools/abn.sh r03base cur wt > $O/s1_ab_c2.txt 2>&1; cat $O/s1_ab_c2.txt
W=c4 ROUNDS=2 timeout 600 tools/abn.sh r03base cur wt > $O/s1_ab_c4.txt 2>&1; cat $O/s1_ab_c4.txt
timeout 300 python tools/two_in_flight.py c2 40 > $O/s1_two_in_flight.txt 2>&1; tail -2 $O/s1_two_in_flight.txt
timeout 600 python bench.py > $O/s1_c2_bench.json 2> $O/s1_c2_bench.err; tail -c 1500 $O/s1_c2_bench.json; tail -3 $O/s1_c2_bench.err
