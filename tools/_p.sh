cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
rm -rf $O/kt_train
rocprofv3 --kernel-trace --stats -d $O/kt_train -o k -- python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_train.log 2>&1
f=$(find $O/kt_train -name "*.db" | head -1)
python $R/profiles/summarize_rocpd.py $f "round 2 (a): python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline (C2 training step, HIP-graph replay)" > $O/c2_train_kernel_stats.txt
find $O -name "*.db" -delete
head -40 $O/c2_train_kernel_stats.txt | cut -c1-175
tail -2 $O/kt_train.log | cut -c1-300
