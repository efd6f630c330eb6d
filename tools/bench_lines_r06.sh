#!/bin/bash
# The bench lines of the round-6 profile set, taken AFTER the kernel summaries were committed (their `# csrc_sha16:` then
# equals the tree's fingerprint and bench.py quotes them): outputs in gpurun_out/r06/, copied by tools/collect_r06.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for w in c2 c4 c5 c1; do
  steps=20; [ $w != c2 ] && [ $w != c1 ] && steps=6
  python bench.py --workload $w --steps $steps --warmup 3 --train-steps $([ $w = c5 ] && echo 0 || echo 2) > $O/${w}_bench.json 2> $O/${w}_bench.err
done
python bench.py > $O/default_bench.json 2> $O/default_bench.err
python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/c2_train_bench.json 2> $O/c2_train_bench.err
python bench.py --workload c5 --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/c5_train_bench.json 2> $O/c5_train_bench.err
python - <<PY
import json
for w in ("c2","c4","c5","c1","default","c2_train","c5_train"):
    j=json.loads(open("$O/%s_bench.json"%w).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print(w, j["ms_per_step"], j.get("value"), "roofline", r.get("frac"), r.get("avg_us"), r.get("source"), r.get("profile_matches_build"))
PY
