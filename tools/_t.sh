cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in base pf2 mmc mmc_np pf2_mmc ilp; do
TSPGNN_LIB=$PWD/tools/variants/$v.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 > gpurun_out/b.json 2> gpurun_out/b.err || tail -3 gpurun_out/b.err; python - <<PY
import json
r=json.load(open("gpurun_out/b.json")); print("$v ms", r["ms_per_step"], r["kernels_us"]["tspgnn_lnlstm_mlp_fwd_multi_h2"]["avg_us"])
PY
done; done
