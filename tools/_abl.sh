cd $GRAFT_REPO_ROOT
for m in 15 79 143 271 527 207 463 975 31; do
    TSPGNN_LIB=$PWD/tools/variants/abl$m.so TSPGNN_H2_WAVES=16 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 > gpurun_out/abl_$m.json 2> gpurun_out/abl_$m.err || tail -5 gpurun_out/abl_$m.err
    python - <<PY
import json
r=json.load(open("gpurun_out/abl_$m.json"))
k=r.get("kernels_us",{})
print("abl %s: ms_per_step %.4f  cell %.1f us" % ("$m", r["ms_per_step"], k.get("tspgnn_lnlstm_mlp_fwd_multi_h2",{}).get("avg_us",0)))
PY
done
