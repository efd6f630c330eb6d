# usage: bash tools/_ab.sh "<variant names>" [waves...]
cd $GRAFT_REPO_ROOT
for v in $1; do
  for w in ${2:-16}; do
    TSPGNN_LIB=$PWD/tools/variants/$v.so TSPGNN_H2_WAVES=$w python bench.py --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || tail -5 gpurun_out/ab_$v.err
    python - <<PY
import json
r=json.load(open("gpurun_out/ab_$v.json"))
k=r.get("kernels_us",{})
print("%-24s waves $w: ms_per_step %.4f  cell %.1f us  rowsum %.1f us" % ("$v", r["ms_per_step"], k.get("tspgnn_lnlstm_mlp_fwd_multi_h2",{}).get("avg_us",0), k.get("tspgnn_csr_rowsum_f32",{}).get("avg_us",0)))
PY
  done
done
