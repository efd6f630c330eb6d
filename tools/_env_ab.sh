cd $GRAFT_REPO_ROOT
run() {
    env "$@" TSPGNN_H2_WAVES=16 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 > gpurun_out/envab.json 2> gpurun_out/envab.err || tail -5 gpurun_out/envab.err
    python - "$*" <<PY
import json,sys
r=json.load(open("gpurun_out/envab.json"))
k=r.get("kernels_us",{})
print("%-50s ms_per_step %.4f  cell %.1f us  rowsum %.1f us" % (sys.argv[1], r["ms_per_step"], k.get("tspgnn_lnlstm_mlp_fwd_multi_h2",{}).get("avg_us",0), k.get("tspgnn_csr_rowsum_f32",{}).get("avg_us",0)))
PY
}
run A=0
run TSPGNN_SINGLE_MO=1
run TSPGNN_INPLACE=1
run TSPGNN_INPLACE=1 TSPGNN_SINGLE_MO=1
run A=0
python -m pytest tests/test_gpu_model.py -x -q -k "parity_with_oracle" 2>&1 | tail -3
TSPGNN_INPLACE=1 TSPGNN_SINGLE_MO=1 python -m pytest tests/test_gpu_model.py tests/test_gpu_anchors.py -x -q -k "parity_with_oracle or anchor" 2>&1 | tail -3
