#!/bin/bash
R=$GRAFT_REPO_ROOT
for x in 0 1; do for w in 8 16 32; do echo "== XCD=$x WPC=$w"; TSPGNN_WGRAD_XCD=$x TSPGNN_WGRAD_WPC=$w python $R/tools/wgrad_bench.py c5; done; done
echo "== c2"; for x in 0 1; do TSPGNN_WGRAD_XCD=$x python $R/tools/wgrad_bench.py c2; done
