#!/bin/bash
# Development helper: builds libtspgnn variants for A/B runs on the GPU box.
#   [FILE=dense_bwd_h2] [SRC=path/to/copy.hip] tools/build_variant.sh NAME <extra hipcc flags for that file (default dense_h2.hip)>
#   ->  tools/variants/NAME.so  (use with TSPGNN_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/tools/variants/obj_$NAME
cd $ROOT/tsp-gnn_amd/csrc
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -Wall -Wno-unused-function "$@" -I$ROOT/tsp-gnn_amd/csrc -c ${SRC:-${FILE:-dense_h2}.hip} -o $ROOT/tools/variants/obj_$NAME/${FILE:-dense_h2}.o
OBJS=$(ls build/*.o | grep -v "/${FILE:-dense_h2}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/variants/$NAME.so $OBJS $ROOT/tools/variants/obj_$NAME/${FILE:-dense_h2}.o
echo built tools/variants/$NAME.so
