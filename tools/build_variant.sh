#!/bin/bash
# build 3dssd_amd/csrc/variants/lib_$1.so = the product objects with ONE source recompiled with extra flags:
#   tools/build_variant.sh pd4 mlp_gemm "-DSA_GEMM_PD=4"
# (A/B experiments: run with SA3D_LIB=.../lib_pd4.so python bench.py --allow-knobs ...)
set -e
NAME=$1; SRC=$2; EXTRA=$3
cd "$(dirname "$0")/../3dssd_amd/csrc"
mkdir -p variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -fPIC"
[ "$SRC" = "mlp_rowwave" ] && FLAGS="$FLAGS -mllvm -pragma-unroll-threshold=4000000"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $SRC.hip -o variants/${SRC}_$NAME.o
OBJS=$(ls *.o | grep -v "^$SRC.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$NAME.so $OBJS variants/${SRC}_$NAME.o
echo built variants/lib_$NAME.so
