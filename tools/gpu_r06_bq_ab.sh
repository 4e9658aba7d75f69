#!/bin/bash
# layer-1 ball query alone over library variants: bash tools/gpu_r06_bq_ab.sh "v1 v2 ..." ("product" = the in-tree library)
cd "$GRAFT_REPO_ROOT"
for d in default rings64 dense; do
  f=128; [ $d = dense ] && f=32
  for v in $1; do
    if [ "$v" != product ]; then export SA3D_LIB=$PWD/3dssd_amd/csrc/variants/lib_$v.so; else unset SA3D_LIB; fi
    chk=""; [ "$2" = check ] && chk=check
    echo -n "$v: "; python tools/bq_bench.py $f $d 7 $chk 2>&1 | grep -v amdgpu.ids
  done
done
