#!/bin/bash
# A/B of library variants on the single-stream stage timings: $1 = tag, rest = variant names ("base" = the product lib)
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in "$@"; do
  if [ "$v" = "base" ]; then unset SA3D_LIB; K=""; else export SA3D_LIB=$GRAFT_REPO_ROOT/3dssd_amd/csrc/variants/lib_$v.so; K="--allow-knobs"; fi
  timeout 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --no-uncoalesced --profile-iters 5 --verify 2 $K ${BENCH_EXTRA} > $OUT/$v.json 2> $OUT/$v.err
  python - <<P
import json
try:
    d = json.loads(open("$OUT/$v.json").read().strip().splitlines()[-1])
    st = {(s["kernel"], s["label"]): s["avg_ms"] for s in d["stages"]}
    mlp = [(k[1][:48], round(v * 1e3, 1)) for k, v in st.items() if k[0] == "sa_group_mlp_max_layer"]
    print("%-10s lat %.3f verify %s  " % ("$v", d["single_stream_batch_latency_ms"], d["verify"]["all_equal_eager"]), mlp, "plan", round(1e3 * sum(v for k, v in st.items() if k[0] == "sa_group_mlp_plan"), 1))
except Exception as e:
    print("$v failed", e); print(open("$OUT/$v.err").read()[-800:])
P
done
