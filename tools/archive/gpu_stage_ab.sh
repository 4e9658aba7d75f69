#!/bin/bash
# A/B of library variants on ONE stage (single-stream stage timing): $1 = tag, $2 = substring of the stage's kernel name,
# rest = variant names ("base" = the product lib)
TAG=$1; PAT=$2; shift; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in "$@"; do
  if [ "$v" = "base" ]; then unset SA3D_LIB; K=""; else export SA3D_LIB=$GRAFT_REPO_ROOT/3dssd_amd/csrc/variants/lib_$v.so; K="--allow-knobs"; fi
  timeout 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --no-uncoalesced --profile-iters 8 --verify ${VERIFY:-2} $K ${BENCH_EXTRA} > $OUT/$v.json 2> $OUT/$v.err
  python - <<P
import json
try:
    d = json.loads(open("$OUT/$v.json").read().strip().splitlines()[-1])
    st = [(s["label"][:40], round(s["avg_ms"] * 1e3, 1)) for s in d["stages"] if "$PAT" in s["kernel"]]
    print("%-10s lat %.3f verify %s  " % ("$v", d["single_stream_batch_latency_ms"], (d.get("verify") or {}).get("all_equal_eager")), st)
except Exception as e:
    print("$v failed", e); print(open("$OUT/$v.err").read()[-800:])
P
done
