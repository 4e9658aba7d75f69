#!/bin/bash
# round 6: the randomised parity sweep on the final code (tests/fuzz_ops.py), small + mid-size shapes
OUT=gpurun_out/r06_fuzz; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 400 python tests/fuzz_ops.py 240 20260926 > $OUT/fuzz_small.txt 2>&1; tail -3 $OUT/fuzz_small.txt
timeout 300 python tests/fuzz_ops.py 150 505 big > $OUT/fuzz_big.txt 2>&1; tail -3 $OUT/fuzz_big.txt
echo "== done"
