#!/bin/bash
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "mlp or backbone or pipeline or wide" -p no:cacheprovider 2>&1 | tail -2
for d in default rings64; do for i in 1 2; do timeout 300 python tools/stages_at.py 128 data=$d 2>&1 | grep "m=256\|total"; done; done
timeout 300 python tools/stages_at.py 8 2>&1 | grep "m=256\|total"
timeout 300 python tools/stages_at.py 32 2>&1 | grep "m=256\|total"
