#!/bin/bash
# round 5: one poller per workgroup in ffps_fly.hip -- parity and the matrix-free F-FPS probe
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "ffps or fly" -p no:cacheprovider 2>&1 | tail -2
timeout 300 python tools/ffps_fly_probe.py 2>&1 | grep -v amdgpu.ids
