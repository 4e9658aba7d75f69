#!/bin/bash
# Marginal cost of every kernel class on the 16-stream throughput: the class's launches are skipped (results are then
# garbage; timing experiment only) and ms/step is compared with the full run.  usage: gpurun -- 'bash tools/ablate.sh'
# (ball query and FPS cannot be ablated this way: their outputs shape the work of everything downstream)
OUT=gpurun_out/ablate; mkdir -p $OUT
for a in none mlp:layer4 mlp:layer3 mlp:layer2 mlp:layer1 dense sqdist "mlp:layer1,mlp:layer2,mlp:layer3,mlp:layer4" "mlp:layer1,mlp:layer2,mlp:layer3,mlp:layer4,dense,sqdist,plan"; do
  SA_ABLATE=$a python bench.py --no-cpu-baseline --profile-iters 0 > $OUT/x.json 2>$OUT/x.err
  python -c "
import json; d=json.loads(open('$OUT/x.json').read().strip().splitlines()[-1]); print('%-70s ms/step %.4f  latency %.3f' % ('$a', d['ms_per_step'], d['single_stream_batch_latency_ms']))"
done | tee $OUT/ablate.txt
