#!/bin/bash
# marginal cost of every kernel class on the pipelined throughput (tools/ablate.py), one process per configuration
export TMPDIR=/tmp PYTHONUNBUFFERED=1
A="mlp:layer1,mlp:layer2,mlp:layer3,mlp:layer4"
for c in none mlp:layer4 mlp:layer3 mlp:layer2 mlp:layer1 dense sqdist $A $A,dense,sqdist none; do
  timeout 120 python tools/ablate.py $c 2>&1 | grep "ms/step\|Error\|error" | tail -2
done
