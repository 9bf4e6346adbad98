#!/bin/bash
# GPU box: A/B of an ENVIRONMENT switch of the library on the training step, one device, interleaved rounds.
#   bash tools/ab_env.sh PNR_WGRAD_FULL 1 0      (RAYS=4096|65536, ROUNDS=2)
VAR=$1; shift
R=${ROUNDS:-2}
for r in $(seq 1 $R); do
  for v in "$@"; do
    env $VAR=$v python tools/train_step_bench.py --skip-autograd --steps 8 --rays ${RAYS:-65536} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['fused']
print('round $r $VAR=%-3s' % '$v', 'rays ${RAYS:-65536} forward %.3f backward %.3f ms' % (d['forward_ms'], d['backward_ms']))"
  done
done
