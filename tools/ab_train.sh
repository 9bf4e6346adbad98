#!/bin/bash
# GPU box: A/B of library variants on the training step (65536 rays), one device, interleaved rounds.
#   bash tools/ab_train.sh base tk32     (names of pointnerf2studio_amd/_abl/libpnr_<name>.so; "base" = the shipped library)
R=${ROUNDS:-2}
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ $v = base ]; then L=""; else L="$PWD/pointnerf2studio_amd/_abl/libpnr_$v.so"; fi
    PNR_LIB=$L python tools/train_step_bench.py --skip-autograd --steps 8 --rays ${RAYS:-65536} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['fused']
print('round $r %-10s' % '$v', 'forward %.3f backward %.3f ms' % (d['forward_ms'], d['backward_ms']))"
  done
done
