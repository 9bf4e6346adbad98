#!/bin/bash
# GPU box: tools/ab.sh for the exact fp32 mode
R=${ROUNDS:-2}
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ $v = base ]; then L=""; else L="$PWD/pointnerf2studio_amd/_abl/libpnr_$v.so"; fi
    PNR_LIB=$L python bench.py --precision fp32 --steps 3 --warmup 1 --cpu-rays-side 0 --no-other-mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_launch']
print('round $r %-10s' % '$v', 'pairs %.3f point %.3f color %.3f total %.3f' % (s['shade_pairs'], s['point_part'], s['shade_color'], d['ms_per_step']))"
  done
done
