#!/bin/bash
# GPU box: A/B of library variants on ONE device, interleaved rounds (separate boxes differ by several per cent).
#   bash tools/ab.sh base contig      (names of pointnerf2studio_amd/_abl/libpnr_<name>.so; "base" = the shipped library)
R=${ROUNDS:-3}
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ $v = base ]; then L=""; else L="$PWD/pointnerf2studio_amd/_abl/libpnr_$v.so"; fi
    PNR_LIB=$L python bench.py --steps 6 --warmup 2 --cpu-rays-side 0 --no-other-mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_launch']
print('round $r %-10s' % '$v', 'pairs %.3f point %.3f color %.3f knn %.3f select %.3f total %.3f' % (s['shade_pairs'], s['point_part'], s['shade_color'], s['knn'], s['select'], d['ms_per_step']))"
  done
done
