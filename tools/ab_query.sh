#!/bin/bash
# GPU box: A/B of library variants on the query stage of the eval frame (select / knn stage times), interleaved rounds.
#   bash tools/ab_query.sh base flagchk
R=${ROUNDS:-2}
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ $v = base ]; then L=""; else L="$PWD/pointnerf2studio_amd/_abl/libpnr_$v.so"; fi
    PNR_LIB=$L python bench.py --precision fp32 --steps 6 --warmup 2 --cpu-rays-side 0 --no-other-mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms_per_launch']
print('round $r %-10s' % '$v', 'select %.3f knn %.3f point %.3f pairs %.3f total %.3f' % (s['select'], s['knn'], s['point_part'], s['shade_pairs'], d['ms_per_step']))"
  done
done
