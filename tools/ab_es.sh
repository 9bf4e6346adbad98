#!/bin/bash
# GPU box: early-termination leg of bench.py for library variants (see tools/ab.sh)
for v in "$@"; do
  if [ $v = base ]; then L=""; else L="$PWD/pointnerf2studio_amd/_abl/libpnr_$v.so"; fi
  PNR_LIB=$L python bench.py --steps 2 --warmup 1 --cpu-rays-side 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['with_early_ray_termination']; s=e['stages_ms_per_launch']
print('%-8s' % '$v', 'full %.3f ms | early: total %.3f pairs %.3f color %.3f shaded %.0f of %.0f' % (d['ms_per_step'], e['ms_per_step'], s['shade_pairs'], s['shade_color'], e['samples_shaded_per_launch'], e['samples_valid_per_launch']))"
done
