mkdir -p gpurun_out
for a in ${ABLS:-0 8 32 64}; do
  if [ $a = 0 ]; then L=""; else L="$PWD/pointnerf2studio_amd/_abl/libpnr_abl$a.so"; fi
  PNR_LIB=$L python bench.py --steps 3 --warmup 1 --cpu-rays-side 0 --precision bf16x3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('abl $a', 'pairs_ms %.2f color_ms %.2f total %.2f' % (d['stages_ms_per_launch']['shade_pairs'], d['stages_ms_per_launch']['shade_color'], d['ms_per_step']))"
done
