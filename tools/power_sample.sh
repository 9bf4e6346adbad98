#!/bin/bash
# GPU box: board power and shader clock sampled (rocm-smi, read-only) while bench.py runs many steps of one mode;
# prints the samples with the highest power
mode=${1:-bf16x3}
python bench.py --steps 1200 --warmup 2 --cpu-rays-side 0 --no-other-mode --precision $mode > gpurun_out/power_bench_$mode.json 2>/dev/null &
pid=$!
: > gpurun_out/power_$mode.log
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Graphics Package Power|sclk clock level" | tr '\n' ' ' >> gpurun_out/power_$mode.log
  echo >> gpurun_out/power_$mode.log
  sleep 0.3
done
wait $pid
python - <<PY
import re
rows=[]
for l in open("gpurun_out/power_$mode.log"):
    p=re.search(r"Power \(W\): ([0-9.]+)", l); c=re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)
    if p: rows.append((float(p.group(1)), int(c.group(1)) if c else -1))
rows.sort(reverse=True)
print("$mode top samples (W, sclk MHz):", rows[:8], "of", len(rows))
PY
python -c "
import json; d=json.load(open('gpurun_out/power_bench_$mode.json')); print('$mode', round(d['ms_per_step'],3), 'ms/step', round(d['stages_ms_per_launch']['shade_pairs'],3), 'ms pairs')"
