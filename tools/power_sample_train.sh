#!/bin/bash
# GPU box: board power and shader clock sampled (rocm-smi, read-only) while the TRAINING step runs back to back
# (tools/train_step_bench.py --no-sync): is the 2.0 GHz of its matrix kernels (pmc_train.sh) a power limit?
#   bash tools/power_sample_train.sh [rays=65536] [steps=400]
rays=${1:-65536}
steps=${2:-400}
python3 tools/train_step_bench.py --skip-autograd --no-sync --rays $rays --steps $steps --warmup 3 > gpurun_out/power_train_$rays.json 2>/dev/null &
pid=$!
: > gpurun_out/power_train_$rays.log
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Graphics Package Power|sclk clock level" | tr '\n' ' ' >> gpurun_out/power_train_$rays.log
  echo >> gpurun_out/power_train_$rays.log
  sleep 0.3
done
wait $pid
python3 - <<PY
import re, json
rows = []
for l in open("gpurun_out/power_train_$rays.log"):
    p = re.search(r"Power \(W\): ([0-9.]+)", l); c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)
    if p: rows.append((float(p.group(1)), int(c.group(1)) if c else -1))
busy = sorted(rows, reverse=True)[:max(1, len(rows) // 3)]          # the third of the samples with the highest power
print("training step, $rays rays: %d samples; highest-power third: %.0f-%.0f W, sclk %d-%d MHz (median %.0f W / %d MHz)" % (
    len(rows), busy[-1][0], busy[0][0], min(b[1] for b in busy), max(b[1] for b in busy),
    sorted(b[0] for b in busy)[len(busy) // 2], sorted(b[1] for b in busy)[len(busy) // 2]))
print("top samples (W, MHz):", sorted(rows, reverse=True)[:8])
d = json.load(open("gpurun_out/power_train_$rays.json"))
print("step: %.2f + %.2f ms" % (d["fused"]["forward_ms"], d["fused"]["backward_ms"]))
PY
