#!/bin/bash
# GPU box: the effective shader clock (tools/ub_clock_probe.hip, a resident one-wave probe of another process, 2-ms windows) while
# (a) nothing else runs, (b) the eval frame loop runs, (c) the 65 536-ray training-step loop runs.
#   bash tools/clock_probe_run.sh
cd "$GRAFT_REPO_ROOT"
P=tools/bin/ub_clock_probe
[ -x $P ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ub_clock_probe.hip -o $P || exit 1
summ() {   # file, label, t0_ms, t1_ms: the windows between t0 and t1
python3 - "$@" <<'PY'
import sys
f, label, t0, t1 = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])
v = [float(l.split()[2]) for l in open(f) if l[0] != '#' and t0 <= float(l.split()[1]) <= t1]
v.sort()
print("%-34s %3d windows: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f MHz" % (
    label, len(v), v[0], v[len(v) // 10], v[len(v) // 2], v[(9 * len(v)) // 10], v[-1]) if v else label + ": no windows")
PY
}
$P 1.5 2 > gpurun_out/probe_idle.txt
summ gpurun_out/probe_idle.txt "idle device" 200 1400
# (b) eval frames: the probe first, then ~6 s of frames
$P 14 2 > gpurun_out/probe_eval.txt &
pp=$!
sleep 0.5
python bench.py --steps 170 --warmup 2 --cpu-rays-side 0 --no-other-mode --no-cfg0 > gpurun_out/probe_bench_eval.json 2> /dev/null
wait $pp
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/probe_bench_eval.json"))
print("eval loop: %.2f ms/frame, pair kernel %.2f ms" % (d["ms_per_step"], d["stages_ms_per_launch"]["shade_pairs"]))
PY
cp gpurun_out/probe_eval.txt gpurun_out/probe_eval_full.txt
# the frame loop is the last ~6.3 s before bench.py ends; the probe started 0.5 s before it: take the busiest stretch
python3 - <<'PY'
rows = [(float(l.split()[1]), float(l.split()[2])) for l in open("gpurun_out/probe_eval.txt") if l[0] != '#']
# windows are 2 ms: a 3-s stretch = 1500 windows; the stretch with the lowest mean clock is the loaded one
N = 1500
pre = [0.0]
for r in rows: pre.append(pre[-1] + r[1])
best = min(((pre[i + N] - pre[i]) / N, i) for i in range(0, max(1, len(rows) - N)))
v = sorted(r[1] for r in rows[best[1]:best[1] + N])
print("eval frame loop (3-s stretch with the lowest mean clock, from %.1f s): min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f MHz" % (
    rows[best[1]][0] / 1e3, v[0], v[N // 10], v[N // 2], v[9 * N // 10], v[-1]))
PY
# (c) training steps
$P 14 2 > gpurun_out/probe_train.txt &
pp=$!
sleep 0.5
python3 tools/train_step_bench.py --skip-autograd --no-sync --rays 65536 --steps 330 --warmup 3 > gpurun_out/probe_bench_train.json 2> /dev/null
wait $pp
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/probe_bench_train.json"))
print("training loop: %.2f + %.2f ms per step" % (d["fused"]["forward_ms"], d["fused"]["backward_ms"]))
rows = [(float(l.split()[1]), float(l.split()[2])) for l in open("gpurun_out/probe_train.txt") if l[0] != '#']
N = 1500
pre = [0.0]
for r in rows: pre.append(pre[-1] + r[1])
best = min(((pre[i + N] - pre[i]) / N, i) for i in range(0, max(1, len(rows) - N)))
v = sorted(r[1] for r in rows[best[1]:best[1] + N])
print("training-step loop (3-s stretch with the lowest mean clock, from %.1f s): min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f MHz" % (
    rows[best[1]][0] / 1e3, v[0], v[N // 10], v[N // 2], v[9 * N // 10], v[-1]))
PY
