# Diagnostic: per-rank step time of an N-rank run emulated on ONE GPU (no collectives), against the N=1 step time.
# usage (GPU box): bash tools/emulate_scaling.sh 8
N=${1:-8}
python bench.py --steps 4 --warmup 1 --cpu-rays-side 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=1 ms_per_step', round(d['ms_per_step'],3))"
for r in $(seq 0 $((N-1))); do
  python bench.py --steps 4 --warmup 1 --cpu-rays-side 0 --emulate-world $N --emulate-rank $r 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('emulated rank $r of $N ms_per_step', round(d['ms_per_step'],3), 'pairs', int(d['counters_per_launch']['pairs_valid']))"
done
