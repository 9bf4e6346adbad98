#!/bin/bash
# GPU box: per-kernel averages of the training step (rocprofv3 kernel trace of tools/train_step_bench.py, fused path only)
#   bash tools/train_kernel_times.sh [rays] [mode]   -> gpurun_out/train_kt_<rays>_<mode>.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAYS=${1:-65536}; MODE=${2:-fp32}
rm -rf gpurun_out/tkt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tkt -- python3 tools/train_step_bench.py --skip-autograd --steps 8 --warmup 3 --rays $RAYS --mode $MODE > gpurun_out/tkt.json 2> gpurun_out/tkt.err
python3 - "$RAYS" "$MODE" <<PY
import csv,glob,sys,shutil
f=glob.glob("gpurun_out/tkt/**/*kernel_stats.csv",recursive=True)[0]
shutil.copy(f, "gpurun_out/train_kt_%s_%s.csv" % (sys.argv[1], sys.argv[2]))
tot=0
for r in csv.DictReader(open(f)):
    if "pnr::" in r["Name"]:
        per_step=float(r["TotalDurationNs"])/11/1e3
        tot+=per_step
        if per_step>8: print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), "%9.1f us avg %9.1f us per step" % (float(r["AverageNs"])/1e3, per_step))
print("sum of pnr kernels per step (11 steps incl. warm-up): %.1f us" % tot)
PY
