#!/bin/bash
# GPU box: average time of ONE kernel of the training step for library variants (rocprofv3 kernel trace), one device.
#   KERNEL=k_wgrad_batch96 RAYS=65536 bash tools/ab_train_kernel.sh base nolds noglob
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ $v = base ]; then L=""; else L="$GRAFT_REPO_ROOT/pointnerf2studio_amd/_abl/libpnr_$v.so"; fi
  rm -rf gpurun_out/abk
  PNR_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk -- python3 tools/train_step_bench.py --skip-autograd --steps 6 --warmup 2 --rays ${RAYS:-65536} > /dev/null 2> gpurun_out/abk.err
  python3 - "$v" "${KERNEL:-k_wgrad}" <<PY
import csv,glob,sys
f=glob.glob("gpurun_out/abk/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"]: print("%-8s %-60s %9.1f us avg (%s calls)" % (sys.argv[1], r["Name"][:60], float(r["AverageNs"])/1e3, r["Calls"]))
PY
done
