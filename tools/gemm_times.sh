#!/bin/bash
# GPU box: per-launch times of the backward's GEMM kernels for library variants (65536-ray training step).
#   bash tools/gemm_times.sh base gabl1 gabl2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ $v = base ]; then export PNR_LIB=""; else export PNR_LIB="$PWD/pointnerf2studio_amd/_abl/libpnr_$v.so"; fi
  rm -rf gpurun_out/gt_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gt_$v -- python tools/train_step_bench.py --skip-autograd --steps 4 --warmup 2 --rays ${RAYS:-65536} > /dev/null 2> gpurun_out/gt_$v.err
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/gt_$v/**/*kernel_stats.csv",recursive=True)[0]
print("== $v")
for r in csv.DictReader(open(f)):
    if "gemm" in r["Name"] or "k_train" in r["Name"]: print("  ", r["Name"][:60].ljust(60), r["Calls"].rjust(5), "%10.1f us avg" % (float(r["AverageNs"])/1e3))
PY
done
