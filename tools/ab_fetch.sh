#!/bin/bash
# GPU box: FETCH_SIZE per launch of the pair kernel for library variants:  bash tools/ab_fetch.sh <fp32|bf16x3> base xc4 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mode=$1; shift
for v in "$@"; do
  if [ $v = base ]; then export PNR_LIB=""; else export PNR_LIB="$PWD/pointnerf2studio_amd/_abl/libpnr_$v.so"; fi
  rm -rf gpurun_out/pmc_ab
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_ab -- python3 bench.py --steps 2 --warmup 1 --cpu-rays-side 0 --no-other-mode --precision $mode > /dev/null 2> gpurun_out/pmc_ab.err
  python3 - "$v" "$mode" <<PY
import csv,glob,collections,sys
f=glob.glob("gpurun_out/pmc_ab/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_shade_pairs" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(sys.argv[2], sys.argv[1], k, "FETCH_SIZE KiB/launch", round(sum(v)/len(v)), "-> GB read", round(2*1024*sum(v)/len(v)/1e9,2))
PY
done
