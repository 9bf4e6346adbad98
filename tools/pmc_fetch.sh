#!/bin/bash
# GPU box: one rocprofv3 FETCH_SIZE pass over a short bench run; prints KiB per launch of the shading kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_q; rocprofv3 --pmc ${1:-FETCH_SIZE} --output-format csv -d gpurun_out/pmc_q -- python bench.py --steps 2 --warmup 1 --cpu-rays-side 0 > gpurun_out/pmc_q.json 2> gpurun_out/pmc_q.err
python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_q/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "pnr::" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "shade" in k or "point_part" in k or "knn" in k: print(k, [round(x/1e6,3) for x in v[-3:]], "x1e6 (KiB for *_SIZE)")
PY
