#!/bin/bash
# GPU box: PMC pass over the training step's kernels (tools/train_step_bench.py, fused path only): matrix-pipe busy, effective
# clock, wave wait / issue-stall / active shares, LDS conflicts.   bash tools/pmc_train.sh [rays] -> gpurun_out/pmc_train_<rays>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAYS=${1:-65536}
rm -rf gpurun_out/pmc_train
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d gpurun_out/pmc_train -- python3 tools/train_step_bench.py --skip-autograd --steps 2 --warmup 1 --rays $RAYS > gpurun_out/pmc_train.json 2> gpurun_out/pmc_train.err
python3 - "$RAYS" <<'PY' > gpurun_out/pmc_train_$RAYS.txt
import csv, glob, sys
from collections import defaultdict
f = glob.glob("gpurun_out/pmc_train/**/*counter_collection.csv", recursive=True)[0]
rows = defaultdict(dict)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "pnr::" not in n:
        continue
    k = (int(r["Dispatch_Id"]), n[:60])
    rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
    rows[k]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
per = defaultdict(list)
for (d, n), v in sorted(rows.items()):
    per[n].append(v)
print("%-62s %5s %9s %6s %6s %6s %6s %6s %8s" % ("kernel", "n", "us", "GHz", "mfma", "wait", "stall", "active", "ldsconf"))
for n, vs in sorted(per.items(), key=lambda kv: -sum(v["ns"] for v in kv[1])):
    ns = sum(v["ns"] for v in vs)
    if ns / len(vs) < 20000:
        continue
    gui = sum(v.get("GRBM_GUI_ACTIVE", 0) for v in vs) / 8
    wc = sum(v.get("SQ_WAVE_CYCLES", 0) for v in vs) + 1e-9
    print("%-62s %5d %9.1f %6.3f %6.3f %6.3f %6.3f %6.3f %8.3f" % (
        n, len(vs), ns / len(vs) / 1e3, gui / ns, sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for v in vs) / (gui * 1024 + 1e-9),
        sum(v.get("SQ_WAIT_ANY", 0) for v in vs) / wc, sum(v.get("SQ_WAIT_INST_ANY", 0) for v in vs) / wc,
        sum(v.get("SQ_ACTIVE_INST_ANY", 0) for v in vs) / wc,
        sum(v.get("SQ_LDS_BANK_CONFLICT", 0) for v in vs) / (sum(v.get("SQ_LDS_IDX_ACTIVE", 0) for v in vs) + 1e-9)))
PY
cat gpurun_out/pmc_train_$RAYS.txt
