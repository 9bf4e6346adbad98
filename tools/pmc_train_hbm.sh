#!/bin/bash
# GPU box: HBM traffic of the training step's kernels (tools/train_step_bench.py, fused path): one rocprofv3 --pmc pass per
# counter (FETCH_SIZE, WRITE_SIZE: KiB at the L2's memory side; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
# gfx950).   bash tools/pmc_train_hbm.sh [rays] -> gpurun_out/pmc_train_hbm_<rays>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAYS=${1:-65536}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_train_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_train_$c -- python3 tools/train_step_bench.py --skip-autograd --steps 2 --warmup 1 --rays $RAYS > gpurun_out/pmc_train_$c.json 2> gpurun_out/pmc_train_$c.err
done
python3 - "$RAYS" <<'PY' > gpurun_out/pmc_train_hbm_$RAYS.txt
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_train_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "pnr::" in n and r["Counter_Name"] == c:
            acc[n[:60]][c].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("%-62s %4s %9s %9s %9s %9s" % ("kernel (last 2 launches = the timed steps)", "n", "us", "read GB", "write GB", "TB/s"))
tot = [0.0, 0.0, 0.0]
for n, d in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].get("FETCH_SIZE", [])[-2:])):
    fe, wr = d.get("FETCH_SIZE", [])[-2:], d.get("WRITE_SIZE", [])[-2:]
    if not fe or not wr:
        continue
    us = sum(v[1] for v in fe) / len(fe) / 1e3
    rd = 2 * sum(v[0] for v in fe) / len(fe) * 1024 / 1e9
    ww = sum(v[0] for v in wr) / len(wr) * 1024 / 1e9
    if us < 20:
        continue
    tot[0] += rd; tot[1] += ww; tot[2] += us
    print("%-62s %4d %9.1f %9.3f %9.3f %9.2f" % (n, len(d["FETCH_SIZE"]), us, rd, ww, (rd + ww) / us * 1e3))
print("%-62s %4s %9.1f %9.3f %9.3f %9.2f" % ("sum (kernels above)", "", tot[2], tot[0], tot[1], (tot[0] + tot[1]) / tot[2] * 1e3))
PY
cat gpurun_out/pmc_train_hbm_$RAYS.txt
