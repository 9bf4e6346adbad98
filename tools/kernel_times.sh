#!/bin/bash
# GPU box: average duration of every pnr:: kernel over a short bench run (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python bench.py --steps 4 --warmup 1 --cpu-rays-side 0 --no-other-mode > gpurun_out/kt.json 2> gpurun_out/kt.err
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/kt/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "pnr::" in r["Name"] or "rocclr" in r["Name"]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%10.1f us avg" % (float(r["AverageNs"])/1e3))
PY
