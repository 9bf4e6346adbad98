#!/bin/bash
# GPU box: effective clock and matrix-pipe occupancy of the shading kernels.
# clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch time; MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 4 SIMD * 256 CU)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_c; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_c -- python bench.py --steps 3 --warmup 1 --cpu-rays-side 0 > gpurun_out/pmc_c.json 2> gpurun_out/pmc_c.err
python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_c/**/*counter_collection.csv",recursive=True)[0]
rows=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "pnr::" in r["Kernel_Name"] and ("shade" in r["Kernel_Name"] or "point_part" in r["Kernel_Name"]):
        k=(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])
        rows[k][r["Counter_Name"]]=float(r["Counter_Value"]); rows[k]["ns"]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for (d,k),v in list(rows.items())[-8:]:
    gui=v.get("GRBM_GUI_ACTIVE",0)/8
    print(k, "ms %.3f"%(v["ns"]/1e6), "clock GHz %.3f"%(gui/v["ns"]), "mfma busy %.3f"%(v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(gui*1024+1e-9)))
PY
