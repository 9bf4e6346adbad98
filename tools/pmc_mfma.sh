#!/bin/bash
# GPU box: matrix-pipe occupancy and effective clock of the MLP kernels of both arithmetic modes -> gpurun_out/pmc_mfma_busy.json
#   clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch time;  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 4 SIMD * 256 CU)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in bf16x3 fp32; do
  rm -rf gpurun_out/pmc_m_$mode
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_m_$mode -- python bench.py --steps 3 --warmup 1 --cpu-rays-side 0 --no-other-mode --precision $mode > gpurun_out/pmc_m_$mode.json 2> gpurun_out/pmc_m_$mode.err
done
python - <<PY
import csv,glob,collections,json
out={"command":"rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -- python bench.py --steps 3 --warmup 1 --cpu-rays-side 0 --no-other-mode --precision <MODE>",
     "definitions":{"clock_ghz":"GRBM_GUI_ACTIVE / 8 XCDs / dispatch ns","mfma_busy":"SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 4 SIMDs * 256 CUs)"},"kernels":{}}
for mode in ("bf16x3","fp32"):
    f=glob.glob("gpurun_out/pmc_m_%s/**/*counter_collection.csv"%mode,recursive=True)[0]
    rows=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0]
        if "pnr::" in n and ("shade" in n or "point_part" in n):
            k=(int(r["Dispatch_Id"]), n)
            rows[k][r["Counter_Name"]]=float(r["Counter_Value"]); rows[k]["ns"]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    per=collections.defaultdict(list)
    for (d,n),v in sorted(rows.items()): per[n].append(v)
    for n,vs in per.items():
        vs=vs[-3:]   # the timed launches
        gui=[v.get("GRBM_GUI_ACTIVE",0)/8 for v in vs]
        out["kernels"][mode+" "+n]={"launches":len(vs),"ms":[round(v["ns"]/1e6,3) for v in vs],
            "clock_ghz":[round(g/v["ns"],3) for g,v in zip(gui,vs)],
            "mfma_busy":[round(v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(g*1024+1e-9),3) for g,v in zip(gui,vs)]}
json.dump(out,open("gpurun_out/pmc_mfma_busy.json","w"),indent=1)
print(json.dumps(out["kernels"],indent=1))
PY
