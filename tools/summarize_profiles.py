"""Host side of tools/collect_profiles.sh: condenses gpurun_out/prof_rNN/ into the files kept under profiles/<round>/.
Usage: python tools/summarize_profiles.py gpurun_out/prof_r04 profiles/r04 [git-sha]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
sha = sys.argv[3] if len(sys.argv) > 3 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True,
                                                           text=True).stdout.strip()
os.makedirs(dst, exist_ok=True)
MODES = ("fp32", "bf16x3")


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None  # the newest run if the directory holds several


def bench_line(path):
    with open(path) as f:
        lines = [ln for ln in f if ln.startswith("{")]
    return json.loads(lines[-1])


if os.path.exists(os.path.join(src, "bench_default.json")):
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, "bench_cfg1_n1_default.json"))

for mode in MODES:
    stats = one(f"trace_{mode}/**/*kernel_stats.csv")
    if not stats:
        continue
    shutil.copy(os.path.join(src, f"bench_under_rocprof_{mode}.json"),
                os.path.join(dst, f"bench_cfg1_n1_{mode}_under_rocprof.json"))
    # kernel stats: keep our kernels (pnr::) and the few largest others
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if "pnr::" in r["Name"]] + [r for r in rows if "pnr::" not in r["Name"]][:8]
    with open(os.path.join(dst, f"kernel_stats_bench_cfg1_{mode}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(keep)
    # per-dispatch durations of the MLP kernels (to compare with the HIP-event figure of the same run)
    trace = list(csv.DictReader(open(one(f"trace_{mode}/**/*kernel_trace.csv"))))
    disp = defaultdict(list)
    for r in trace:
        n = r["Kernel_Name"]
        if "k_shade_pairs" in n or "k_point_part" in n or "k_shade_color" in n:
            disp[n.split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    b = bench_line(os.path.join(src, f"bench_under_rocprof_{mode}.json"))
    json.dump({"command": f"rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 4 --warmup 1 "
                          f"--cpu-rays-side 0 --no-other-mode --precision {mode}",
               "collected_at": sha, "workload_key": b["config"]["workload_key"],
               "hip_event_avg_launch_ms_of_the_same_run": b["roofline"]["avg_launch_ms"],
               "timed_dispatches": 4, "dispatch_ms": dict(disp)},
              open(os.path.join(dst, f"dispatch_ms_bench_cfg1_{mode}.json"), "w"), indent=1)

# HBM traffic PMC passes
keys = {}
for mode in MODES:
    p = os.path.join(src, f"bench_pmc_FETCH_SIZE_{mode}.json")
    if os.path.exists(p):
        keys[mode] = bench_line(p)["config"]["workload_key"]
if keys:
    out = {"command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-rays-side 0 "
                      "--no-other-mode --precision <MODE> (one pass per counter and mode)",
           "collected_at": sha,
           # bench.py quotes a kernel's traffic only when its own workload_key equals the one recorded here
           "workload_key": keys.get("fp32") or next(iter(keys.values())), "workload_keys": keys,
           "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; bytes = value * 1024",
           "gfx950_correction": "FETCH_SIZE reads 1/2 of the bytes of a wide 16-B/lane read (MI355X_MICROARCH.md, HBM): "
                                "hbm_read_bytes = 2 * FETCH_SIZE * 1024 for the kernels below marked corrected",
           "kernels": {}}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = defaultdict(list)
        for i, mode in enumerate(MODES):
            f = one(f"pmc_{counter}_{mode}/**/*counter_collection.csv")
            if not f:
                continue
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                # each mode's run contributes its own MLP kernels; the mode-independent kernels come from the first run
                mlp = "k_shade" in name or "k_point_part" in name
                if r["Counter_Name"] == counter and "pnr::" in name and (mlp or i == 0):
                    acc[name].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            e = out["kernels"].setdefault(k, {})
            e[f"{counter}_KiB_avg_per_launch"] = sum(v) / len(v)
            e[f"launches_{counter}"] = len(v)
            # the timed launches are the last ones; warm-up / sizing launches render other views
            e[f"{counter}_KiB_timed_launches"] = v[-2:]
    for k, e in out["kernels"].items():
        if "FETCH_SIZE_KiB_timed_launches" in e and "WRITE_SIZE_KiB_timed_launches" in e:
            f = sum(e["FETCH_SIZE_KiB_timed_launches"]) / len(e["FETCH_SIZE_KiB_timed_launches"])
            w = sum(e["WRITE_SIZE_KiB_timed_launches"]) / len(e["WRITE_SIZE_KiB_timed_launches"])
            e["hbm_bytes_per_launch_corrected"] = (2 * f + w) * 1024
    json.dump(out, open(os.path.join(dst, "pmc_hbm_traffic.json"), "w"), indent=1)

# ---- the training step (tools/collect_profiles.sh train): per-kernel tables, wall-clock lines, the PMC table -------------
for name in ("train_kernel_times_4096.txt", "train_kernel_times_65536.txt", "train_step_4096.json", "train_step_65536.json",
             "pmc_train_65536.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))

# ---- the lego-like configuration (cfg2_lego_6m with the lego script's max_o = 830000, P = 9): its bench line ---------------
if os.path.exists(os.path.join(src, "bench_cfg2.json")):
    shutil.copy(os.path.join(src, "bench_cfg2.json"), os.path.join(dst, "bench_cfg2_n1_fp32.json"))

# ---- the K = 12 configuration (cfg4_scannet_20m, fp32): bench line, kernel stats, traffic, MFMA busy -------------------
if os.path.exists(os.path.join(src, "bench_cfg4.json")):
    shutil.copy(os.path.join(src, "bench_cfg4.json"), os.path.join(dst, "bench_cfg4_n1_fp32.json"))
stats4 = one("trace_cfg4/**/*kernel_stats.csv")
if stats4:
    rows = list(csv.DictReader(open(stats4)))
    keep = [r for r in rows if "pnr::" in r["Name"]] + [r for r in rows if "pnr::" not in r["Name"]][:4]
    with open(os.path.join(dst, "kernel_stats_bench_cfg4_fp32.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(keep)
    trace = list(csv.DictReader(open(one("trace_cfg4/**/*kernel_trace.csv"))))
    disp = defaultdict(list)
    for r in trace:
        n = r["Kernel_Name"]
        if "k_shade_pairs" in n or "k_point_part" in n or "k_shade_color" in n or "k_pair_weights" in n:
            disp[n.split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    b = bench_line(os.path.join(src, "bench_under_rocprof_cfg4.json"))
    json.dump({"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --config cfg4_scannet_20m "
                          "--steps 3 --warmup 1 --cpu-rays-side 0 --no-other-mode --precision fp32",
               "collected_at": sha, "workload_key": b["config"]["workload_key"],
               "hip_event_avg_launch_ms_of_the_same_run": b["roofline"]["avg_launch_ms"],
               "timed_dispatches": 3, "dispatch_ms": dict(disp)},
              open(os.path.join(dst, "dispatch_ms_bench_cfg4_fp32.json"), "w"), indent=1)
p4 = os.path.join(src, "bench_pmc_FETCH_SIZE_cfg4.json")
if os.path.exists(p4):
    b = bench_line(p4)
    out4 = {"command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python bench.py --config cfg4_scannet_20m --steps 2 "
                       "--warmup 1 --cpu-rays-side 0 --no-other-mode --precision fp32 (one pass per counter)",
            "collected_at": sha, "workload_key": b["config"]["workload_key"],
            "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; bytes = value * 1024",
            "gfx950_correction": "hbm_read_bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section)",
            "counters_per_launch": b["counters_per_launch"], "kernels": {}}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        f = one(f"pmc_{counter}_cfg4/**/*counter_collection.csv")
        if not f:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if r["Counter_Name"] == counter and "pnr::" in name:
                acc[name].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            e = out4["kernels"].setdefault(k, {})
            e[f"{counter}_KiB_timed_launches"] = v[-2:]
    for k, e in out4["kernels"].items():
        if "FETCH_SIZE_KiB_timed_launches" in e and "WRITE_SIZE_KiB_timed_launches" in e:
            f_ = sum(e["FETCH_SIZE_KiB_timed_launches"]) / len(e["FETCH_SIZE_KiB_timed_launches"])
            w_ = sum(e["WRITE_SIZE_KiB_timed_launches"]) / len(e["WRITE_SIZE_KiB_timed_launches"])
            e["hbm_bytes_per_launch_corrected"] = (2 * f_ + w_) * 1024
    c = b["counters_per_launch"]
    out4["survey_8d_bytes_per_launch"] = {"M_x_164": c["pairs_valid"] * 164,
                                          "formula": "pairs_valid * 164 B (xyz 12 + emb 128 + color 12 + dir 12)"}
    json.dump(out4, open(os.path.join(dst, "pmc_hbm_traffic_cfg4.json"), "w"), indent=1)

# MFMA busy / effective clock
busy = {"command": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -- python bench.py --steps 3 "
                   "--warmup 1 --cpu-rays-side 0 --no-other-mode --precision <MODE>", "collected_at": sha,
        "definitions": {"clock_ghz": "GRBM_GUI_ACTIVE / 8 XCDs / dispatch ns",
                        "mfma_busy": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 4 SIMDs * 256 CUs)"}, "kernels": {}}
for mode in MODES + ("cfg4",):
    f = one(f"pmc_mfma_{mode}/**/*counter_collection.csv")
    if not f:
        continue
    rows = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if "pnr::" in n and ("shade" in n or "point_part" in n):
            k = (int(r["Dispatch_Id"]), n)
            rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[k]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per = defaultdict(list)
    for (d, n), v in sorted(rows.items()):
        per[n].append(v)
    for n, vs in per.items():
        vs = vs[-(2 if mode == "cfg4" else 3):]   # the timed launches
        gui = [v.get("GRBM_GUI_ACTIVE", 0) / 8 for v in vs]
        busy["kernels"][("fp32 cfg4_scannet_20m" if mode == "cfg4" else mode) + " " + n] = {
            "launches": len(vs), "ms": [round(v["ns"] / 1e6, 3) for v in vs],
            "clock_ghz": [round(g / v["ns"], 3) for g, v in zip(gui, vs)],
            "mfma_busy": [round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 1024 + 1e-9), 3) for g, v in zip(gui, vs)]}
if busy["kernels"]:
    json.dump(busy, open(os.path.join(dst, "pmc_mfma_busy.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
