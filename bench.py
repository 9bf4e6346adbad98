#!/usr/bin/env python3
"""bench.py -- rays/s of the full-image Point-NeRF render hot path on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): synthetic chair-bbox
neural point cloud of ~6 M points, 800x800 image, D = 400 coarse samples, SR = 80 shading samples per ray,
K = 8 neighbours, fp32 arithmetic, the reference's coarse-sample jitter of 0.3 (studio_utils.py:166; seeded:
seed 7) in the timed legs and jitter 0 in the parity leg; MLP weights Xavier-initialised (seed 0), density head
scaled so that opacities are non-trivial.  Datasets / checkpoints are not reachable: data is synthetic, seeded.

One step = N views (N = number of GPUs).  Every view is cut into 16x16-pixel tiles dealt round-robin to the
N ranks; each rank renders its tiles of all N views in ONE multi-camera call (N * 640000 / N = 640000 rays per
rank per step: weak scaling) and ONE all_gather per step (RCCL over xGMI) puts the N full RGB+depth images on
every rank.
Inputs (ray directions, point tensors, weights, voxel structure) are resident in HBM before the timed
region; the timed region covers query + gather + MLPs + composite + all_gather for K steps.

  python bench.py --gpus 1 --steps 8 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` is measured in the reference's arithmetic: --precision fp32 (default; every
product and sum in fp32 on v_mfma_f32_32x32x2_f32), `roofline` prices the dominant kernel (the MLP chain
k_shade_pairs) against the dense fp32 MFMA peak of 157.3 TFLOP/s.  `other_mode` holds the same workload, same
steps and warm-up, in the opt-in bf16x3 mode (every fp32 product as 3 bf16 MFMA products on hi/lo splits: narrower
than fp32, never `value`).  `cpu_baseline` times the CPU oracle (oracle/pnr_oracle.py, the PyTorch-CPU restatement
of the reference path) on a bounded sample of the same workload on this box's host cores: median of 5 passes with
the voxel grid built once, and one pass "as written" (grid rebuilt per 2304-ray chunk, studio_config.py:25).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# Runtime environment defaults, set before torch (and with it the HIP / HSA runtime) is even imported: a value set after
# the first HIP call is never read.  dmabuf IPC is the only form the pool's host driver supports (RCCL and tensor
# sharing across processes fail with `hipIpcGetMemHandle: invalid argument` under the legacy mode).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointnerf2studio_amd import _lib, synthetic  # noqa: E402
from pointnerf2studio_amd.distributed import ViewGatherPipe, make_shard  # noqa: E402
from pointnerf2studio_amd.renderer import (RendererHIP, SceneHIP, View, WeightsHIP, grid_hyperparameters)  # noqa: E402

FLOPS_PER_PAIR = 542_720       # 2 * (284*256 + 256*256 + 263*256 + 256*256 + 256)   SURVEY.md section 8d
# mlp_base layer 0 is factorised (both modes): the pair kernels multiply the 60 pair inputs only, the 224
# point-only inputs are contracted once per distinct neighbour point by k_point_part(_f32)
FLOPS_PER_PAIR_KERNEL = 428_032   # 2 * (60*256 + 256*256 + 263*256 + 256*256 + 256): what the pair kernels do
FLOPS_PER_POINT_PART = 114_688         # 2 * 224*256
MFMA_FLOPS_PER_PAIR_BF16 = 1_302_528   # executed: 1272 x v_mfma_f32_32x32x16_bf16 (32768 FLOP) per 32 pairs
FLOPS_PER_SAMPLE = 137_984     # 2 * (280*128 + 2*128*128 + 128*3)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak (no 2:1 sparsity)
PEAK_HBM_GBS = 8000.0           # MI355X HBM3E, ~8 TB/s (MI355X_MICROARCH.md)

VSCALE = [2, 2, 2]
KSIZE = [3, 3, 3]


REF_CHUNK = 2304   # eval_num_rays_per_chunk of the reference (studio_config.py:25)


def host_cpu():
    """(model name, physical cores, logical CPUs) of this box from /proc/cpuinfo: physical = distinct (package, core id)."""
    model, cores, logical = "", set(), 0
    try:
        with open("/proc/cpuinfo") as f:
            pkg = None
            for ln in f:
                key, _, val = ln.partition(":")
                key, val = key.strip(), val.strip()
                if key == "processor":
                    logical += 1
                elif key == "model name" and not model:
                    model = val
                elif key == "physical id":
                    pkg = val
                elif key == "core id":
                    cores.add((pkg, val))
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    return model, (len(cores) or logical), logical


def cpu_quota():
    """CPUs the cgroup of this process may use at once (cpu.max / cfs quota), None when unlimited: a GPU box shows a test
    all 256 logical CPUs of its host and schedules 16 of them."""
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t[0] == "max" else float(t[0]) / float(t[1])),):
        try:
            with open(path) as f:
                return parse(f.read().split())
        except (OSError, ValueError, IndexError):
            pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            q, per = float(f.read()), float(g.read())
            return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_threads(requested: int) -> int:
    """Threads of the CPU-baseline legs: --cpu-threads, default (0) one per PHYSICAL core (BASELINE.md section 2 /
    SURVEY.md section 8d: "all physical cores"; the GEMM-bound oracle gains nothing from the SMT siblings), capped by the
    CPUs this process may actually use: its affinity mask and its cgroup's CPU quota.  (Measured on a pool box -- 2 x 64
    cores, quota 16 CPUs -- for the 64 x 64 window: 256 threads 32 s, 128 threads 4.5 s, 64 threads 3.1 s, 32 threads 2.8 s,
    16 threads 3.0 s: oversubscribing the quota is what made the earlier rounds' baseline slow.)"""
    _, physical, logical = host_cpu()
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = logical
    quota = cpu_quota()
    if quota is not None:
        allowed = min(allowed, max(1, int(quota)))
    return max(1, min(requested or physical, allowed))


def cpu_baseline(points, weights, cfgd, n_side, view, passes=5, budget_s=270.0, threads=0):
    """Times the CPU oracle (a port of the reference's PyTorch path) on an n_side x n_side centre window of the
    workload, as BASELINE.md section 2 specifies: wall clock, median of `passes` after one warm-up, in two flavours --
    (ii) the voxel grid built once for the sample (`value`: the stricter baseline) and (i) "as written": the sample
    rendered in chunks of 2304 rays, the grid rebuilt for every chunk as the reference does
    (query_worldcoords.cu:314-365), one pass."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pnr_oracle as O
    O.build_c_oracle()
    cfg = O.OracleConfig()
    cfg.SR, cfg.K, cfg.P, cfg.max_o = cfgd["SR"], cfgd["K"], cfgd["P"], cfgd["max_o"]
    cfg.ranges = list(cfgd["ranges"])
    cfg.vsize = [cfgd["vsize"]] * 3
    H, W = cfgd["H"], cfgd["W"]
    near, far = cfgd["near"], cfgd["far"]
    campos, camrot = synthetic.make_scene_camera(cfgd, view)
    y0, x0 = (H - n_side) // 2, (W - n_side) // 2
    dirs = synthetic.make_rays(H, W, campos, camrot, cfgd["angle_x"], y0=y0, y1=y0 + n_side, x0=x0, x1=x0 + n_side)
    n = dirs.shape[0]
    torch.set_num_threads(cpu_threads(threads))

    def one(d):
        return O.render(points, weights, cfg, campos[None].expand(d.shape[0], 3), d, near, far, camrot)
    one(dirs[:64].contiguous())    # warm-up on a sliver (thread pools, oneDNN primitives)
    t_leg = time.time()
    times, ref, dt_written = [], None, None
    for i in range(passes):
        # time-boxed: the default bench run must finish within minutes on whatever host the box has; every pass after
        # the first is taken only while the leg's budget allows one more
        if times and time.time() - t_leg + max(times) > budget_s:
            break
        t0 = time.time()
        ref = one(dirs)
        times.append(time.time() - t0)
        if i == 0:
            t0 = time.time()
            for c0 in range(0, n, REF_CHUNK):
                one(dirs[c0:c0 + REF_CHUNK].contiguous())
            dt_written = time.time() - t0
    dt = sorted(times)[len(times) // 2]
    cpu_model, physical, logical = host_cpu()
    return dict(value=n / dt, unit="rays/s", cores=torch.get_num_threads(), threads=torch.get_num_threads(),
                physical_cores=physical, logical_cpus=logical, cpu_quota=cpu_quota(), kind="port",
                sample=f"{n_side}x{n_side} centre window of view {view} ({n} rays) against the full "
                       f"{points['xyz'].shape[0]}-point cloud, jitter 0, voxel grid built once for the sample; median of "
                       f"{len(times)} passes after a warm-up ({', '.join(f'{t:.1f}' for t in times)} s)",
                seconds=dt, passes=times, cpu_model=cpu_model,
                as_written={"value": n / dt_written, "unit": "rays/s", "seconds": dt_written,
                            "note": f"the same {n} rays in chunks of {REF_CHUNK}, the voxel grid rebuilt for every chunk "
                                    f"as the reference does per eval chunk; one pass"},
                counts={"R": n, "R_hit": ref["stats"]["rays_hit"], "S": ref["stats"]["valid_samples"],
                        "M": ref["stats"]["valid_pairs"]}), ref, dirs, campos, camrot


def pmc_traffic(kernel, workload_key):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC pass (bench.py cannot collect PMC
    counters itself: they need rocprofv3 --pmc passes, tools/pmc_hbm.sh).  Returned only when that pass was collected
    on THIS workload (config, N, K, SR, precision, jitter, world size); otherwise None."""
    import glob
    for rnd in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", rnd, "pmc_hbm_traffic*.json"))):
            try:
                with open(path) as f:
                    d = json.load(f)
                if d.get("workload_key") != workload_key:
                    continue
                k = next((v for n, v in d["kernels"].items() if n.split("<")[0].endswith("::" + kernel)), None)
                if k is not None:
                    return (k["hbm_bytes_per_launch_corrected"], os.path.relpath(path, ROOT), d.get("collected_at", ""))
            except (OSError, KeyError, ValueError):
                continue
    return None


QUERY_STAGE_KERNELS = ("k_select", "k_expand", "k_knn3", "k_knn3_coop", "k_knn", "k_compact_valid", "k_list_points",
                       "k_pair_weights")


def pmc_traffic_sum(kernels, workload_key):
    """Sum of hbm_bytes_per_launch_corrected over the named kernels (every template instance of each) in the newest
    committed PMC pass collected on THIS workload; (bytes, {kernel: bytes}, path, collected_at) or None."""
    import glob
    for rnd in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", rnd, "pmc_hbm_traffic*.json"))):
            try:
                with open(path) as f:
                    d = json.load(f)
                if d.get("workload_key") != workload_key:
                    continue
                per = {n: v["hbm_bytes_per_launch_corrected"] for n, v in d["kernels"].items()
                       if n.split("<")[0].split("::")[-1] in kernels}
                if per:
                    return sum(per.values()), per, os.path.relpath(path, ROOT), d.get("collected_at", "")
            except (OSError, KeyError, ValueError):
                continue
    return None


def cfg0_leg(args, dev):
    """BASELINE cfg[0] -- 50 k-point chair cloud, 64 x 64 image, SR 32, K 8: "the reference's CPU-runnable case" -- rendered
    WHOLE by both sides: the CPU oracle on the host cores (median of --cpu-passes after a warm-up; grid built once per
    frame) and the HIP path through the C ABI on the same rays at jitter 0, with the parity of the two images."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pnr_oracle as O
    O.build_c_oracle()
    c = dict(synthetic.SCENE_CONFIGS["cfg0_chair_50k"])
    H, W, SR, K = c["H"], c["W"], c["SR"], c["K"]
    points = synthetic.make_scene_points(c, seed=1234)
    weights = synthetic.make_weights(0, sigma_scale=args.sigma_scale, bias_scale=0.1)
    cfg = O.OracleConfig()
    cfg.SR, cfg.K, cfg.P, cfg.max_o, cfg.ranges, cfg.vsize = SR, K, c["P"], c["max_o"], list(c["ranges"]), [c["vsize"]] * 3
    campos, camrot = synthetic.make_scene_camera(c, 0)
    dirs = synthetic.make_rays(H, W, campos, camrot, c["angle_x"])
    n = dirs.shape[0]
    torch.set_num_threads(cpu_threads(args.cpu_threads))

    def one():
        return O.render(points, weights, cfg, campos[None].expand(n, 3), dirs, c["near"], c["far"], camrot)
    one()
    times = []
    for _ in range(max(args.cpu_passes, 1)):
        t0 = time.time()
        ref = one()
        times.append(time.time() - t0)
    dt = sorted(times)[len(times) // 2]
    # the same frame on the GPU
    xyz = points["xyz"].to(dev)
    vs = [c["vsize"]] * 3
    hyp = grid_hyperparameters(xyz, vs, VSCALE, KSIZE, c["ranges"])
    scene = SceneHIP()
    scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, KSIZE, KSIZE, c["P"], c["max_o"], True)
    scene.pack_points(xyz, points["embedding"].to(dev), points["conf"].to(dev), points["dir"].to(dev), points["color"].to(dev))
    wh = WeightsHIP()
    wh.pack(weights, points["Rw2c"], dev)
    rnd = RendererHIP(scene, wh, SR=SR, K=K, D=400, radius_limit=4 * c["vsize"], vsize_z=c["vsize"], precision=args.precision)
    d_dev = dirs.to(dev)
    out = rnd.render(d_dev, campos, camrot, c["near"], c["far"])
    cap = int(out["counters"]["samples_selected"] * 1.05) + 4096
    for _ in range(5):
        rnd.render(d_dev, campos, camrot, c["near"], c["far"], cap_samples=cap, sync_counters=False, out=out)
    torch.cuda.synchronize()
    iters = 50
    t0 = time.perf_counter()
    for _ in range(iters):
        rnd.render(d_dev, campos, camrot, c["near"], c["far"], cap_samples=cap, sync_counters=False, out=out)
    torch.cuda.synchronize()
    g_dt = (time.perf_counter() - t0) / iters
    diff = out["rgb"].cpu() - ref["coarse_raycolor"]
    cpu_model, physical, logical = host_cpu()
    return {
        "workload": f"cfg0_chair_50k: N={c['N']}, {H}x{W}, D=400, SR={SR}, K={K}, P={c['P']}, jitter 0, view 0, the WHOLE frame "
                    f"({n} rays) on both sides",
        "cpu": {"value": n / dt, "unit": "rays/s", "kind": "port", "seconds": dt, "passes": times,
                "threads": torch.get_num_threads(), "physical_cores": physical, "logical_cpus": logical,
                "cpu_quota": cpu_quota(), "cpu_model": cpu_model},
        "gpu": {"value": n / g_dt, "unit": "rays/s", "ms_per_frame": g_dt * 1e3, "mode": args.precision,
                "note": "one pnr_render call per frame, 50 frames back to back; not a throughput figure: 0.23 ms of a 4096-ray "
                        "frame are two rounds of the pair kernel (458 tiles of 128 rows on 256 CUs; a round is one 32-row "
                        "tile through the four layers, ~0.115 ms whatever the batch), the rest about twenty small launches"},
        "parity": {"max_abs_rgb_err": diff.abs().max().item(),
                   "ray_mask_equal": bool(torch.equal(out["ray_mask"].cpu(), ref["ray_mask"])),
                   "psnr_vs_oracle_db": float(-10 * torch.log10((diff ** 2).mean() + 1e-20)),
                   "rays_kept": int(ref["stats"]["rays_kept"]), "valid_pairs": int(ref["stats"]["valid_pairs"])},
        "speedup": (n / g_dt) / (n / dt),
    }


def plugin_legs(args, cfgd, points, weights, cams, dev, near, far):
    """rays/s through the plugin mirror's surface on the SAME workload as `value` (cloud, views, SR, K, jitter 0.3, planes
    near / far handed over in the bundle as the datamanager's config states them, studio_datamanager.py:40-41): camera
    ray bundles as studio_datamanager.py:104-110 builds them ([H, W, .] tensors resident in HBM, the rotation expanded
    per pixel), one get_outputs_for_camera_ray_bundle call per frame.  Then the training step at the batch `ns-train`
    draws (4096 random pixels of one image, studio_config.py:20-21)."""
    from pointnerf2studio_amd.model import PointNerf, PointNerfConfig
    from pointnerf2studio_amd.ns_compat import RayBundle
    H, W = cfgd["H"], cfgd["W"]
    sd = {"neural_points.xyz": points["xyz"], "neural_points.points_embeding": points["embedding"],
          "neural_points.points_conf": points["conf"], "neural_points.points_dir": points["dir"],
          "neural_points.points_color": points["color"], "neural_points.Rw2c": points["Rw2c"]}
    # as `ns-train pointnerf-original` configures it: nerfstudio's collider writes the planes (its near plane kept in eval
    # too, so that the frames are the ones `value` renders), bundles are one camera each (studio_datamanager.py:62-110)
    cfg = PointNerfConfig(ranges=list(cfgd["ranges"]), max_o=cfgd["max_o"], SR=cfgd["SR"], K=cfgd["K"], P=cfgd["P"],
                          vsize=[cfgd["vsize"]] * 3, hip_mlp_mode=args.precision, enable_collider=True,
                          collider_params={"near_plane": float(near), "far_plane": float(far)},
                          eval_num_rays_per_chunk=REF_CHUNK, hip_single_camera_bundles=True)
    model = PointNerf(cfg, point_state_dict=sd).to(dev)
    model.load_state_dict(weights, strict=False)
    model.neural_points.jitter = float(args.jitter)
    if hasattr(model.collider, "reset_near_plane"):
        model.collider.reset_near_plane = False
    bundles = []
    for campos, camrot in cams:
        d = synthetic.make_rays(H, W, campos, camrot, cfgd["angle_x"]).to(dev).reshape(H, W, 3)
        bundles.append(RayBundle(
            origins=campos.to(dev)[None, None].expand(H, W, 3).contiguous(), directions=d,
            metadata={"camrotc2w": camrot.to(dev)[None, None].expand(H, W, -1, -1).reshape(H, W, -1)}))
    model.eval()
    for s in range(max(args.warmup, 1) + len(bundles)):      # warm-up: every view once (capacity, camera memo)
        model.get_outputs_for_camera_ray_bundle(bundles[s % len(bundles)])
    torch.cuda.synchronize()
    reads0, calls0 = model.host_reads, model._render_calls
    t0 = time.perf_counter()
    for s in range(args.steps):
        model.get_outputs_for_camera_ray_bundle(bundles[(args.warmup + s) % len(bundles)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev = {"value": H * W * args.steps / dt, "unit": "rays/s", "steps": args.steps, "ms_per_frame": dt / args.steps * 1e3,
          "fused_calls_per_frame": (model._render_calls - calls0) / args.steps,
          "host_reads_per_frame": (model.host_reads - reads0) / args.steps, "mode": args.precision,
          "path": "PointNerf.get_outputs_for_camera_ray_bundle([H, W] RayBundle) -> one pnr_render_views call per frame "
                  "(nerfstudio's inherited loop: 278 calls of 2304 rays, studio_config.py:25)"}
    # the inherited chunk loop on the same frames, for the record (3 frames)
    model.config.hip_eval_one_call = False
    model.get_outputs_for_camera_ray_bundle(bundles[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(3):
        model.get_outputs_for_camera_ray_bundle(bundles[s % len(bundles)])
    torch.cuda.synchronize()
    ev["chunk_loop_of_2304_rays"] = {"value": H * W * 3 / (time.perf_counter() - t0), "unit": "rays/s", "frames": 3}
    model.config.hip_eval_one_call = True

    # ---- training step: 4096 random pixels of view 0, a NEW bundle object per step as the datamanager hands them over
    model.train()
    torch.manual_seed(12)
    full = bundles[0].directions.reshape(-1, 3)
    campos0_dev, camrot0_dev = cams[0][0].to(dev), cams[0][1].to(dev)
    from pointnerf2studio_amd.optim import PointRowAdam
    groups = model.get_param_groups()
    # the optimisers as studio_config registers them (reference studio_config.py:33-48: Adam 5e-4 / 2e-3): "fields" torch
    # Adam, "neural_points" the same Adam over the rows that ever had a gradient (optim.PointRowAdam); and, for the record,
    # torch.optim.Adam for both groups -- what nerfstudio would construct from the reference's config, a dense sweep
    opts = {"registered": [torch.optim.Adam(groups["fields"], lr=5e-4, eps=1e-8),
                           PointRowAdam(groups["neural_points"], lr=2e-3, eps=1e-8)],
            "torch_dense": [torch.optim.Adam(groups["fields"], lr=5e-4, eps=1e-8),
                            torch.optim.Adam(groups["neural_points"], lr=2e-3, eps=1e-8)]}
    callbacks = model.get_training_callbacks(None)
    n_rays = 4096

    def one_step(which):
        pick = torch.randint(0, full.shape[0], (n_rays,), device=dev)      # (drawn on the device: no host work)
        b = RayBundle(origins=campos0_dev[None].expand(n_rays, 3), directions=full.index_select(0, pick),
                      metadata={"camrotc2w": camrot0_dev})
        batch = {"image": torch.rand((n_rays, 3), device=dev)}
        model.zero_grad(set_to_none=True)
        out = model(b)
        loss = sum(model.get_loss_dict(out, batch).values())
        loss.backward()
        for o in opts.get(which, ()):
            o.step()
        for cb in callbacks:
            cb.run_callback(step=0)

    tr = {"rays": n_rays, "mode": args.precision}
    for name, which, iters in (("plugin_step_ms", None, 20), ("plugin_step_with_adam_ms", "registered", 40),
                               ("plugin_step_with_torch_dense_adam_ms", "torch_dense", 20)):
        for _ in range(5):
            one_step(which)
        torch.cuda.synchronize()
        reads0 = model.host_reads
        t0 = time.perf_counter()
        for _ in range(iters):
            one_step(which)
        torch.cuda.synchronize()
        tr[name] = (time.perf_counter() - t0) / iters * 1e3
        tr["host_reads_per_step"] = (model.host_reads - reads0) / iters
    tr["rays_per_sec"] = n_rays / (tr["plugin_step_with_adam_ms"] * 1e-3)
    tr["point_rows_ever_touched"] = opts["registered"][1].ever_touched()
    tr["note"] = ("wall clock per step, host running ahead of the device: PointNerf.forward (fused render, rows of the "
                  "touched points refreshed from the bound parameters) + get_loss_dict + backward (fused; point gradients "
                  "accumulated into persistent dense buffers) + the after-step callback.  _with_adam: + the optimisers "
                  "studio_config registers (torch Adam for the nine Linear layers, optim.PointRowAdam -- torch's Adam update "
                  "over the rows that ever had a gradient -- for the point tensors); _with_torch_dense_adam: torch.optim.Adam "
                  "for both groups (a dense sweep over 6 M x 38 point values per step).  rays_per_sec: the step with Adam")
    del model
    return {"eval": ev, "train": tr}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg1_chair_6m", choices=sorted(synthetic.SCENE_CONFIGS))
    ap.add_argument("--points", type=int, default=None, help="override the number of points")
    ap.add_argument("--points-per-voxel", type=int, default=None, help="override P, the points a voxel list keeps (the "
                    "reference's scripts: 12 / 9 / 26): a larger P puts more of the cloud into the neighbour search -- a "
                    "stress of the gather, not a BASELINE configuration")
    ap.add_argument("--cpu-rays-side", type=int, default=64, help="side of the CPU-baseline window (0 = skip)")
    ap.add_argument("--cpu-passes", type=int, default=5, help="timed passes of the CPU baseline (median is reported)")
    ap.add_argument("--cpu-budget-s", type=float, default=270.0, help="wall-clock budget of the CPU-baseline leg: no "
                    "further pass is started once one more would exceed it")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU-baseline legs (0 = one per physical "
                    "core of the box)")
    ap.add_argument("--no-cfg0", action="store_true", help="skip the BASELINE cfg[0] leg (50 k points, 64x64: the whole "
                    "frame on the CPU oracle and on the GPU)")
    ap.add_argument("--sigma-scale", type=float, default=300.0)
    ap.add_argument("--no-other-mode", action="store_true", help="skip the side legs (other arithmetic mode, early "
                    "termination, training step)")
    ap.add_argument("--jitter", type=float, default=0.3, help="coarse-sample jitter of the timed legs (the reference "
                    "renders with 0.3, studio_utils.py:166); the parity leg always runs at 0")
    ap.add_argument("--jitter-seed", type=int, default=7)
    ap.add_argument("--rays-from-tensor", action="store_true", help="render from a resident [R,3] direction tensor "
                    "(pnr_render_views) instead of generating the rays in the kernels (pnr_render_camera)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic, single process: shard as rank --emulate-rank of this many ranks, no collectives")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="MLP arithmetic of `value`: fp32 MFMA (the reference's arithmetic), or the opt-in 3 bf16 MFMAs "
                         "per fp32 product (hi/lo split)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path is HIP-only (no CPU fallback)")
    # PNR_BENCH_REHEARSE=1 (diagnostic): all ranks share cuda:0 and talk over gloo -- a rehearsal of the N > 1 code
    # path on a one-GPU box (RCCL refuses two ranks on one device); the numbers it prints mean nothing
    rehearse = os.environ.get("PNR_BENCH_REHEARSE") == "1" and world > 1
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_world = None
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # what the collective library itself says about the job (the line's n_gpus is the launcher's WORLD_SIZE): an
        # all_reduce of ones counts the ranks that took part, an all_gather collects every rank's device
        ones = torch.ones(1, dtype=torch.int32, device="cpu" if rehearse else dev)
        dist.all_reduce(ones)
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device": f"cuda:{local_rank}", "name": props.name,
                "gcn_arch": getattr(props, "gcnArchName", ""), "pci_bus_id": getattr(props, "pci_bus_id", None),
                "uuid": str(getattr(props, "uuid", ""))}
        every = [None] * world
        dist.all_gather_object(every, mine)
        rccl_world = {"all_reduce_of_ones": int(ones.item()), "backend": dist.get_backend(),
                      "equals_n_gpus": int(ones.item()) == world, "ranks": every,
                      "distinct_devices": len({(e["uuid"], e["pci_bus_id"], e["device"]) for e in every})}
        if int(ones.item()) != world:
            raise SystemExit(f"the collective saw {int(ones.item())} ranks, WORLD_SIZE says {world}")
    emulate = args.emulate_world > 1 and world == 1
    if emulate:   # per-rank work of an N-rank run, without the collectives (local copy instead of all_gather)
        world, rank = args.emulate_world, args.emulate_rank

    cfgd = dict(synthetic.SCENE_CONFIGS[args.config])
    if args.points:
        cfgd["N"] = args.points
    if args.points_per_voxel:
        cfgd["P"] = args.points_per_voxel
    H, W, SR, K = cfgd["H"], cfgd["W"], cfgd["SR"], cfgd["K"]

    # ---- scene resident in HBM (replicated on every rank) ----------------------------------------------
    VSIZE = [cfgd["vsize"]] * 3
    near, far = cfgd["near"], cfgd["far"]
    points = synthetic.make_scene_points(cfgd, seed=1234)
    weights = synthetic.make_weights(0, sigma_scale=args.sigma_scale, bias_scale=0.1)
    xyz = points["xyz"].to(dev)
    hyp = grid_hyperparameters(xyz, VSIZE, VSCALE, KSIZE, cfgd["ranges"])
    scene = SceneHIP()
    t0 = time.time()
    info = scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, KSIZE, KSIZE, cfgd["P"], cfgd["max_o"], True)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    scene.pack_points(xyz, points["embedding"].to(dev), points["conf"].to(dev), points["dir"].to(dev),
                      points["color"].to(dev))
    wh = WeightsHIP()
    wh.pack(weights, points["Rw2c"], dev)
    rnd = RendererHIP(scene, wh, SR=SR, K=K, D=400, radius_limit=4 * max(VSIZE[0], VSIZE[1]), vsize_z=VSIZE[2],
                      precision=args.precision, jitter=args.jitter, seed=args.jitter_seed)
    workload_key = (f"{args.config}:N={cfgd['N']}:K={K}:SR={SR}:{args.precision}:jitter={args.jitter:g}:"
                    f"world={world}" + (f":P={cfgd['P']}" if args.points_per_voxel else ""))

    # ---- rays: `world` views per step, this rank's tiles of each ----------------------------------------
    azimuths = list(range(8))   # the eight views of the configuration
    # (rotate: the i-th view of a step is rendered with the tiles of owner (rank + i) % world -- see TileShard)
    shard = make_shard(H, W, world, rank, rotate=True).to(dev)
    # A view is pose + intrinsics; its rays are generated inside the kernels from the pixel ids this rank owns
    # (pnr_render_camera): no direction tensor exists.  --rays-from-tensor renders from a resident [R,3] tensor instead
    # (pnr_render_views, the plugin's ray-bundle contract).
    shard_px = (shard.view_pixels if shard.rotate else shard.pixels).to(torch.int32).contiguous()   # [world, n_pad] / [n_pad]
    view_dirs, cams, views = [], [], []
    for az in azimuths:
        campos, camrot = synthetic.make_scene_camera(cfgd, az)
        cams.append((campos, camrot))
        views.append(View.from_angle(campos, camrot, H, W, cfgd["angle_x"], near, far))
        if args.rays_from_tensor:
            view_dirs.append(synthetic.make_rays(H, W, campos, camrot, cfgd["angle_x"]).to(dev))   # whole frame
    n_local = shard.n_pad * world          # this rank's rays of one step: its tiles of `world` views
    # the view sets a step can consist of: views (s*world + i) % 8, i < world -- concatenated once, resident in HBM
    step_sets = {}
    for s0 in range(len(azimuths)):
        vs = tuple((s0 * world + i) % len(azimuths) for i in range(world))
        if vs not in step_sets:
            step_sets[vs] = (torch.cat([view_dirs[v].index_select(0, shard.pixels_of_view(i)) for i, v in enumerate(vs)]).contiguous()
                             if args.rays_from_tensor else None,
                             [(cams[v][0], cams[v][1], near, far) for v in vs], [views[v] for v in vs])

    def render_step(renderer, vs, **kw):
        dirs_s, cams_s, views_s = step_sets[vs]
        if args.rays_from_tensor:
            return renderer.render_views(dirs_s, cams_s, shard.n_pad, **kw)
        return renderer.render_camera(views_s, H, W, pixels=shard_px, **kw)

    outs = {
        "rgb": torch.empty((n_local, 3), dtype=torch.float32, device=dev),
        "depth": torch.empty((n_local,), dtype=torch.float32, device=dev),
        "acc": torch.empty((n_local,), dtype=torch.float32, device=dev),
        "ray_mask": torch.empty((n_local,), dtype=torch.int8, device=dev),
        "counters_dev": torch.zeros(_lib.NUM_COUNTERS, dtype=torch.int64, device=dev),
    }
    # the step's exchange: ONE all_gather of the rank's RGB+depth tiles of all views, issued asynchronously so that it
    # overlaps the next step's render (images of step s are complete when step s + 1 is submitted / at drain())
    pipe = ViewGatherPipe(shard, world, 4, torch.float32, dev, local_copy=emulate)

    # capacity: size the workspace once from the heaviest step (untimed)
    cap = 0
    for vs in step_sets:
        o = render_step(rnd, vs, out=outs)
        cap = max(cap, o["counters"]["samples_selected"])
    cap = int(cap * 1.05) + 4096
    render_step(rnd, next(iter(step_sets)), cap_samples=cap, out=outs)

    lib = _lib.load()
    n_calls_max = max(args.steps, 4)  # the side legs (other mode, early termination) time up to 3 steps
    counters_all = torch.zeros((n_calls_max, _lib.NUM_COUNTERS), dtype=torch.int64, device=dev)

    def run_steps(renderer, first, count, counters=None):
        """`count` steps; NOTHING in here waits for the device (stage times and counters are read afterwards).
        One step = ONE render call over this rank's tiles of all `world` views + ONE all_gather."""
        call = 0
        for s in range(first, first + count):
            vs = tuple((s * world + i) % len(azimuths) for i in range(world))
            if counters is not None:
                outs["counters_dev"] = counters[call]
            render_step(renderer, vs, cap_samples=cap, sync_counters=False, out=outs)
            local4 = pipe.stage()
            local4[:, :3].copy_(outs["rgb"])
            local4[:, 3].copy_(outs["depth"])
            pipe.submit()
            call += 1
        pipe.drain()   # the last step's images: inside the timed region
        return call

    def timed(renderer, steps, first):
        """barrier + synchronize, `steps` steps, barrier + synchronize; max over ranks; per-stage device times of
        every launch of the region from the HIP events pnr_render recorded on its stream."""
        _lib.check(lib.pnr_profile_enable(1), "pnr_profile_enable")
        if world > 1 and not emulate:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = run_steps(renderer, first, steps, counters_all)
        if world > 1 and not emulate:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1 and not emulate:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = (C.c_float * _lib.NUM_STAGES)()
        acc = [0.0] * _lib.NUM_STAGES
        recorded = int(lib.pnr_profile_calls())
        lo = max(0, recorded - 256)
        for c in range(lo, recorded):
            _lib.check(lib.pnr_profile_read(c, C.byref(ms)), "pnr_profile_read")
            for k in range(_lib.NUM_STAGES):
                acc[k] += ms[k]
        lib.pnr_profile_enable(0)
        cnt = counters_all[lo:n].sum(0).tolist()
        return float(t.item()), acc, cnt, max(recorded - lo, 1)

    def roofline(mode, acc_ms, cnt, n_launch):
        """Dominant kernel = the per-pair MLP chain.  achieved = valid pairs x the fp32 FLOPs THIS kernel computes per
        pair / its device time (HIP events on the render stream, pnr_profile_*); invalid neighbour slots and tile
        padding are computed but not counted."""
        pairs, samples, upoints = cnt[4], cnt[3], cnt[7]
        t_pairs = acc_ms[2] / 1e3
        bf = mode == "bf16x3"
        per_pair = FLOPS_PER_PAIR_KERNEL   # both modes run the factorised first layer
        achieved = pairs * per_pair / t_pairs / 1e12 if t_pairs > 0 else 0.0
        peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_F32_MFMA_TFLOPS
        # (fp32, K that does not fill a DPP segment -- cfg[4]'s K = 12 --: the pair kernel on dense units)
        kernel = "k_shade_pairs_bf16" if bf else ("k_shade_pairs_dense" if K >= 11 and K != 16 else "k_shade_pairs")
        executed = pairs * MFMA_FLOPS_PER_PAIR_BF16 / t_pairs / 1e12 if (bf and t_pairs > 0) else achieved
        alg_bytes = pairs * 8 + upoints * 1072 + samples * 1064
        key = workload_key.replace(f":{args.precision}:", f":{mode}:")
        traffic = pmc_traffic(kernel, key)
        t_both = (acc_ms[2] + acc_ms[5]) / 1e3
        return {
            "bound": "mfma", "kernel": kernel, "mode": mode,
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "peak_note": ("dense bf16 MFMA peak.  achieved = algorithmic fp32 FLOPs of THIS kernel (mlp_base layer 0 is "
                          "factorised: its 224 point-only inputs are contracted once per distinct neighbour point by "
                          "k_point_part, stage point_part) / its launch time; the kernel executes 3 bf16 MFMA products "
                          "per algorithmic product on padded tiles: executed_mfma_frac prices those"
                          if bf else "dense fp32 MFMA peak; achieved = algorithmic fp32 FLOPs of THIS kernel (mlp_base "
                          "layer 0 is factorised: its 224 point-only inputs are contracted once per distinct neighbour "
                          "point by k_point_part_f32, stage point_part) / its launch time"),
            "executed_mfma_frac": executed / peak,
            "traffic": traffic[0] if traffic else None,
            "traffic_source": (f"{traffic[1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this workload, collected "
                               f"at {traffic[2]})" if traffic else
                               "no committed PMC pass matches this workload (config / N / K / SR / mode / jitter / world)"),
            "algorithmic_bytes_per_launch": alg_bytes / n_launch,
            "algorithmic_bytes_formula": "pairs*8 + distinct points*1072 + samples*1064",
            "avg_launch_ms": acc_ms[2] / n_launch,
            "valid_pairs_per_launch": pairs / n_launch,
            "flops_per_pair": per_pair,
            "distinct_points_per_launch": upoints / n_launch,
            "point_part_ms": acc_ms[5] / n_launch,
            "reference_flops_per_pair": FLOPS_PER_PAIR,
            # the reference's per-pair arithmetic (542,720 FLOP) over the time of both kernels that replace it
            "reference_equivalent_tflops": pairs * FLOPS_PER_PAIR / t_both / 1e12 if t_both > 0 else 0.0,
        }

    def roofline_hbm(mode, acc_ms, cnt, n_launch):
        """The second regime of SURVEY.md section 8(d): stage Q (sample selection + neighbour search: k_select, the scans,
        k_expand, k_knn3, the valid-sample / distinct-point lists [+ k_pair_weights on the dense-unit path]) is priced
        against HBM.  achieved = section 8(d)'s ALGORITHMIC bytes of the stage -- R x 40 (ray in, pixel out) + R x D / 8
        (one occupancy bit per coarse sample) + S x 27 x 8 (cell probes of a shading sample) + C_cand x 16 (candidates
        distance-tested) -- / the stage's device time (HIP events on the render stream: select + knn)."""
        rays = n_local
        S, cand = cnt[3] / n_launch, cnt[5] / n_launch
        parts = {"rays x 40": rays * 40.0, "rays x D / 8": rays * 400 / 8.0, "valid samples x 216": S * 216.0,
                 "candidates x 16": cand * 16.0}
        alg = sum(parts.values())
        t = (acc_ms[0] + acc_ms[1]) / n_launch / 1e3
        achieved = alg / t / 1e9 if t > 0 else 0.0
        key = workload_key.replace(f":{args.precision}:", f":{mode}:")
        tr = pmc_traffic_sum(QUERY_STAGE_KERNELS, key)
        out = {
            "bound": "hbm", "stage": "Q = select + knn (k_select, scans, k_expand, k_knn3, sample / point lists)",
            "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
            "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_parts": parts,
            "algorithmic_bytes_formula": "R*40 + R*D/8 + S*216 + C_cand*16 (SURVEY.md section 8d; S = valid samples)",
            "avg_stage_ms": t * 1e3, "select_ms": acc_ms[0] / n_launch, "knn_ms": acc_ms[1] / n_launch,
            "traffic": tr[0] if tr else None,
            "traffic_per_kernel": tr[1] if tr else None,
            "traffic_source": (f"{tr[2]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this workload, collected at "
                               f"{tr[3]}; the scan kernels, which the file averages together with the scene build's, are "
                               f"not in the sum)" if tr else
                               "no committed PMC pass matches this workload"),
            "note": "latency-bound, not bandwidth-bound: the stage moves a few hundred MB per frame in ~1.3 ms through "
                    "chains of dependent probes (brick record -> list bounds -> candidates); its PMC traffic exceeds the "
                    "algorithmic bytes mainly by the K x 4-byte neighbour lists and 16-byte sample records it WRITES "
                    "for the shading stage, which section 8(d)'s formula does not count",
        }
        if tr:
            out["traffic_frac_of_peak"] = tr[0] / t / 1e9 / PEAK_HBM_GBS if t > 0 else None
        return out

    run_steps(rnd, 0, args.warmup)
    elapsed, acc_ms, acc_cnt, n_launch = timed(rnd, args.steps, args.warmup)
    rays_per_step = world * H * W
    value = rays_per_step * args.steps / elapsed

    # the other arithmetic mode, for the record: same workload, same steps and warm-up, never part of `value`
    alt = None
    if world == 1 and not emulate and not args.no_other_mode:
        alt_mode = "fp32" if args.precision == "bf16x3" else "bf16x3"
        rnd_alt = RendererHIP(scene, wh, SR=SR, K=K, D=400, radius_limit=4 * max(VSIZE[0], VSIZE[1]),
                              vsize_z=VSIZE[2], precision=alt_mode, jitter=args.jitter, seed=args.jitter_seed)
        rnd_alt._ws, rnd_alt._ws_key, rnd_alt.cap_samples = rnd._ws, rnd._ws_key, rnd.cap_samples
        run_steps(rnd_alt, 0, args.warmup)
        a_el, a_ms, a_cnt, a_n = timed(rnd_alt, args.steps, args.warmup)
        alt = {"mode": alt_mode, "dtype": "f32" if alt_mode == "fp32" else "bf16x3 products, f32 accumulate",
               "value": rays_per_step * args.steps / a_el, "unit": "rays/s", "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": a_el / args.steps * 1e3,
               "roofline": roofline(alt_mode, a_ms, a_cnt, a_n),
               "note": "opt-in fast mode: 16-bit-significand operand splits, narrower than the reference's fp32" if
                       alt_mode == "bf16x3" else "the reference's arithmetic"}

    # opt-in early ray termination (pnr_render_opts_t.early_stop_eps), for the record: NOT part of `value`, which
    # keeps the reference's sample set (every sample with a neighbour is shaded)
    early = None
    if world == 1 and not emulate and not args.no_other_mode:
        rnd_es = RendererHIP(scene, wh, SR=SR, K=K, D=400, radius_limit=4 * max(VSIZE[0], VSIZE[1]),
                             vsize_z=VSIZE[2], precision=args.precision, early_stop_eps=1e-5, jitter=args.jitter,
                             seed=args.jitter_seed)
        rnd_es._ws, rnd_es._ws_key, rnd_es.cap_samples = rnd._ws, rnd._ws_key, rnd.cap_samples
        run_steps(rnd_es, 0, 1)
        e_rgb = outs["rgb"].clone()
        run_steps(rnd, 0, 1)
        e_err = (outs["rgb"] - e_rgb).abs().max().item()
        e_el, e_ms, e_cnt, e_n = timed(rnd_es, 3, 1)
        early = {"early_stop_eps": 1e-5, "value": rays_per_step * 3 / e_el, "unit": "rays/s",
                 "ms_per_step": e_el / 3 * 1e3,
                 "stages_ms_per_launch": {n: e_ms[i] / e_n for i, n in enumerate(_lib.STAGE_NAMES)},
                 "samples_valid_per_launch": e_cnt[3] / e_n, "samples_shaded_per_launch": e_cnt[8] / e_n,
                 "max_abs_rgb_diff_vs_full_render": e_err,
                 "note": "rays stop being shaded once their transmittance is below eps (chunks of 3, 3, 6, 12, rest "
                         "samples); the reference shades every sample, so this is reported beside `value`, never as it"}

    # the training step of SURVEY.md section 8f rank 1, for the record (NOT part of `value`): pnr_render without clamp
    # at the reference's 0.3 jitter + pnr_render_backward on random pixels of view 0 -- 4096 rays is the batch
    # `ns-train pointnerf-original` draws (studio_config.py:20-21), 65536 shows the throughput regime
    train = None
    if world == 1 and not emulate and not args.no_other_mode:
        train = []
        full = synthetic.make_rays(H, W, cams[0][0], cams[0][1], cfgd["angle_x"]).to(dev)
        w_dev = {k: v.to(dev) for k, v in weights.items()}   # the trainer's parameters live on the GPU
        gen = torch.Generator().manual_seed(11)
        for n_rays in (4096, 65536):
            pick = torch.randperm(full.shape[0], generator=gen)[:n_rays].to(dev)
            dirs_t = full.index_select(0, pick).contiguous()
            g_rgb = torch.randn(n_rays, 3, generator=gen).to(dev)
            entry = {"rays": n_rays, "mode": args.precision}
            # twice: the backward recomputing the MLP chain itself (a render that knows nothing of it), and the render
            # writing the backward's activation tape as it shades (pnr_render_opts_t.d_tape: what the plugin's training
            # path does) -- the step is what counts; the backward's FLOPs are 3 x the forward's in the first form
            # (recompute + data + weight gradients) and 2 x in the second (the recompute moved into the render)
            for tape in (False, True):
                rnd_t = RendererHIP(scene, wh, SR=SR, K=K, D=400, radius_limit=4 * max(VSIZE[0], VSIZE[1]),
                                    vsize_z=VSIZE[2], precision=args.precision, eval_clamp=False, jitter=0.3, seed=1,
                                    tape=tape)
                fw, bw = [], []
                for it in range(5):
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    ev[0].record()
                    o = rnd_t.render(dirs_t, cams[0][0], cams[0][1], near, far)
                    ev[1].record()
                    rnd_t.backward(g_rgb, w_dev, cfgd["N"])
                    ev[2].record()
                    torch.cuda.synchronize()
                    if it >= 2:
                        fw.append(ev[0].elapsed_time(ev[1]))
                        bw.append(ev[1].elapsed_time(ev[2]))
                c = o["counters"]
                fwd_flops = c["pairs_valid"] * FLOPS_PER_PAIR + c["samples_valid"] * FLOPS_PER_SAMPLE
                f_ms, b_ms = sorted(fw)[1], sorted(bw)[1]
                tag = "_taped" if tape else ""
                entry.update({"forward_ms" + tag: f_ms, "backward_ms" + tag: b_ms, "step_ms" + tag: f_ms + b_ms,
                              "backward_tflops" + tag: (2 if tape else 3) * fwd_flops / (b_ms * 1e-3) / 1e12,
                              "step_tflops" + tag: 3 * fwd_flops / ((f_ms + b_ms) * 1e-3) / 1e12})
                del rnd_t
            # the form a training LOOP uses (and the plugin's step is built on): persistent point-gradient buffers the
            # backward accumulates into, the touched rows listed and zeroed again on the device, no host read anywhere --
            # ten steps back to back
            rnd_i = RendererHIP(scene, wh, SR=SR, K=K, D=400, radius_limit=4 * max(VSIZE[0], VSIZE[1]), vsize_z=VSIZE[2],
                                precision=args.precision, eval_clamp=False, jitter=0.3, seed=1, tape=True)
            o_i = rnd_i.render(dirs_t, cams[0][0], cams[0][1], near, far)
            cap_i, n_pts = rnd_i.cap_samples, cfgd["N"]
            into = {"embedding": torch.zeros(n_pts * 32, device=dev), "color": torch.zeros(n_pts * 3, device=dev),
                    "dir": torch.zeros(n_pts * 3, device=dev)}
            t_index, t_count = rnd_i.touched()

            def loop_step():
                rnd_i.render(dirs_t, cams[0][0], cams[0][1], near, far, cap_samples=cap_i, sync_counters=False, out=o_i)
                rnd_i.backward(g_rgb, w_dev, n_pts, into=into)
                rnd_i.touched(t_index, t_count)
                rnd_i.clear_point_grads(into["embedding"], into["color"], into["dir"], n_pts, t_index, t_count)
            for _ in range(3):
                loop_step()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                loop_step()
            ev[1].record()
            torch.cuda.synchronize()
            entry["step_ms_taped_loop"] = ev[0].elapsed_time(ev[1]) / 10
            del rnd_i, into
            entry.update({"rays_per_sec": n_rays / (entry["step_ms_taped"] * 1e-3), "pairs_valid": c["pairs_valid"],
                          "samples_valid": c["samples_valid"],
                          "note": "render + pnr_render_backward through the C ABI, dense [N, .] point gradients zero-filled "
                                  "by the caller each step.  Untagged: the backward recomputes the MLPs into a row-major "
                                  "tape (its FLOPs: 3 x the forward's, reference arithmetic 542,720 per pair + 137,984 per "
                                  "sample).  _taped: the render writes the tape (the backward's FLOPs: 2 x).  step_ms_taped_loop: the taped step "
                                  "as a training loop issues it -- persistent gradient buffers accumulated into, touched rows "
                                  "listed and zeroed on the device, no host read, ten steps back to back.  step_tflops = "
                                  "3 x the forward's FLOPs (forward + data + weight gradients: what the reference's "
                                  "autograd step performs; a recompute is this design's own and not counted) / step time; "
                                  "fp32 peak 157.3"})
            train.append(entry)

    # the PLUGIN surface, for the record beside `value` (which times the C-ABI call): the same workload through
    # PointNerf.get_outputs_for_camera_ray_bundle -- the path `ns-eval` / the trainer's eval images take
    # (studio_datamanager.py:104-110 -> studio_model.py:263) -- and one training step as `ns-train` runs it
    # (PointNerf.forward + get_loss_dict + backward + the after-step callback, studio_model.py:263-431)
    plugin = None
    if world == 1 and not emulate and not args.no_other_mode:
        plugin = plugin_legs(args, cfgd, points, weights, cams, dev, near, far)

    if rank == 0 or emulate:
        samples = acc_cnt[3]
        result = {
            "metric": "rays_per_sec", "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3 products, f32 accumulate" if args.precision == "bf16x3" else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: synthetic {cfgd['points']} cloud N={cfgd['N']}, {H}x{W} (HxW) image, D=400, "
                            f"SR={SR}, K={K}, P={cfgd['P']}, jitter={args.jitter:g} (seed {args.jitter_seed}), {world} "
                            f"view(s)/step",
                "workload_key": workload_key,
                "rays_per_step": rays_per_step, "global_batch": rays_per_step, "mlp_mode": args.precision,
                "rays_per_rank_per_step": n_local,
                "parallelism": f"ray-tile shard x{world} (16x16 tiles round-robin, owner rotated per view), one multi-camera render + one "
                               f"all_gather per step",
                "rays": "directions resident in HBM (pnr_render_views)" if args.rays_from_tensor else
                        "generated in the kernels from pose + intrinsics + pixel ids (pnr_render_camera)",
            },
            "roofline": roofline(args.precision, acc_ms, acc_cnt, n_launch),
            "roofline_hbm": roofline_hbm(args.precision, acc_ms, acc_cnt, n_launch),
            "stages_ms_per_launch": {n: acc_ms[i] / n_launch for i, n in enumerate(_lib.STAGE_NAMES)},
            "counters_per_launch": {n: acc_cnt[i] / n_launch for i, n in enumerate(_lib.COUNTER_NAMES)},
            "color_mlp_tflops": (samples * FLOPS_PER_SAMPLE / (acc_ms[3] / 1e3) / 1e12) if acc_ms[3] > 0 else None,
            "scene": {"occupied_voxels": info["occupied_voxels"], "points_in_voxel_lists": info["points_in_lists"],
                      "structure_bytes": info["device_bytes"], "build_s": build_s, "cap_samples": cap},
        }
        if rccl_world is not None:
            result["rccl_world"] = rccl_world
        if alt is not None:
            result["other_mode"] = alt
        if early is not None:
            result["with_early_ray_termination"] = early
        if train is not None:
            result["training_step"] = train
        if plugin is not None:
            result["plugin_eval"] = plugin["eval"]
            result["plugin_eval"]["fraction_of_value"] = plugin["eval"]["value"] / value
            result["plugin_training_step"] = plugin["train"]
        if world == 1 and not emulate and args.cpu_rays_side > 0:
            cb, ref, dirs, campos, camrot = cpu_baseline(points, weights, cfgd, args.cpu_rays_side, azimuths[0], args.cpu_passes,
                                                         args.cpu_budget_s, args.cpu_threads)
            # parity on the very same rays: HIP render vs the oracle that was just timed
            out = RendererHIP(scene, wh, SR=SR, K=K, D=400, radius_limit=4 * max(VSIZE[0], VSIZE[1]),
                              vsize_z=VSIZE[2], precision=args.precision).render(dirs.to(dev), campos, camrot, near, far)
            # (jitter 0 on both sides: the oracle pass that was timed and this render see the same sample positions)
            err = (out["rgb"].cpu() - ref["coarse_raycolor"]).abs().max().item()
            result["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "threads", "physical_cores",
                                                         "logical_cpus", "cpu_quota", "kind", "sample", "seconds",
                                                         "cpu_model", "as_written", "counts")}
            result["parity_on_cpu_sample"] = {
                "max_abs_rgb_err": err, "ray_mask_equal": bool(torch.equal(out["ray_mask"].cpu(), ref["ray_mask"])),
                "psnr_vs_oracle_db": float(-10 * torch.log10(((out["rgb"].cpu() - ref["coarse_raycolor"]) ** 2).mean()
                                                             + 1e-20))}
            result["speedup_vs_cpu_baseline"] = value / cb["value"]
            # BASELINE cfg[0] whole, on both sides (the one configuration defined as the CPU reference path's)
            if not args.no_cfg0 and args.config == "cfg1_chair_6m":
                del out
                result["cfg0_whole_frame"] = cfg0_leg(args, dev)
        print(json.dumps(result), flush=True)
    if world > 1 and not emulate:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
