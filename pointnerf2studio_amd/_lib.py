"""ctypes binding of libpnr_hip.so (include/pnr.h).  Tensors in, tensors out; no torch C++ ABI.

There is NO fallback: if the HIP library is missing and cannot be built, or a call fails, a RuntimeError is raised.
Where the library is looked for (`find_library`): $PNR_LIB, then next to this file (a source tree after
`python -m pointnerf2studio_amd.build`, or a wheel that was packaged with the library), then the per-user cache an
earlier first-use build wrote.  An installed copy without a library builds it ONCE on first use from the HIP sources it
ships (pyproject.toml: package-data) when hipcc is present -- $PNR_NO_AUTOBUILD=1 turns that off.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# PNR_LIB overrides the library path for diagnostic (ablation) builds; never set in normal use
LIB_PATH = os.environ.get("PNR_LIB") or os.path.join(PKG_DIR, "libpnr_hip.so")

NUM_COUNTERS = 10
COUNTER_NAMES = ["rays_hit", "rays_kept", "samples_selected", "samples_valid", "pairs_valid", "candidates",
                 "overflow", "points_unique", "samples_shaded", "reserved"]
POINT_ROW_FLOATS = 48
MAX_K = 32
MAX_D = 512
MAX_CAMS = 16

# every symbol include/pnr.h declares (tests check that the library exports all of them)
EXPORTED_SYMBOLS = [
    "pnr_last_error", "pnr_version", "pnr_abi_sizes", "pnr_jitter_uniform",
    "pnr_scene_create", "pnr_scene_destroy", "pnr_scene_build", "pnr_scene_info", "pnr_points_pack",
    "pnr_scene_update", "pnr_scene_update_info", "pnr_render_probe", "pnr_points_pack_rows", "pnr_render_touched", "pnr_points_bind", "pnr_point_grads_clear",
    "pnr_weights_create", "pnr_weights_destroy", "pnr_weights_pack", "pnr_weights_update",
    "pnr_query_workspace_bytes", "pnr_query_raypos",
    "pnr_render_workspace_bytes", "pnr_render_workspace_bytes_for", "pnr_render", "pnr_render_views", "pnr_render_pose",
    "pnr_render_camera", "pnr_render_camera_lists", "pnr_camera_rays", "pnr_pinhole_ray",
    "pnr_render_taps",
    "pnr_backward_workspace_bytes", "pnr_render_backward",
    "pnr_conf_loss_workspace_bytes", "pnr_conf_loss", "pnr_conf_loss_backward",
    "pnr_rows_merge", "pnr_adam_rows",
    "pnr_profile_enable", "pnr_profile_calls", "pnr_profile_read",
]
NUM_STAGES = 6
STAGE_NAMES = ["select", "knn", "shade_pairs", "shade_color", "composite", "point_part"]


class GridParams(C.Structure):
    _fields_ = [("ranges", C.c_float * 6), ("vox", C.c_float * 3), ("dims", C.c_int32 * 3),
                ("kernel_size", C.c_int32 * 3), ("query_size", C.c_int32 * 3), ("P", C.c_int32),
                ("max_o", C.c_int32), ("compat_drop_voxel0", C.c_int32)]


class CameraC(C.Structure):
    _fields_ = [("campos", C.c_float * 3), ("camrotc2w", C.c_float * 9), ("near_plane", C.c_float),
                ("far_plane", C.c_float)]


class ViewC(C.Structure):
    """pnr_view_t: pose + pinhole intrinsics of one view (nerfstudio Cameras)."""
    _fields_ = [("campos", C.c_float * 3), ("camrotc2w", C.c_float * 9), ("near_plane", C.c_float),
                ("far_plane", C.c_float), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class RenderOpts(C.Structure):
    _fields_ = [("SR", C.c_int32), ("K", C.c_int32), ("D", C.c_int32), ("radius_limit", C.c_float),
                ("vsize_z", C.c_float), ("eval_clamp", C.c_int32), ("bg", C.c_float * 3), ("precision", C.c_int32), ("jitter", C.c_float), ("seed", C.c_uint32),
                ("early_stop_eps", C.c_float), ("d_tape", C.c_void_p), ("tape_bytes", C.c_size_t)]


PRECISION = {"fp32": 0, "bf16x3": 1}


class RenderTaps(C.Structure):
    _fields_ = [("smp_loc", C.c_void_p), ("smp_ray", C.c_void_p), ("smp_pidx", C.c_void_p),
                ("smp_out", C.c_void_p), ("ray_cnt", C.c_void_p), ("ray_off", C.c_void_p), ("ray_dirs", C.c_void_p)]


class ProbeC(C.Structure):
    """pnr_probe_t"""
    _fields_ = [("d_max_opacity", C.c_void_p), ("d_max_loc", C.c_void_p), ("d_far_dist", C.c_void_p),
                ("d_avg_color", C.c_void_p), ("d_avg_dir", C.c_void_p), ("d_avg_conf", C.c_void_p),
                ("d_avg_embedding", C.c_void_p), ("d_max_index", C.c_void_p)]


class GradsC(C.Structure):
    _fields_ = [("d_embedding", C.c_void_p), ("d_color", C.c_void_p), ("d_dir", C.c_void_p),
                ("d_w", C.c_void_p * 9), ("d_b", C.c_void_p * 9),
                ("d_point_grads", C.c_void_p), ("d_point_index", C.c_void_p), ("point_cap", C.c_int64)]


class AdamTensorC(C.Structure):
    """pnr_adam_tensor_t"""
    _fields_ = [("d_param", C.c_void_p), ("d_grad", C.c_void_p), ("d_exp_avg", C.c_void_p), ("d_exp_avg_sq", C.c_void_p),
                ("width", C.c_int32)]


ADAM_MAX_TENSORS = 8

_lib: Optional[C.CDLL] = None


def find_library() -> str:
    """Path of libpnr_hip.so, building it on first use where that is possible; raises (never falls back) otherwise."""
    import sys
    from . import build
    if os.path.exists(LIB_PATH):
        return LIB_PATH
    default = os.path.join(PKG_DIR, "libpnr_hip.so")
    why = f"{LIB_PATH} is missing"
    if LIB_PATH == default and not os.environ.get("PNR_LIB"):
        cached = os.path.join(build.cache_dir(), "libpnr_hip.so")
        if os.path.exists(cached):
            return cached
        if os.environ.get("PNR_NO_AUTOBUILD"):
            why += " and PNR_NO_AUTOBUILD is set"
        elif not build.sources_present():
            why += f" and the HIP sources are not under {build.CSRC}"
        elif build.find_hipcc() is None:
            why += " and there is no hipcc to build it with"
        else:
            out_dir = build.default_out_dir()
            print(f"pointnerf2studio_amd: building libpnr_hip.so for gfx950 into {out_dir} (first use, ~1-2 min)",
                  file=sys.stderr, flush=True)
            return build.build_library(out_dir=None if out_dir == PKG_DIR else out_dir)
    raise RuntimeError(
        f"{why}: build it with `python -m pointnerf2studio_amd.build` "
        "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for the render path.")


def load() -> C.CDLL:
    """Loads libpnr_hip.so (find_library); raises (never falls back) when it is absent and cannot be built."""
    global _lib
    if _lib is not None:
        return _lib
    lib = C.CDLL(find_library())
    vp, i64, i32, f32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t
    lib.pnr_last_error.restype = C.c_char_p
    lib.pnr_last_error.argtypes = []
    lib.pnr_version.restype = C.c_int
    lib.pnr_abi_sizes.argtypes = [C.POINTER(C.c_int64 * 8)]
    lib.pnr_jitter_uniform.restype = C.c_float
    lib.pnr_jitter_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    lib.pnr_scene_create.argtypes = [C.POINTER(vp)]
    lib.pnr_scene_destroy.argtypes = [vp]
    lib.pnr_scene_build.argtypes = [vp, vp, i64, C.POINTER(GridParams), vp]
    lib.pnr_scene_info.argtypes = [vp, C.POINTER(i64 * 8)]
    lib.pnr_points_pack.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp]
    lib.pnr_points_pack_rows.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, vp]
    lib.pnr_point_grads_clear.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp]
    lib.pnr_points_bind.argtypes = [vp, vp, vp, vp, vp, vp, i64]
    lib.pnr_render_touched.argtypes = [vp, C.POINTER(RenderOpts), i64, vp, sz, i64, vp, i64, vp, vp]
    lib.pnr_scene_update.argtypes = [vp, vp, i64, C.POINTER(GridParams), vp, vp]
    lib.pnr_scene_update_info.argtypes = [vp, C.POINTER(i64 * 4)]
    lib.pnr_render_probe.argtypes = [vp, C.POINTER(CameraC), i32, vp, i64, C.POINTER(RenderOpts), i64, vp, sz, i64,
                                     C.POINTER(ProbeC), vp]
    lib.pnr_weights_create.argtypes = [C.POINTER(vp)]
    lib.pnr_weights_destroy.argtypes = [vp]
    lib.pnr_weights_pack.argtypes = [vp, C.POINTER(vp * 9), C.POINTER(vp * 9), vp, vp]
    lib.pnr_weights_update.argtypes = [vp, C.POINTER(vp * 9), C.POINTER(vp * 9), i32, vp]
    lib.pnr_query_workspace_bytes.restype = sz
    lib.pnr_query_workspace_bytes.argtypes = [i64, i32, i32, i32]
    lib.pnr_query_raypos.argtypes = [vp, vp, i64, i32, i32, i32, f32, vp, vp, vp, vp, vp, sz, vp]
    lib.pnr_render_workspace_bytes.restype = sz
    lib.pnr_render_workspace_bytes.argtypes = [i64, i64, i32]
    lib.pnr_render_workspace_bytes_for.restype = sz
    lib.pnr_render_workspace_bytes_for.argtypes = [vp, C.POINTER(RenderOpts), i64, i64]
    lib.pnr_render.argtypes = [vp, vp, vp, i64, C.POINTER(CameraC), vp, C.POINTER(RenderOpts), vp, vp, vp, vp, vp,
                               vp, sz, i64, vp]
    lib.pnr_render_pose.argtypes = [vp, vp, vp, i64, vp, vp, f32, f32, vp, C.POINTER(RenderOpts), vp, vp, vp, vp, vp, vp, sz,
                                    i64, vp]
    lib.pnr_render_views.argtypes = [vp, vp, vp, i64, C.POINTER(CameraC), i32, vp, i64, vp, C.POINTER(RenderOpts), vp, vp,
                                     vp, vp, vp, vp, sz, i64, vp]
    lib.pnr_render_camera.argtypes = [vp, vp, C.POINTER(ViewC), i32, i32, i32, vp, i64, vp, C.POINTER(RenderOpts), vp, vp, vp,
                                      vp, vp, vp, sz, i64, vp]
    lib.pnr_render_camera_lists.argtypes = lib.pnr_render_camera.argtypes
    lib.pnr_camera_rays.argtypes = [C.POINTER(ViewC), i32, i32, i32, vp, i64, vp, vp]
    lib.pnr_pinhole_ray.restype = None
    lib.pnr_pinhole_ray.argtypes = [C.POINTER(ViewC), i32, i32, C.POINTER(C.c_float * 3)]
    lib.pnr_render_taps.argtypes = [vp, sz, i64, i64, i32, C.POINTER(RenderTaps)]
    lib.pnr_backward_workspace_bytes.restype = sz
    lib.pnr_backward_workspace_bytes.argtypes = [i64, i32]
    lib.pnr_render_backward.argtypes = [vp, vp, C.POINTER(vp * 9), C.POINTER(vp * 9), vp, i64, C.POINTER(CameraC), i32,
                                        vp, i64, C.POINTER(RenderOpts), vp, vp, sz, i64, vp, sz, C.POINTER(GradsC), vp,
                                        vp]
    lib.pnr_conf_loss_workspace_bytes.restype = sz
    lib.pnr_conf_loss_workspace_bytes.argtypes = []
    lib.pnr_conf_loss.argtypes = [vp, C.POINTER(RenderOpts), i64, vp, sz, i64, vp, f32, vp, vp, vp]
    lib.pnr_conf_loss_backward.argtypes = [vp, C.POINTER(RenderOpts), i64, vp, sz, i64, vp, f32, vp, vp, vp, vp, vp]
    lib.pnr_rows_merge.argtypes = [vp, i64, vp, vp, i64, vp, i64, vp, vp]
    lib.pnr_adam_rows.argtypes = [C.POINTER(AdamTensorC), i32, i64, vp, i64, vp, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_double, vp]
    lib.pnr_profile_enable.argtypes = [C.c_int]
    lib.pnr_profile_calls.restype = C.c_int64
    lib.pnr_profile_calls.argtypes = []
    lib.pnr_profile_read.argtypes = [C.c_int64, C.POINTER(C.c_float * NUM_STAGES)]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("pnr_last_error", "pnr_query_workspace_bytes", "pnr_render_workspace_bytes",
                        "pnr_render_workspace_bytes_for", "pnr_profile_calls", "pnr_jitter_uniform",
                        "pnr_backward_workspace_bytes", "pnr_pinhole_ray", "pnr_conf_loss_workspace_bytes"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().pnr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed with status {rc}: {msg}")
