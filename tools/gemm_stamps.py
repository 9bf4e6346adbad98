"""GPU box, diagnostic (-DPNR_STAMPS=1 build): shader-clock cycles of the phases of a chunk of k_gemm_nt_bf16x3 (wave 0 of
every workgroup), averaged over the chunks of a few backward calls.
    bash tools/build_variant.sh gstamps -DPNR_STAMPS=1
    PNR_LIB=$PWD/pointnerf2studio_amd/_abl/libpnr_gstamps.so python tools/gemm_stamps.py [rays]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnerf2studio_amd import _lib, synthetic  # noqa: E402
from pointnerf2studio_amd.renderer import RendererHIP, SceneHIP, WeightsHIP, grid_hyperparameters  # noqa: E402

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
pts = synthetic.make_points(6_000_000)
w = {k: v.to(dev) for k, v in synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1).items()}
xyz = pts["xyz"].to(dev)
hyp = grid_hyperparameters(xyz, (0.004,) * 3, (2, 2, 2), (3, 3, 3), list(synthetic.CHAIR_RANGES))
scene = SceneHIP()
scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, (3, 3, 3), (3, 3, 3), 12, 410000, True)
scene.pack_points(xyz, pts["embedding"].to(dev), pts["conf"].to(dev), pts["dir"].to(dev), pts["color"].to(dev))
wh = WeightsHIP()
wh.pack(w, pts["Rw2c"], dev)
rnd = RendererHIP(scene, wh, precision="bf16x3", eval_clamp=False)
campos, camrot = synthetic.make_camera(35.0, 30.0)
full = synthetic.make_rays(800, 800, campos, camrot)
pick = torch.randperm(full.shape[0], generator=torch.Generator().manual_seed(11))[:rays]
dirs = full[pick].contiguous().to(dev)
G = torch.randn(rays, 3, device=dev)
lib = _lib.load()
buf = (C.c_uint64 * 32)()
for it in range(3):
    rnd.render(dirs, campos, camrot, 2.0, 6.0)
    rnd.backward(G, w, 6_000_000)
    lib.pnr_debug_read(buf, 1)
n = max(buf[4], 1)
print(f"chunks {buf[4]}: per chunk  load issue {buf[0] / n:7.0f}  MFMA block {buf[1] / n:7.0f}  wait+split+LDS store {buf[2] / n:7.0f}"
      f"  barrier {buf[3] / n:7.0f}  whole loop {buf[5] / n:7.0f} cycles")
