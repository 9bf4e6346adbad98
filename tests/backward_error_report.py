"""Not a test: per-tensor gradient errors of pnr_render_backward in both arithmetic modes against autograd through the
CPU oracle (run on a GPU box: python tests/backward_error_report.py).  Lives under tests/ because it uses the oracle."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import torch, pnr_oracle as oracle
oracle.build_c_oracle()
from helpers import build_hip, camera_rays, oracle_cfg, small_scene
from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.renderer import MLP_TENSOR_ORDER, RendererHIP
import test_gpu_backward as T
dev = torch.device("cuda:0")
N, SR, K, P = 60000, 80, 8, 12
pts = small_scene(N); cfg = oracle_cfg(oracle, SR=SR, K=K, P=P)
w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
campos, camrot, dirs = camera_rays(24, 24, az=35.0)
torch.manual_seed(5); G = torch.randn(dirs.shape[0], 3)
ref, want = T._oracle_grads(oracle, pts, w, cfg, campos, camrot, dirs, G, True)
scene, wh, hyp, info = build_hip(pts, cfg, dev, weights=w)
for prec in ("fp32", "bf16x3"):
    rnd = RendererHIP(scene, wh, SR=SR, K=K, D=cfg.z_depth_dim, radius_limit=float(oracle.radius_limit(cfg)), vsize_z=cfg.vsize[2], precision=prec, eval_clamp=False)
    rnd.render(dirs.to(dev), campos, camrot, 2.0, 6.0)
    got = rnd.backward(G.to(dev), w, N)
    print(prec, "rgb", (got["rgb"].cpu() - ref["coarse_raycolor"]).abs().max().item())
    for k in want:
        sc = want[k].abs().max().item(); e = (got[k].cpu() - want[k]).abs()
        l2 = (got[k].cpu() - want[k]).norm().item() / want[k].norm().item()
        print("  %-36s max rel %.2e  l2 rel %.2e (n>1e-3*scale: %d of %d)" % (k, e.max().item() / sc, l2, int((e > 1e-3 * sc).sum()), e.numel()))
