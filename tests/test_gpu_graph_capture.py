"""The calls of the render path are CAPTURABLE into a hipGraph: sizes live in device memory, nothing returns to the host,
nothing is allocated, and every clear is a kernel (a graph holding hipMemsetAsync nodes faulted on its second launch under
ROCm 7.2 -- found with tools/graph_replay.py, which is why the render path has none).  torch.cuda.CUDAGraph drives
hipStreamBeginCapture / hipGraphLaunch around the ctypes calls.  Each graph is replayed several times with direct launches
of the same call in between (the sequence that exposed the memset nodes); every replay must reproduce the directly
launched results bit for bit.
"""
import pytest
import torch

from helpers import build_hip, camera_rays, oracle_cfg, small_scene
from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.renderer import RendererHIP, View

pytestmark = pytest.mark.gpu


def _capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    return g, keep


def _scene(oracle, device, SR, K, N=40000):
    pts = small_scene(N)
    cfg = oracle_cfg(oracle, SR=SR, K=K)
    weights = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    scene, wh, _, _ = build_hip(pts, cfg, device, weights)
    return pts, weights, scene, wh


@pytest.mark.parametrize("precision,early", [("fp32", 0.0), ("bf16x3", 0.0), ("fp32", 1e-5)])
def test_eval_render_replays_bit_identically(oracle, gpu_device, precision, early):
    dev = gpu_device
    _, _, scene, wh = _scene(oracle, dev, SR=40, K=8)
    campos, camrot, dirs = camera_rays(48, 48)
    dirs = dirs.to(dev)
    rnd = RendererHIP(scene, wh, SR=40, K=8, precision=precision, jitter=0.3, seed=5, early_stop_eps=early)
    out = rnd.render(dirs, campos, camrot, 2.0, 6.0)
    assert out["counters"]["rays_kept"] > 100 and out["counters"]["overflow"] == 0
    cap = rnd.cap_samples

    def call():
        return rnd.render(dirs, campos, camrot, 2.0, 6.0, cap_samples=cap, sync_counters=False, out=out)
    call()
    torch.cuda.synchronize()
    keys = ("rgb", "depth", "acc", "ray_mask", "counters_dev")
    want = {k: out[k].clone() for k in keys}
    g, _ = _capture(call)
    for rep in range(3):
        for k in keys:
            out[k].fill_(0 if rep % 2 == 0 else 1)
        g.replay()
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(out[k], want[k]), f"replay {rep}: {k}"
        for _ in range(3):          # direct launches between the replays
            call()
        torch.cuda.synchronize()


def test_camera_render_replays_bit_identically(oracle, gpu_device):
    """pnr_render_camera: rays generated in the kernels (the call the bench times)"""
    dev = gpu_device
    _, _, scene, wh = _scene(oracle, dev, SR=40, K=8)
    campos, camrot = synthetic.make_camera(35.0, 30.0)
    view = View.from_angle(campos, camrot, 40, 40, 0.6911112070083618)
    rnd = RendererHIP(scene, wh, SR=40, K=8, jitter=0.3, seed=9)
    out = rnd.render_camera([view], 40, 40)
    assert out["counters"]["rays_kept"] > 100
    cap = rnd.cap_samples

    def call():
        return rnd.render_camera([view], 40, 40, cap_samples=cap, sync_counters=False, out=out)
    call()
    torch.cuda.synchronize()
    want = {k: out[k].clone() for k in ("rgb", "depth", "ray_mask", "counters_dev")}
    g, _ = _capture(call)
    for rep in range(2):
        for k in want:
            out[k].fill_(0)
        g.replay()
        torch.cuda.synchronize()
        for k in want:
            assert torch.equal(out[k], want[k]), f"replay {rep}: {k}"
        call()
        torch.cuda.synchronize()


@pytest.mark.parametrize("K", [8, 12])
def test_training_step_replays_bit_identically(oracle, gpu_device, K):
    """taped render (K = 8) / recomputing backward (K = 12) + backward into persistent point-gradient buffers + touched-row
    list + row clear, as one graph"""
    dev = gpu_device
    N = 40000
    pts, weights, scene, wh = _scene(oracle, dev, SR=40, K=K, N=N)
    w_dev = {k: v.to(dev).contiguous() for k, v in weights.items()}
    campos, camrot, dirs = camera_rays(40, 40)
    dirs = dirs.to(dev)
    g_rgb = torch.randn(dirs.shape[0], 3, generator=torch.Generator().manual_seed(3)).to(dev)
    rnd = RendererHIP(scene, wh, SR=40, K=K, eval_clamp=False, jitter=0.3, seed=2, tape=True)
    out = rnd.render(dirs, campos, camrot, 2.0, 6.0)
    assert out["counters"]["rays_kept"] > 100
    cap = rnd.cap_samples
    into = {"embedding": torch.zeros(N * 32, device=dev), "color": torch.zeros(N * 3, device=dev),
            "dir": torch.zeros(N * 3, device=dev)}
    index, count = rnd.touched()

    def step():
        rnd.render(dirs, campos, camrot, 2.0, 6.0, cap_samples=cap, sync_counters=False, out=out)
        g = rnd.backward(g_rgb, w_dev, N, into=into)
        rnd.touched(index, count)
        # (what an optimiser would read here: the accumulated rows, before they are zeroed for the next step)
        g["point_rows"] = torch.cat([into["embedding"].view(N, 32), into["color"].view(N, 3), into["dir"].view(N, 3)], 1)
        rnd.clear_point_grads(into["embedding"], into["color"], into["dir"], N, index, count)
        return g
    res = step()
    torch.cuda.synchronize()
    want = {k: v.clone() for k, v in res.items()}
    assert float(want["point_rows"].abs().max()) > 0 and all(float(t.abs().max()) == 0 for t in into.values())
    assert float(want["mlp_base.layers.0.weight"].abs().max()) > 0
    g, gres = _capture(step)          # gres: the tensors the graph writes at every replay
    for rep in range(3):
        for v in gres.values():
            v.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        for k in want:
            assert torch.equal(gres[k], want[k]), f"replay {rep}: {k}"
        assert all(float(t.abs().max()) == 0 for t in into.values()), f"replay {rep}: rows not cleared"
        step()
        torch.cuda.synchronize()


def test_replayed_training_loop_equals_the_launched_one(oracle, gpu_device):
    """A whole loop step as ONE graph -- pnr_weights_update, pnr_points_pack_rows (the rows the previous step changed),
    taped render, pnr_conf_loss, pnr_render_backward, pnr_conf_loss_backward, pnr_render_touched, an in-place SGD update of
    every parameter (torch ops), pnr_point_grads_clear -- replayed four times from a saved state, against the same four
    steps launched one by one: every parameter and the last image equal bit for bit.  (The graph repeats ONE batch and
    ONE jitter seed: host arguments are baked in.)"""
    dev = gpu_device
    N, K, lr = 40000, 8, 1e-3
    pts, weights, scene, wh = _scene(oracle, dev, SR=40, K=K, N=N)
    w_dev = {k: v.to(dev).contiguous() for k, v in weights.items()}
    xyz = pts["xyz"].to(dev).contiguous()
    live = {k: pts[k].to(dev).contiguous() for k in ("embedding", "conf", "dir", "color")}
    campos, camrot, dirs = camera_rays(40, 40)
    dirs = dirs.to(dev)
    g_rgb = torch.randn(dirs.shape[0], 3, generator=torch.Generator().manual_seed(3)).to(dev) * 1e-2
    upstream = torch.tensor([1e-4], device=dev)
    rnd = RendererHIP(scene, wh, SR=40, K=K, eval_clamp=False, jitter=0.3, seed=2, tape=True)
    out = rnd.render(dirs, campos, camrot, 2.0, 6.0)
    cap = rnd.cap_samples
    into = {"embedding": torch.zeros(N * 32, device=dev), "color": torch.zeros(N * 3, device=dev),
            "dir": torch.zeros(N * 3, device=dev)}
    grad_conf = torch.zeros(N, device=dev)
    index, count = rnd.touched()

    def step():
        wh.update(w_dev, dev, "fp32")
        scene.pack_point_rows(xyz, live["embedding"], live["conf"], live["dir"], live["color"], index, count)
        rnd.render(dirs, campos, camrot, 2.0, 6.0, cap_samples=cap, sync_counters=False, out=out)
        fwd = rnd.conf_loss(live["conf"], 1e-3)
        g = rnd.backward(g_rgb, w_dev, N, into=into)
        rnd.conf_loss_backward(live["conf"], 1e-3, fwd, upstream, grad_conf)
        rnd.touched(index, count)
        for name, p in w_dev.items():
            p.add_(g[name], alpha=-lr)
        live["embedding"].view(-1).add_(into["embedding"], alpha=-lr)
        live["color"].view(-1).add_(into["color"], alpha=-lr)
        live["dir"].view(-1).add_(into["dir"], alpha=-lr)
        live["conf"].view(-1).add_(grad_conf, alpha=-lr)
        rnd.clear_point_grads(into["embedding"], into["color"], into["dir"], N, index, count)
        grad_conf.zero_()
        return fwd

    state = {**w_dev, **{"pt." + k: v for k, v in live.items()}, "index": index, "count": count}
    saved = {k: v.clone() for k, v in state.items()}

    def restore():
        for k, v in state.items():
            v.copy_(saved[k])
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    want = {k: v.clone() for k, v in state.items()}
    want_rgb = out["rgb"].clone()
    moved = sum(float((want[k] - saved[k]).abs().max()) > 0 for k in ("pt.embedding", "pt.color", "pt.conf",
                                                                       "mlp_base.layers.0.weight", "field_output_color.net.bias"))
    assert moved == 5, "the steps must change points and weights for the comparison to mean anything"
    restore()
    g, _ = _capture(step)          # (its warm-up steps move the state: restored below)
    restore()
    out["rgb"].fill_(0)
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
    for k in state:
        assert torch.equal(state[k], want[k]), k
    assert torch.equal(out["rgb"], want_rgb)
