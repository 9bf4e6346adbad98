"""Row G of the scope table: the drop-in is installable and nerfstudio can discover it.

The reference registers its method through a packaging entry point (reference pyproject.toml:20-21:
`[project.entry-points.'nerfstudio.method_configs'] pointnerf2studio = 'pointnerf.nerfstudio.studio_config:pointnerf_original'`);
nerfstudio's `ns-train` lists `importlib.metadata.entry_points(group="nerfstudio.method_configs")`, loads each and keys
the result by `spec.config.method_name` [ns-mem plugins/registry.py].  Here the package is pip-installed (no index, no
dependencies, no build isolation) from a copy of the packaged files into a scratch directory, and a fresh interpreter
that sees ONLY that directory -- cwd outside the repo, the repo not on sys.path -- must find the entry point, load it
against the stand-in nerfstudio (tests/fake_nerfstudio_check.py: nerfstudio itself is not installable in the build
image) and get a spec whose method name is "pointnerf-original"; the installed copy must also bring its library, the
HIP sources and the C-ABI header, and `_lib.load()` must work from there.
"""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def installed(tmp_path_factory):
    from pointnerf2studio_amd import build
    build.build_library()           # (a wheel is packaged AFTER the build: the library is package data)
    top = tmp_path_factory.mktemp("pkg")
    src, site = top / "src", top / "site"
    src.mkdir()
    for name in ("pyproject.toml", "setup.py", "README.md"):
        shutil.copy2(os.path.join(ROOT, name), src / name)
    shutil.copytree(os.path.join(ROOT, "include"), src / "include")
    shutil.copytree(os.path.join(ROOT, "pointnerf2studio_amd"), src / "pointnerf2studio_amd",
                    ignore=shutil.ignore_patterns("_obj", "__pycache__"))
    p = subprocess.run([sys.executable, "-m", "pip", "install", "--no-deps", "--no-build-isolation", "--no-index",
                        "--quiet", "--target", str(site), str(src)], capture_output=True, text=True, timeout=600,
                       cwd=str(top))
    assert p.returncode == 0, p.stdout + p.stderr
    return top, site


def _run(site, cwd, code, **env_extra):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env.update(PYTHONPATH=str(site), PNR_TEST_INSTALLED_COPY="1", **env_extra)
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, timeout=600,
                          cwd=str(cwd), env=env)


def test_installed_copy_ships_library_sources_and_header(installed):
    top, site = installed
    pkg = site / "pointnerf2studio_amd"
    assert (pkg / "libpnr_hip.so").exists() and (pkg / "include" / "pnr.h").exists()
    assert (pkg / "csrc" / "pnr_query.hip").exists() and (pkg / "csrc" / "pnr_internal.h").exists()
    assert not (pkg / "csrc" / "_obj").exists()
    p = _run(site, top, """
        import os, re
        import pointnerf2studio_amd
        from pointnerf2studio_amd import _lib, build
        here = os.path.dirname(pointnerf2studio_amd.__file__)
        assert here.startswith(os.environ["PYTHONPATH"]), here
        assert build.INCLUDE == os.path.join(here, "include") and build.sources_present()
        lib = _lib.load()
        assert _lib.find_library() == os.path.join(here, "libpnr_hip.so")
        declared = set(re.findall(r"\\b(pnr_[a-z0-9_]+)\\s*\\(", open(os.path.join(build.INCLUDE, "pnr.h")).read()))
        declared -= {"pnr_scene", "pnr_weights"}
        assert declared == set(_lib.EXPORTED_SYMBOLS) and lib.pnr_version() == 100
        print("installed copy ok")
    """)
    assert p.returncode == 0 and "installed copy ok" in p.stdout, p.stdout + p.stderr


def test_entry_point_resolves_to_pointnerf_original(installed):
    top, site = installed
    p = _run(site, top, f"""
        import importlib.metadata as md, importlib.util, os, sys
        assert {ROOT!r} not in sys.path and os.getcwd() != {ROOT!r}
        eps = [e for e in md.entry_points(group="nerfstudio.method_configs") if e.name == "pointnerf2studio"]
        assert len(eps) == 1, eps
        assert eps[0].value == "pointnerf2studio_amd.studio_config:pointnerf_original"      # reference pyproject.toml:20-21
        # without nerfstudio the package still imports; the spec exists only where nerfstudio does
        assert eps[0].load() is None
        spec = importlib.util.spec_from_file_location("fake_ns", {os.path.join(ROOT, "tests", "fake_nerfstudio_check.py")!r})
        fake = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(fake)
        assert {ROOT!r} not in sys.path
        fake.install_stand_in()
        method = eps[0].load()           # what nerfstudio's plugin registry does per entry point
        import pointnerf2studio_amd
        assert pointnerf2studio_amd.__file__.startswith({str(site)!r}), pointnerf2studio_amd.__file__
        assert method is not None and method.config.method_name == "pointnerf-original"   # studio_config.py:14
        assert method.config.pipeline.model._target.__name__ == "PointNerf"
        print("entry point ok:", method.config.method_name)
    """)
    assert p.returncode == 0 and "entry point ok: pointnerf-original" in p.stdout, p.stdout + p.stderr


def test_first_use_build_of_an_installed_copy_without_library(installed, tmp_path):
    """A copy installed without the binary (an sdist install): find_library builds from the shipped sources when hipcc
    exists -- into the package directory, or the per-user cache when that is read-only -- and raises where it does not.
    The compile itself is build_library's (covered by every other test); here it is replaced by a recorder."""
    top, site = installed
    bare = tmp_path / "bare"
    shutil.copytree(site, bare)
    os.remove(bare / "pointnerf2studio_amd" / "libpnr_hip.so")
    code = """
        import os, shutil, sys
        from pointnerf2studio_amd import _lib, build
        calls = []
        def fake_build(force=False, verbose=False, out_dir=None):
            calls.append(out_dir)
            d = build.PKG_DIR if out_dir is None else out_dir
            os.makedirs(d, exist_ok=True)
            shutil.copy2(os.environ["PNR_TEST_PREBUILT"], os.path.join(d, "libpnr_hip.so"))
            return os.path.join(d, "libpnr_hip.so")
        build.build_library = fake_build
        mode = os.environ["PNR_TEST_MODE"]
        if mode == "no_hipcc":
            build.find_hipcc = lambda: None
            try:
                _lib.load()
            except RuntimeError as e:
                assert "no hipcc" in str(e) and "no CPU or PyTorch fallback" in str(e), e
                print("raised ok")
        elif mode == "off":
            try:
                _lib.load()
            except RuntimeError as e:
                assert "PNR_NO_AUTOBUILD" in str(e), e
                print("raised ok")
        else:
            if mode == "readonly":
                build._writable = lambda d: False
            lib = _lib.load()
            assert lib.pnr_version() == 100 and len(calls) == 1
            want = None if mode == "writable" else build.cache_dir()
            assert calls[0] == want, (calls, want)
            _lib._lib = None
            assert _lib.load().pnr_version() == 100 and len(calls) == 1      # found again, not rebuilt
            print("built ok")
    """
    prebuilt = str(site / "pointnerf2studio_amd" / "libpnr_hip.so")
    for mode, extra, want in (("no_hipcc", {}, "raised ok"), ("off", {"PNR_NO_AUTOBUILD": "1"}, "raised ok"),
                              ("readonly", {"PNR_CACHE_DIR": str(tmp_path / "cache")}, "built ok"),
                              ("writable", {}, "built ok")):
        p = _run(bare, top, code, PNR_TEST_MODE=mode, PNR_TEST_PREBUILT=prebuilt, **extra)
        assert p.returncode == 0 and want in p.stdout, (mode, p.stdout + p.stderr)


@pytest.mark.gpu
def test_installed_copy_renders_from_a_cwd_outside_the_repo(installed, gpu_device, tmp_path):
    """The GPU smoke through the INSTALLED package (its own libpnr_hip.so, found relative to the installed files), run by
    a child interpreter whose cwd and sys.path know nothing of the repo: the image must be bit-identical to the one the
    in-tree package renders here (which the parity tests hold to the oracle)."""
    import torch
    import pnr_oracle as O
    from helpers import build_hip, camera_rays, oracle_cfg, small_scene
    from pointnerf2studio_amd import synthetic
    from pointnerf2studio_amd.renderer import RendererHIP
    top, site = installed
    out_file = tmp_path / "rgb.pt"
    p = _run(site, tmp_path, f"""
        import os, sys, torch
        import pointnerf2studio_amd
        assert pointnerf2studio_amd.__file__.startswith({str(site)!r}) and {ROOT!r} not in sys.path
        from pointnerf2studio_amd import _lib, synthetic
        from pointnerf2studio_amd.renderer import RendererHIP, SceneHIP, WeightsHIP, grid_hyperparameters
        assert _lib.find_library().startswith({str(site)!r})
        dev = torch.device("cuda:0")
        pts = synthetic.make_points(50000, seed=1234)
        xyz = pts["xyz"].to(dev)
        hyp = grid_hyperparameters(xyz, [0.004] * 3, [2, 2, 2], [3, 3, 3], synthetic.CHAIR_RANGES)
        scene = SceneHIP()
        scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, [3, 3, 3], [3, 3, 3], 12, 410000, True)
        scene.pack_points(xyz, *(pts[k].to(dev) for k in ("embedding", "conf", "dir", "color")))
        wh = WeightsHIP()
        wh.pack(synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1), pts["Rw2c"], dev)
        campos, camrot = synthetic.make_camera(35.0, 30.0)
        dirs = synthetic.make_rays(32, 32, campos, camrot)
        out = RendererHIP(scene, wh, SR=32, K=8).render(dirs.to(dev), campos, camrot, 2.0, 6.0)
        torch.save({{"rgb": out["rgb"].cpu(), "mask": out["ray_mask"].cpu(), "pairs": out["counters"]["pairs_valid"]}},
                   {str(out_file)!r})
        print("rendered", out["counters"]["pairs_valid"])
    """)
    assert p.returncode == 0 and "rendered" in p.stdout, p.stdout + p.stderr
    got = torch.load(out_file)
    pts = small_scene(50000)
    cfg = oracle_cfg(O, SR=32, K=8)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    campos, camrot, dirs = camera_rays(32, 32, az=35.0)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    here = RendererHIP(scene, wh, SR=32, K=8).render(dirs.to(gpu_device), campos, camrot, 2.0, 6.0)
    assert got["pairs"] == here["counters"]["pairs_valid"] > 1000
    assert torch.equal(got["mask"], here["ray_mask"].cpu()) and torch.equal(got["rgb"], here["rgb"].cpu())
