"""Host-side pieces of bench.py that need no GPU: the core / thread accounting of the CPU-baseline legs, and the lookup of
the committed PMC passes that `roofline.traffic` / `roofline_hbm.traffic` quote (only for the very workload they were
collected on)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_env_defaults_are_set_before_torch_is_imported():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")') < src.index("import torch")
    main = src[src.index("def main():"):]
    assert "HSA_ENABLE_IPC_MODE_LEGACY" not in main          # (not set late, after the runtime is up)


def test_cpu_accounting_reports_physical_cores_and_threads_separately():
    b = _bench()
    model, physical, logical = b.host_cpu()
    assert 1 <= physical <= logical == (os.cpu_count() or logical)
    quota = b.cpu_quota()
    assert quota is None or quota > 0
    allowed = len(os.sched_getaffinity(0)) if quota is None else min(len(os.sched_getaffinity(0)), max(1, int(quota)))
    assert b.cpu_threads(0) == min(physical, allowed)      # default: one per physical core the process may actually use
    assert b.cpu_threads(3) == min(3, allowed)
    assert b.cpu_threads(10 ** 6) == allowed               # never more than this process may run on at once


def test_pmc_lookup_matches_the_workload_key_only():
    b = _bench()
    key = "cfg1_chair_6m:N=6000000:K=8:SR=80:fp32:jitter=0.3:world=1"
    hit = b.pmc_traffic("k_shade_pairs", key)
    assert hit is not None and 3e9 < hit[0] < 7e9 and hit[1].startswith("profiles/")
    newest = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r"))[-1]
    assert hit[1].startswith(f"profiles/{newest}/"), hit[1]                        # the newest round's pass wins
    assert b.pmc_traffic("k_shade_pairs", key.replace("world=1", "world=8")) is None
    assert b.pmc_traffic("k_shade_pairs", key.replace("K=8", "K=12")) is None
    q = b.pmc_traffic_sum(b.QUERY_STAGE_KERNELS, key)
    assert q is not None and 5e8 < q[0] < 3e9
    names = {n.split("<")[0].split("::")[-1] for n in q[1]}
    assert {"k_select", "k_expand", "k_knn3"} <= names and "k_shade_pairs" not in names
    assert abs(sum(q[1].values()) - q[0]) < 1.0
    with open(os.path.join(ROOT, q[2])) as f:
        assert json.load(f)["workload_key"] == key
