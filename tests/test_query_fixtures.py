"""Query stage on the CPU: both oracle statements (oracle/query_oracle.c and pnr_oracle.query_py) against
  * the stored drift-guard fixtures (tests/golden/query_stage.npz),
  * a hand-derived case whose expected lists come from reading query_worldcoords.cu (tests/query_cases.py),
  * each other over a hypothesis sweep of grid sizes, P, K, kernel sizes and adversarial geometry (points on voxel
    faces, samples at zero distance of a point, candidates at exactly radius^2).
All comparisons are exact (int32 lists, float32 positions, int8 masks)."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from query_cases import hand_case, stored_case_args, stored_cases

_G, _NAMES = stored_cases()


@pytest.mark.parametrize("name", _NAMES)
def test_c_oracle_reproduces_stored_fixture(oracle, name):
    args, want, stats = stored_case_args(oracle, _G, name)
    pidx, loc, mask, st_ = oracle.query(*args)
    assert torch.equal(pidx, want[0]) and torch.equal(loc, want[1]) and torch.equal(mask, want[2])
    got = [st_[k] for k in ("occupied_voxels", "rays_hit", "rays_kept", "valid_samples", "valid_pairs",
                            "selected_samples")]
    assert got == stats.tolist()


@pytest.mark.parametrize("name", [n for n in _NAMES if "sr8_" in n])     # the SR = 80 cases take the Python loops ~10 s
def test_python_statement_reproduces_stored_fixture(oracle, name):
    args, want, _ = stored_case_args(oracle, _G, name)
    pidx, loc, mask = oracle.query_py(*args)
    assert torch.equal(pidx, want[0]) and torch.equal(loc, want[1]) and torch.equal(mask, want[2])


@pytest.mark.parametrize("compat", [True, False])
def test_hand_derived_case(oracle, compat):
    """Expected values derived by hand from query_worldcoords.cu (see tests/query_cases.py): a voxel with more than P
    points, a replace-the-farthest step with a tie, the voxel-0 drop, a hit ray without neighbours, a partial list."""
    args, want, stats = hand_case(compat)
    for impl in ("c", "py"):
        if impl == "c":
            pidx, loc, mask, st_ = oracle.query(*args)
            assert st_["rays_hit"] == stats["rays_hit"] and st_["rays_kept"] == stats["rays_kept"]
            assert st_["occupied_voxels"] == stats["occupied_voxels"]
        else:
            pidx, loc, mask = oracle.query_py(*args)
        assert torch.equal(mask, want[2]), impl
        assert torch.equal(pidx, want[0]), (impl, pidx)
        assert torch.equal(loc, want[1]), impl


# ---- differential sweep ------------------------------------------------------------------------------------------
@st.composite
def _problem(draw):
    dims = [draw(st.integers(1, 6)) for _ in range(3)]
    vox = draw(st.sampled_from([0.25, 0.5, 1.0, 0.1]))      # 0.1 is not a dyadic rational: rounding at the faces
    ks = draw(st.sampled_from([1, 3, 3, 5]))
    P = draw(st.integers(1, 5))
    K = draw(st.integers(1, 9))
    SR = draw(st.integers(1, 6))
    D = draw(st.integers(1, 10))
    n = draw(st.integers(1, 40))
    R = draw(st.integers(1, 5))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    compat = draw(st.booleans())
    radius = draw(st.sampled_from([0.0, 0.5, 1.0, 2.5])) * vox     # 0 = no limit (cu:272)
    return dims, vox, ks, P, K, SR, D, n, R, seed, compat, radius


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(_problem())
def test_c_and_python_statements_agree_on_adversarial_geometry(oracle, prob):
    dims, vox, ks, P, K, SR, D, n, R, seed, compat, radius = prob
    rng = np.random.RandomState(seed)
    f32 = np.float32
    ext = np.array(dims, dtype=f32) * f32(vox)
    # points: random, snapped onto voxel faces (exact multiples of the voxel size), and duplicates
    xyz = (rng.rand(n, 3).astype(f32) * (ext + f32(2 * vox)) - f32(vox)).astype(f32)       # some outside the grid
    snap = rng.rand(n, 3) < 0.3
    xyz[snap] = (np.round(xyz[snap] / f32(vox)) * f32(vox)).astype(f32)
    if n > 3:
        xyz[n // 2] = xyz[0]                                                              # coincident points
    # ray samples: random, ON points (d2 == 0), and at an axis offset of exactly `radius` from a point (d2 == r^2 when
    # the offset is representable), plus positions on faces and outside the grid
    pos = (rng.rand(R, D, 3).astype(f32) * (ext + f32(2 * vox)) - f32(vox)).astype(f32)
    for r in range(R):
        for j in range(D):
            u = rng.rand()
            p = xyz[rng.randint(n)]
            if u < 0.25:
                pos[r, j] = p
            elif u < 0.45 and radius > 0:
                off = np.zeros(3, dtype=f32)
                off[rng.randint(3)] = f32(radius) * (1 if rng.rand() < 0.5 else -1)
                pos[r, j] = (p + off).astype(f32)
            elif u < 0.6:
                pos[r, j] = (np.round(pos[r, j] / f32(vox)) * f32(vox)).astype(f32)
    args = (torch.from_numpy(pos)[None], torch.from_numpy(xyz)[None], [ks] * 3, [ks] * 3, SR, K,
            np.array(dims, dtype=np.int32), 10000, P, float(radius), torch.zeros(6), np.array([vox] * 3, dtype=f32),
            compat)
    p1, l1, m1, _ = oracle.query(*args)
    p2, l2, m2 = oracle.query_py(*args)
    assert torch.equal(m1, m2)
    assert torch.equal(p1, p2)
    assert torch.equal(l1, l2)
