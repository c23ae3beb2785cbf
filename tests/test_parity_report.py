"""tests/_parity.py (the set-based people comparison behind bench.py's `parity` dict) on constructed cases."""
import numpy as np

import _parity


def _people(n, P=18, seed=0):
    rs = np.random.RandomState(seed)
    j = np.zeros((n, P, 3), np.float32)
    j[..., 0] = rs.uniform(10, 1270, (n, P))
    j[..., 1] = rs.uniform(10, 710, (n, P))
    j[..., 2] = rs.uniform(0.2, 0.9, (n, P))
    return j


def test_identical_sets_in_any_order_match_fully():
    a = _people(7)
    b = a[::-1].copy()
    r = _parity.people_parity(a, b)
    assert r["people_matched"] == 7 and r["joints_matched"] == r["joints_ref"] == 7 * 18 and r["joints_structural"] == 0
    assert r["max_dx_px"] == r["max_dy_px"] == r["max_dc"] == 0.0


def test_numeric_deviation_inside_and_outside_the_tolerance():
    a = _people(3)
    b = a.copy()
    b[1, 4, 0] += 0.4
    b[2, 5, 2] += 5e-4
    r = _parity.people_parity(a, b)
    assert r["people_matched"] == 3 and abs(r["max_dx_px"] - 0.4) < 1e-4 and abs(r["max_dc"] - 5e-4) < 1e-6
    b[2, 5, 2] += 2e-3           # score outside 1e-3: the person is paired, the joint is not matched, nothing is structural
    r = _parity.people_parity(a, b)
    assert r["people_matched"] == 2 and r["joints_matched"] == 3 * 18 - 1 and r["joints_structural"] == 0 and r["max_dc"] > 1e-3
    r = _parity.people_parity(a, b, c_norm=4.0)   # the same deviation on maps with a maximum of 4
    assert r["people_matched"] == 3


def test_structural_differences_are_counted_not_averaged_away():
    a = _people(4)
    b = a.copy()
    b[0, 3] = 0                   # a part missing on one side
    b[1, 7, :2] += 25.0           # a different peak
    r = _parity.people_parity(a, b[:3])   # and one person missing altogether
    assert r["people_engine"] == 4 and r["people_ref"] == 3 and r["people_matched"] == 1
    assert r["joints_structural"] == 1 + 1 + 18 and r["max_dx_px"] <= 1.0


def test_empty_sides():
    a = _people(2)
    z = np.zeros((0, 18, 3), np.float32)
    assert _parity.people_parity(z, z)["people_matched"] == 0
    r = _parity.people_parity(a, z)
    assert r["people_ref"] == 0 and r["joints_structural"] == 36
    m = _parity.merge([_parity.people_parity(a, a), _parity.people_parity(a, z)])
    assert m["frames"] == 2 and m["people_matched"] == 2 and m["people_engine"] == 4


def _bench():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rtp_bench_cpu", os.path.join(root, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_block_never_prints_a_fantasy():
    """bench.py's `roofline` object: a per-launch timing that is implausible against the same kernel timed alone (round 2's driver run:
    1.02e9 ms per launch from cross-XCD clock stamps -> frac 9.5e-12) is replaced by the solo timing and says so; frac stays in (0, 1]."""
    b = _bench()
    flops, peak, solo = 24.226e9, 2.5e15, 0.0255
    good = b.roofline_block(0.0307 * 320 + 0.0513 * 480, 800, flops, {1: (0.0307 * 320, 320), 2: (0.0513 * 480, 480)}, solo, peak)
    assert 0.2 < good["frac"] < 0.25 and "FALLBACK" not in good["how"] and set(good["by_mfma_passes"]) == {"1", "2"}
    assert abs(good["by_mfma_passes"]["2"]["ms_per_launch"] - 0.0513) < 1e-9 and 0.3 < good["executed"]["frac_of_peak"] < 0.5
    bad = b.roofline_block(1.02e9 * 800, 800, flops, {1: (1.02e9 * 320, 320), 2: (1.02e9 * 480, 480)}, solo, peak)   # the round-2 driver record
    assert "FALLBACK" in bad["how"] and bad["ms_per_launch"] == solo and 0.3 < bad["frac"] < 0.45 and "by_mfma_passes" not in bad
    none = b.roofline_block(0.0, 0, flops, {}, solo, peak)                                                            # nothing harvested
    assert "FALLBACK" in none["how"] and 0 < none["frac"] <= 1
    fast = b.roofline_block(1e-6 * 10, 10, flops, {1: (1e-5, 10)}, solo, peak)                                       # absurdly fast
    assert "FALLBACK" in fast["how"] and 0 < fast["frac"] <= 1
