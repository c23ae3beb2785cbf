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
    assert r["joints_structural"] == 1 + 1 + 18 == len(r["structural"]) and r["numeric_out_of_tol"] == 0
    assert sorted(e[0] for e in r["structural"]) == ["both"] + ["engine"] * 19
    v = _parity.verdict(_parity.merge([r]))
    assert v.startswith("FAIL") and "not traced" in v                     # structural differences nobody explained cannot pass
    assert _parity.verdict(_parity.merge([r]), explained=20).startswith("numeric pass")


def test_a_joint_that_moves_out_of_tolerance_is_a_fail_not_a_structural_difference():
    """VERDICT r3 weak #1a: max_dx_px used to be taken over joints already within tol_px, so it could never exceed it.  Joints are now
    paired within pair_px (3 px); without the maps (tests/_explain.py) every pair outside the tolerance is a numeric FAIL."""
    a = _people(3)
    b = a.copy()
    b[1, 4, 0] += 1.3             # the same peak, 1.3 px away: outside +-1 px
    r = _parity.people_parity(a, b)
    assert r["numeric_out_of_tol"] == 1 and r["joints_structural"] == 0 and abs(r["max_dx_px"] - 1.3) < 1e-4
    assert r["people_matched"] == 2 and r["joints_matched"] == 3 * 18 - 1
    assert _parity.verdict(_parity.merge([r])).startswith("FAIL: 1 paired joint")
    b = a.copy()
    b[2, 9, 2] += 3e-3            # score outside +-1e-3
    r = _parity.people_parity(a, b)
    assert r["numeric_out_of_tol"] == 1 and _parity.verdict(_parity.merge([r])).startswith("FAIL")
    b = a.copy()
    b[0, 2, 1] += 1.95            # the maximum moved to the NEXT net pixel: numeric until _explain shows it is two different maxima
    r = _parity.people_parity(a, b)
    assert r["numeric_out_of_tol"] == 1 and r["joints_structural"] == 0 and len(r["out_of_tol"]) == 1
    _parity.reclassify(r, [True])
    assert r["numeric_out_of_tol"] == 0 and r["joints_structural"] == 1 == r["adjacent_pixel_flips"] and r["structural"][0][0] == "both" and r["max_dy_px"] <= 1.0
    assert _parity.verdict(_parity.merge([r]), explained=1).startswith("numeric pass")
    assert _parity.verdict(_parity.merge([r]), map_err=2e-3, explained=1).startswith("FAIL: final maps")
    b[0, 2, 1] += 3.0             # farther than the pairing radius: another peak
    r = _parity.people_parity(a, b)
    assert r["numeric_out_of_tol"] == 0 and r["joints_structural"] == 1


def test_empty_sides():
    a = _people(2)
    z = np.zeros((0, 18, 3), np.float32)
    assert _parity.people_parity(z, z)["people_matched"] == 0
    r = _parity.people_parity(a, z)
    assert r["people_ref"] == 0 and r["joints_structural"] == 36 == len(r["structural"])
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


# ---------------------------------------------------------------------------------------------------------------------------------
# tests/_explain.py: every structural difference traced to a near-tie decision — on the CPU, with the oracle standing in for the engine
# (the engine's post-processing is bit-exact on its own maps, so "engine" = the oracle chain on maps that deviate by <= 7e-4 of the maximum)
# ---------------------------------------------------------------------------------------------------------------------------------
import pytest  # noqa: E402

import _explain  # noqa: E402
import _oracle as orc  # noqa: E402
import _synth  # noqa: E402

GEO = {0: (656, 368, 18, 57, 64), 1: (496, 368, 15, 44, 20)}


def _chain(model, low, thr):
    W, H, parts, _, maxp = GEO[model]
    res = orc.imresize(low, W, H)[0]
    pk = orc.nms(res, parts, maxp, thr["nms_threshold"])
    n, j = orc.connect(model, res, pk, maxp, W, H, 1280, 720, thr)
    return res, pk, n, j


@pytest.mark.parametrize("model,seed,white", [(0, 1, True), (0, 2, False), (1, 3, True)])
def test_every_flip_of_a_sub_tolerance_perturbation_is_explained(model, seed, white):
    """Noise maps (hundreds of maxima, dozens of people) perturbed by 7e-4 of their maximum — white noise and a smooth field, the two
    extremes of what a conv stack's rounding error can look like: people differ structurally, every joint that is the same peak stays
    inside the tolerance, and EVERY decision that differs (NMS compares, PAF samples against their threshold, sample coordinates at a
    rounding boundary, order inversions) has a reference-side margin below twice the measured deviation."""
    W, H, _parts, C, maxp = GEO[model]
    thr = orc.default_thresholds(model)
    base = _synth.smooth_field(C, H // 8, W // 8, seed=seed, scale=1.0)[None]
    d = np.random.RandomState(100 + seed).uniform(-1, 1, base.shape).astype(np.float32) if white else _synth.smooth_field(C, H // 8, W // 8, seed=seed + 77)[None]
    pert = (base + 7e-4 * d / np.abs(d).max()).astype(np.float32)
    res_r, _, nr, jr = _chain(model, base, thr)
    res_e, _, ne, je = _chain(model, pert, thr)
    rep = _parity.people_parity(je[:ne], jr[:nr])
    ex = _explain.explain(model, res_r, res_e, maxp, W, H, 1280, 720, thr, rep["structural"], out_of_tol=rep["out_of_tol"])
    _parity.reclassify(rep, ex["out_of_tol_is_flip"])
    print(rep["people_ref"], rep["people_matched"], rep["joints_structural"], rep["adjacent_pixel_flips"], ex["root_flips"], ex["worst_margin_over_allowance"], ex["attribution"])
    assert nr > 20 and rep["numeric_out_of_tol"] == 0
    assert ex["unexplained"] == 0, ex["unexplained_detail"]
    assert ex["root_flips"]["nms"] > 0 and ex["worst_margin_over_allowance"] < 0.6   # (a margin at 60 % of its allowance would mean the bound has no slack left)
    assert ex["structural_explained"] == rep["joints_structural"] == ex["joints_structural"]
    # the per-joint proof (tests/_replay.py): the reference side replayed with the engine's outcome forced at every decision that differs for
    # the same inputs — each of them a checked near-tie — gives the engine's people exactly; nothing is "assumed downstream"
    print("replay:", ex["replay_forced"], ex["replay_worst_margin_over_allowance"], ex["replay_unexplained_detail"])
    assert ex["replay_identical"] and ex["replay_unexplained"] == 0 and ex["replay_people"] == ne
    assert ex["replay_forced"]["nms"] == ex["root_flips"]["nms"] and ex["replay_forced_total"] >= ex["replay_forced"]["nms"]
    assert ex["replay_worst_margin_over_allowance"] < 0.8
    assert _parity.verdict(_parity.merge([rep]), map_err=7e-4, explained=ex["structural_explained"]).startswith(("pass", "numeric pass"))


def test_a_difference_that_is_not_a_near_tie_is_not_explained():
    """Negative controls: (a) a peak that exists on one side only although the other side's maximum is nowhere near a tie, (b) a deviation
    well outside the tolerance, (c) structural differences without any flipped decision."""
    model = 0
    W, H, _parts, C, maxp = GEO[model]
    thr = orc.default_thresholds(model)
    base = _synth.smooth_field(C, H // 8, W // 8, seed=5, scale=1.0)[None]
    res_r, _pk_r, nr, jr = _chain(model, base, thr)
    # (a) knock ONE clear maximum out of the engine-side maps (a bug that loses a peak): the deviation elsewhere stays tiny
    broken = base.copy()
    p = 3
    y, x = np.unravel_index(np.argmax(base[0, p]), base[0, p].shape)
    broken[0, p, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2] *= 0.5
    res_e, _, ne, je = _chain(model, broken, thr)
    rep = _parity.people_parity(je[:ne], jr[:nr])
    ex = _explain.explain(model, res_r, res_e, maxp, W, H, 1280, 720, thr, rep["structural"])
    # the deviation this "bug" causes is huge, so margins alone would pass under 2 e_map — but the peak scores moved out of the confidence tolerance
    assert ex["unexplained"] > 0 and any(d_.startswith(("peak-score", "map-deviation", "nms")) for d_ in ex["unexplained_detail"])
    assert ex["structural_explained"] == 0
    # (the replay's allowances are multiples of the same measured deviation — huge here — so ITS near-tie tests pass; what fails the verdict is
    # the deviation guard above: a replay can only be trusted inside the tolerance, and explain() demands both)
    # (b) uniform deviation of 5e-3: outside +-1e-3
    res_e2, _, ne2, je2 = _chain(model, (base * np.float32(1.005)).astype(np.float32), thr)
    rep2 = _parity.people_parity(je2[:ne2], jr[:nr])
    ex2 = _explain.explain(model, res_r, res_e2, maxp, W, H, 1280, 720, thr, rep2["structural"])
    assert rep2["max_dc"] > 1e-3 and ex2["unexplained"] > 0
    assert _parity.verdict(_parity.merge([rep2]), explained=ex2["structural_explained"]).startswith("FAIL")
    # (c) people differ (a person dropped after the fact) while no decision differs: nothing explains it
    ex3 = _explain.explain(model, res_r, res_r, maxp, W, H, 1280, 720, thr, [("ref", 0, 1, float(jr[0, 1, 0]), float(jr[0, 1, 1]), float(jr[0, 1, 2]))])
    assert ex3["root_flips_total"] == 0 and ex3["unexplained"] == 0 and ex3["joints_structural"] == 1 and ex3["structural_explained"] == 0
    assert ex3["replay_identical"] and ex3["replay_forced_total"] == 0   # (identical maps replay identically: the claimed difference has no cause in the chain)


class _FakeEngine:
    """bench.parity_report / structured_parity on the CPU: an 'engine' whose conv stack returns the reference maps plus a deviation
    field and whose post-processing is the oracle's (= what the GPU tests prove the HIP chain to be, bit for bit)."""

    def __init__(self, model, ref, dev_rel):
        self.model, self.ref, self.dev = model, ref, dev_rel
        self.W, self.H, self.num_parts, self.heat_channels, self.max_peaks = GEO[model]
        self.low_h, self.low_w, self.N = self.H // 8, self.W // 8, ref.shape[0]
        self._q = []

    def get_thresholds(self):
        return orc.default_thresholds(self.model)

    def forward_heatmaps(self, x):
        return (self.ref + self.dev * np.abs(self.ref).max()).astype(np.float32)

    def _post(self, low):
        _, pk, n, j = _chain(self.model, low, self.get_thresholds())
        return pk, j[:n].copy(), n

    def submit(self, x, tag=0):
        self._q.append(tag)

    def flush(self):
        pass

    def collect(self):
        _, j, n = self._post(self.forward_heatmaps(None))
        return self._q.pop(0), n, j

    def post_from_lowres(self, low):
        return self._post(np.ascontiguousarray(low, np.float32))


@pytest.mark.parametrize("model", [0, 1])
def test_bench_parity_report_with_structured_leg_on_a_stand_in_engine(model):
    b = _bench()
    W, H, _parts, C, _maxp = GEO[model]
    ref = (_synth.smooth_field(C, H // 8, W // 8, seed=9, scale=1.0)[None] * np.float32(4.0)).astype(np.float32)   # a map maximum of 4, like the synthetic network's
    dev = (np.random.RandomState(3).uniform(-1, 1, ref.shape) * 6.5e-4).astype(np.float32)
    eng = _FakeEngine(model, ref, dev)
    rep = b.parity_report(eng, [(None, ref, 0.0)], "coco" if model == 0 else "mpi", 1, 0.3)
    print({k: v for k, v in rep.items() if k not in ("explain", "structured")}, rep["explain"]["root_flips"], rep["structured"]["verdict"])
    assert rep["verdict"].startswith(("pass", "numeric pass")), rep["verdict"]
    assert rep["numeric_out_of_tol"] == 0 and rep["structural_explained"] == rep["joints_structural"] and rep["explain"]["unexplained"] == 0
    assert 5e-4 < rep["map_max_err"] <= 1e-3 and rep["post_on_engine_maps_bit_exact"]
    st = rep["structured"]
    assert st["verdict"].startswith("pass"), st
    assert all(c["identical_within_one_net_pixel"] and c["numeric_out_of_tol"] == 0 for c in st["cases"].values())
    assert [st["cases"][k]["people_ref"] for k in ("P1", "P5", "P20")] == [st["cases"][k]["people_engine"] for k in ("P1", "P5", "P20")]
    assert st["cases"]["P5"]["people_ref"] >= 4 and st["cases"]["P5"]["people_matched"] >= st["cases"]["P5"]["people_ref"] - 2
    # an engine whose deviation is 10x the tolerance must FAIL both legs
    bad = b.parity_report(_FakeEngine(model, ref, dev * 10), [(None, ref, 0.0)], "coco" if model == 0 else "mpi", 1, 0.3)
    assert bad["verdict"].startswith("FAIL") and bad["structured"]["verdict"].startswith("FAIL")
