"""Counterfactual replay: the per-joint proof behind tests/_explain.py's verdict.  Test infrastructure.

The post-processing chain is a deterministic function of its strict compares.  explain() shows that every decision with the SAME keyed
inputs and a DIFFERENT outcome on the two sides (a root flip) is a near-tie on the reference side — and then used to ASSUME that everything
else that differs is downstream of those.  This module removes the assumption: it walks the chain ONCE MORE on the REFERENCE's maps, stage
by stage, and at every decision compares the outcome with the engine side's outcome for the same keyed inputs.  Inductively the inputs
are the same (all earlier decisions agree or were forced), so every decision that still differs is a root flip BY DEFINITION: it must
pass the near-tie test (margin on the reference side < 2 x the measured deviation) and is then forced to the engine's outcome; anything
else is `unexplained`.  At the end the replayed people must be the engine's people EXACTLY (same persons, same peak in every part) —
`replay_identical` — while every VALUE along the way (centroids, scores, PAF samples) was the reference's.

  stage 1  NMS flags       nms_layer.cu:15-46     forced flag set = the engine's; peaks re-evaluated on the reference map (writeResultKernel
                                                  :50-113 restated in float32, checked bit for bit against the oracle on the reference's own flags)
  stage 2  PAF tests       rtpose.cpp:897-951     orc_connect_trace on (reference map, counterfactual peaks): every (limb, i, j) pair's
                                                  accept / count against the engine's trace row of the same indices
  stage 3  order + greedy  :144-152, 953-980      candidates in the engine's order where two near-equal reference scores are inverted
  stage 4  assembly        :983-1049              restated below (checked against the oracle's own rows on BOTH sides' unforced inputs)
  stage 5  keep            :1051-1056             count >= min_subset_cnt, score / count > min_subset_score
"""
import numpy as np

import _oracle as orc

SQRT2 = float(np.sqrt(2.0))
F = np.float32


def peaks_at(res, keys, max_peaks, num_parts):
    """writeResultKernel (nms_layer.cu:50-113) for a GIVEN flag set: keys[p] = flagged pixels (y, x) of part p in raster order.
    float32 accumulation in the kernel's (dy, dx) order; the window's row bound is `width` (sic), rows beyond the part's plane read the
    next plane.  Returns peaks [num_parts][max_peaks + 1][3]."""
    C, H, W = res.shape
    flat = res.reshape(-1)
    out = np.zeros((num_parts, max_peaks + 1, 3), np.float32)
    for p in range(num_parts):
        out[p, 0, 0] = len(keys[p])
        base = p * H * W
        for e, (py, px) in enumerate(keys[p][:max_peaks]):
            xa = ya = sa = F(0)
            for dy in range(-3, 4):
                if 0 < py + dy < W:
                    for dx in range(-3, 4):
                        if 0 < px + dx < W:
                            idx = base + (py + dy) * W + px + dx
                            sc = flat[idx] if idx < flat.size else F(0)
                            if sc > 0:
                                xa = F(xa + F(F(px + dx) * sc))
                                ya = F(ya + F(F(py + dy) * sc))
                                sa = F(sa + sc)
            with np.errstate(divide="ignore", invalid="ignore"):
                out[p, e + 1] = (xa / sa, ya / sa, res[p, py, px])
    return out


def greedy(pairs, nA, nB):
    """rtpose.cpp:956-980 on candidates that are already in pick order: [(i, j, score)] -> picks."""
    usedA, usedB, picks = set(), set(), []
    num = min(nA, nB)
    for i, j, sc in pairs:
        if len(picks) == num:
            break
        if i not in usedA and j not in usedB:
            picks.append((i, j, float(F(sc))))      # `const float score = temp[row][2]`
            usedA.add(i)
            usedB.add(j)
    return picks


def assemble(model, picks_by_limb, peaks, max_peaks, counts):
    """rtpose.cpp:983-1049 (+ the single-sided branches :843-895 / :586-608): picks_by_limb[l] = [(i, j, score)] in pick order,
    counts[p] = usable peaks of part p.  Returns rows [n][num_parts + 2]: peaks offsets per part (0 = absent), count, score."""
    num_parts, num_limbs, limb_seq, _ = orc.model_tables(model)
    coco = model == 0
    poff = 3 * (max_peaks + 1)
    pk = np.asarray(peaks, np.float32).reshape(-1)
    rows = []

    def off(part, i):
        return part * poff + i * 3 + 2

    for l in range(num_limbs):
        a, b = limb_seq[2 * l], limb_seq[2 * l + 1]
        nA, nB = counts[a], counts[b]
        if nA == 0 and nB == 0:
            continue
        if nA == 0 or nB == 0:
            part, n = (b, nB) if nA == 0 else (a, nA)
            for i in range(1, n + 1):
                o = off(part, i)
                if coco and any(r_[part] == o for r_ in rows):
                    continue
                r_ = [0.0] * (num_parts + 2)
                r_[part] = float(o)
                r_[num_parts] = 1.0
                r_[num_parts + 1] = float(pk[o])
                rows.append(r_)
            continue
        conn = [(off(a, i), off(b, j), sc) for i, j, sc in picks_by_limb.get(l, [])]
        if l == 0:
            for ia, ib, sc in conn:
                r_ = [0.0] * (num_parts + 2)
                r_[a], r_[b] = float(ia), float(ib)
                r_[num_parts] = 2.0
                r_[num_parts + 1] = float(F(pk[ia] + pk[ib])) + sc     # `peaks[a] + peaks[b] + conn`: float + float first, then + double
                rows.append(r_)
            continue
        for ia, ib, sc in conn:
            hit = 0
            for r_ in rows:
                if r_[a] == ia:
                    r_[b] = float(ib)
                    hit += 1
                    r_[num_parts] += 1
                    r_[num_parts + 1] = r_[num_parts + 1] + float(pk[ib]) + sc
            if hit == 0:
                r_ = [0.0] * (num_parts + 2)
                r_[a], r_[b] = float(ia), float(ib)
                r_[num_parts] = 2.0
                r_[num_parts + 1] = float(F(pk[ia] + pk[ib])) + sc
                rows.append(r_)
    return rows


def kept(row, num_parts, thr):
    return row[num_parts] >= thr["min_subset_cnt"] and (row[num_parts + 1] / row[num_parts]) > thr["min_subset_score"]


def _cand_by_limb(trace_cand):
    d = {}
    for row in trace_cand:
        d.setdefault(int(row[0]), {})[(int(row[1]), int(row[2]))] = row
    return d


def _order(cands):
    """ColumnCompare + std::sort (rtpose.cpp:144-152, 953): connection score descending; loop order among equals (exact ties do not occur
    on float scores of distinct pairs here — the self-checks below would notice)."""
    return sorted(cands, key=lambda t: (-t[2], t[0], t[1]))


def _rows_struct(rows, keys, max_peaks, num_parts):
    poff = 3 * (max_peaks + 1)
    out = []
    for r_ in rows:
        ks = []
        for p in range(num_parts):
            o = int(r_[p])
            if o:
                ks.append((p, keys[p][(o - p * poff - 2) // 3 - 1]))
        out.append(frozenset(ks))
    return out


def self_check(model, res, pk, keys, trace, max_peaks, thr):
    """The Python restatement of stages 3-5 on ONE side's own unforced values must reproduce the oracle's trace of that side: its greedy
    picks (conn) and its subset rows (structure, count, score, kept).  Returns a list of problems (empty = fine)."""
    num_parts, num_limbs, limb_seq, _ = orc.model_tables(model)
    counts = [min(len(k), max_peaks) for k in keys]
    cb = _cand_by_limb(trace[2])
    picks = {}
    for l in range(num_limbs):
        acc = [(i, j, float(r_[4])) for (i, j), r_ in cb.get(l, {}).items() if r_[3]]
        picks[l] = greedy(_order(acc), counts[limb_seq[2 * l]], counts[limb_seq[2 * l + 1]])
    want = {}
    for r_ in trace[3]:
        want.setdefault(int(r_[0]), []).append((int(r_[1]), int(r_[2]), float(r_[3])))
    bad = []
    for l in range(num_limbs):
        if picks[l] != want.get(l, []):
            bad.append(f"greedy restatement differs from the oracle on limb {l}")
    rows = assemble(model, picks, pk, max_peaks, counts)
    tr_rows = trace[4]
    if len(rows) != len(tr_rows):
        bad.append(f"assembly restatement: {len(rows)} rows, oracle {len(tr_rows)}")
    else:
        for a, b in zip(rows, tr_rows):
            if list(a[:num_parts + 2]) != [float(v) for v in b[:num_parts + 2]] or bool(kept(a, num_parts, thr)) != bool(b[num_parts + 2]):
                bad.append("assembly restatement differs from the oracle's rows")
                break
    return bad


def replay(model, res_r, res_e, kr, ke, mr, pk_r, pk_e, tr, te, max_peaks, net_w, net_h, disp_w, disp_h, thr, e_heat, e_paf, pmax, tol_pos_net):
    """kr / ke: flagged keys per part (raster order) of the reference / the engine; mr: the reference's NMS margins; pk_*: peaks; tr / te:
    orc.connect_trace outputs.  pmax[l]: largest PAF magnitude of limb l.  tol_pos_net: the position tolerance in net pixels.
    Returns a dict (replay_identical, forced counts, unexplained list, ...)."""
    num_parts, num_limbs, limb_seq, _ = orc.model_tables(model)
    forced = dict(nms=0, accept=0, count=0, rounding=0, inversion=0, keep=0)
    unexplained = []
    worst = 0.0

    def check(kind, margin, allow, what):
        nonlocal worst
        forced[kind] += 1
        if not (margin < allow):
            unexplained.append(f"{kind}: margin {margin:.3e} >= allowance {allow:.3e}: {what}")
        elif allow > 0:
            worst = max(worst, float(margin / allow))

    problems = self_check(model, res_r, pk_r, kr, tr, max_peaks, thr) + self_check(model, res_e, pk_e, ke, te, max_peaks, thr)
    if not np.array_equal(peaks_at(res_r, kr, max_peaks, num_parts)[:, 1:], pk_r[:, 1:], equal_nan=True):
        problems.append("peaks_at() differs from the oracle's NMS on the reference's own flags")
    # ---- stage 1: the engine's flag set on the reference's map ------------------------------------------------------------------
    for p in range(num_parts):
        for (y, x) in set(kr[p]) ^ set(ke[p]):
            loc = float(np.abs(res_e[p, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2].astype(np.float64) - res_r[p, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2]).max())
            check("nms", abs(mr[p, y, x]), 2 * loc * (1 + 1e-6) + 1e-12, f"part {p} pixel ({x},{y}): reference margin {mr[p, y, x]:+.3e}, local deviation {loc:.3e}")
    pk_c = peaks_at(res_r, ke, max_peaks, num_parts)
    counts = [min(len(k), max_peaks) for k in ke]
    e_pos = e_sc = 0.0
    for p in range(num_parts):
        n = counts[p]
        if n:
            e_pos = max(e_pos, float(np.abs(pk_c[p, 1:n + 1, :2].astype(np.float64) - pk_e[p, 1:n + 1, :2]).max()))
            e_sc = max(e_sc, float(np.abs(pk_c[p, 1:n + 1, 2].astype(np.float64) - pk_e[p, 1:n + 1, 2]).max()))
    if not e_pos <= tol_pos_net:
        unexplained.append(f"centroid: a peak on the same flag moved by {e_pos:.3e} net pixels between the reference's and the engine's map (tolerance {tol_pos_net:.3e})")
    if e_sc > e_heat * (1 + 1e-6):
        unexplained.append(f"peak-score: {e_sc:.3e} > the map deviation {e_heat:.3e}")
    # ---- stage 2: PAF tests of every pair, reference map + counterfactual peaks ---------------------------------------------------
    tc = orc.connect_trace(model, res_r, pk_c, max_peaks, net_w, net_h, disp_w, disp_h, thr)
    cc, ce = _cand_by_limb(tc[2]), _cand_by_limb(te[2])
    picks = {}
    bound_max = 0.0
    for l in range(num_limbs):
        a, b = limb_seq[2 * l], limb_seq[2 * l + 1]
        rc_, re_ = cc.get(l, {}), ce.get(l, {})
        if set(rc_) != set(re_):
            unexplained.append(f"pair-set: limb {l} evaluates {len(rc_)} pairs on the replay, {len(re_)} on the engine side (coincident peaks on one side only)")
            continue
        lst = []   # (i, j, reference score or the engine's where only the engine accepts, engine score, bound)
        for ij, rr in rc_.items():
            ee = re_[ij]
            bnd = SQRT2 * e_paf + pmax[l] * 2 * SQRT2 * e_pos / max(float(rr[8]), 1e-6)
            what = f"limb {l} pair {ij}: replay accepted {int(rr[3])} count {int(rr[5])} score {rr[4]:.6f}; engine accepted {int(ee[3])} count {int(ee[5])} score {ee[4]:.6f}"
            rounding = rr[7] < 2 * e_pos
            if rr[3] != ee[3]:
                if rounding and not rr[6] < 2 * bnd:
                    check("rounding", rr[7], 2 * e_pos, what)
                else:
                    check("accept", rr[6], 2 * bnd, what)
            elif rr[3] and rr[5] != ee[5]:
                if rounding and not rr[9] < 2 * bnd:
                    check("rounding", rr[7], 2 * e_pos, what)
                else:
                    check("count", rr[9], 2 * bnd, what)
            elif rr[3] and abs(rr[4] - ee[4]) > bnd:
                check("rounding", rr[7], 2 * e_pos, what + f"; score moved by {abs(rr[4] - ee[4]):.3e} > {bnd:.3e}")
            if ee[3]:
                bound_max = max(bound_max, bnd)
                # the score the replay orders this pair by: the reference's own — unless one of the pair's sample decisions was just forced
                # (accepted / count / a sample on another pixel): then its value IS the forced outcome's, i.e. the engine's
                same_samples = bool(rr[3]) and rr[5] == ee[5] and abs(rr[4] - ee[4]) <= bnd
                lst.append((ij[0], ij[1], float(rr[4]) if same_samples else float(ee[4]), float(ee[4]), bnd))
        # ---- stage 3: order.  The greedy result depends on the relative order of the accepted pairs only ------------------------------
        if len(lst) > 1:
            sr_ = np.array([t[2] for t in lst]); se_ = np.array([t[3] for t in lst]); bd = np.array([t[4] for t in lst])
            dr = sr_[:, None] - sr_[None]
            de = se_[:, None] - se_[None]
            inv = np.triu((dr * de < 0) | ((dr == 0) != (de == 0)), 1)
            for i, j in zip(*np.nonzero(inv)):
                check("inversion", abs(dr[i, j]), 2 * (bd[i] + bd[j]), f"limb {l}: pairs {lst[i][:2]} / {lst[j][:2]}: reference scores {sr_[i]:.6f} / {sr_[j]:.6f}, engine {se_[i]:.6f} / {se_[j]:.6f}")
        eng_order = _order([(t[0], t[1], t[3]) for t in lst])                 # the engine's order ...
        ref_score = {(t[0], t[1]): t[2] for t in lst}
        picks[l] = greedy([(i, j, ref_score[(i, j)]) for i, j, _ in eng_order], counts[a], counts[b])   # ... carrying the reference's scores
    # ---- stage 4 + 5: assembly on the reference's values, keep decisions ------------------------------------------------------------
    rows_c = assemble(model, picks, pk_c, max_peaks, counts)
    sc_, se2 = _rows_struct(rows_c, ke, max_peaks, num_parts), _rows_struct(te[4], ke, max_peaks, num_parts)
    keep_allow = 2 * (max(e_heat, e_paf) + bound_max)
    people_c, people_e = [], []
    same_rows = len(sc_) == len(se2) and all(x == y for x, y in zip(sc_, se2))
    if same_rows:
        for row, erow, ks in zip(rows_c, te[4], sc_):
            k_c, k_e = bool(kept(row, num_parts, thr)), bool(erow[num_parts + 2])
            if k_c != k_e:
                margin = abs(row[num_parts + 1] / row[num_parts] - thr["min_subset_score"]) if row[num_parts] >= thr["min_subset_cnt"] else np.inf
                check("keep", margin, keep_allow, f"person with parts {sorted(p for p, _ in ks)}: replay score/count {row[num_parts + 1] / row[num_parts]:.6f}")
            if k_e:
                people_c.append(ks)     # (the keep decision is forced to the engine's once it passed the near-tie test)
    people_e = [ks for ks, erow in zip(se2, te[4]) if erow[num_parts + 2]]
    identical = bool(same_rows and people_c == people_e and not problems)
    return dict(replay_identical=identical, replay_forced=forced, replay_forced_total=int(sum(forced.values())), replay_unexplained=len(unexplained) + len(problems),
                replay_unexplained_detail=(problems + unexplained)[:8], replay_worst_margin_over_allowance=worst, replay_people=len(people_e),
                replay_e_pos_net_px=e_pos)
