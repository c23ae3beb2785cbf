"""Trace every structural difference between the engine's and the reference's people to the DECISION that flipped, and check that
the flip is a near-tie on the reference side.  Test infrastructure + bench.py's `parity` dict; numpy + the oracle.

The post-processing chain is a deterministic function of its decisions (all of them strict compares in the reference):

  NMS        nms_register_kernel, nms_layer.cu:15-46      v > threshold and v > each of 8 neighbours      -> which pixels are peaks
  PAF test   rtpose.cpp:610-640 / 897-951                 each of 10 samples > inter_threshold; count > inter_min_above;
                                                          roundf() of the sample coordinates              -> which (A, B) pairs are candidates
  order      ColumnCompare + std::sort, rtpose.cpp:144-152, 953                                           -> greedy pick order
  keep       rtpose.cpp:1051-1056                         count >= min_subset_cnt, score / count > min_subset_score

Both sides are replayed by the oracle WITH its decision trace (orc_connect_trace): the reference side on the reference's resized
maps, the engine side on the resize of the engine's maps (the engine's own post-processing is bit-exact on those:
`post_on_engine_maps_bit_exact`).  Peaks are keyed by the integer pixel of their maximum, so the re-numbering a flip causes does not
matter.  A decision with the SAME keyed inputs on both sides and a different outcome is a root flip; it is EXPLAINED if its margin on
the reference side is below 2x the deviation measured between the two sides (e_map = max |resized_engine - resized_ref|, e_pos =
max centroid shift of a common peak):

  NMS flip            |min(v - thr, v - max 8-neighbour)| < 2 x the largest deviation in the pixel's 3x3 neighbourhood (local, not the map's maximum)
  sample flip         |sample - inter_threshold| of the sample(s) that must cross < 2 b,  b = sqrt2 e_paf + |PAF|max 2 sqrt2 e_pos / |AB|
                      (the sample is unit(AB) . PAF(pixel): the map moves by e_paf, the unit vector by |d(AB)| / |AB|)
  rounding flip       a sample coordinate within 2 e_pos of the .5 where roundf() changes pixel: the sample reads another pixel
  order inversion     |score_a - score_b| < 2 (b_a + b_b)
  keep flip           |score / count - min_subset_score| < 2 (e_map + max b)

Everything else that differs (a pair with a flipped endpoint, a later greedy pick, a person assembled differently) is downstream of
root flips.  If every root flip is explained — and nothing differs without one — every structural joint is explained."""
import numpy as np

import _oracle as orc

SQRT2 = float(np.sqrt(2.0))


def nms_margin(res, num_parts, thr):
    """Signed margin of nms_register_kernel's test per pixel, float64 [parts][H][W]: > 0 <=> flagged.  Border pixels: -inf."""
    v = np.asarray(res[:num_parts], np.float64)
    m = np.full(v.shape, -np.inf)
    c = v[:, 1:-1, 1:-1]
    H, W = v.shape[1:]
    nb = None
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            if dy == 1 and dx == 1:
                continue
            s = v[:, dy:H - 2 + dy, dx:W - 2 + dx]
            nb = s if nb is None else np.maximum(nb, s)
    m[:, 1:-1, 1:-1] = np.minimum(c - thr, c - nb)
    return m


def _keys(margin, max_peaks):
    """Per part: flagged pixels in raster order (the order the exclusive scan numbers them in), as (y, x) tuples."""
    out = []
    for p in range(margin.shape[0]):
        ys, xs = np.nonzero(margin[p] > 0)
        out.append(list(zip(ys.tolist(), xs.tolist())))
    return out


def explain(model, res_r, res_e, max_peaks, net_w, net_h, disp_w, disp_h, thr, structural=(), tol_px=1.0, tol_c=1e-3, c_norm=1.0, out_of_tol=()):
    """res_r / res_e: resized maps [C][net_h][net_w] of the reference / the engine.  thr: the engine's threshold dict.
    structural: people_parity()['structural'] of the frame (optional, for the per-joint attribution).  tol_px (display pixels) /
    tol_c (on scores / c_norm): every peak that is the same integer maximum on both sides must have its centroid and score inside
    the tolerance — that is the numeric half of the parity claim, asserted here for ALL peaks, not only those that ended up in people.
    out_of_tol: people_parity()['out_of_tol'] — joint pairs closer than the pairing radius but outside the tolerance; the result's
    `out_of_tol_is_flip` says for each whether the two joints sit on DIFFERENT integer maxima one of which only one side has (a flipped
    NMS compare between neighbouring pixels: structural, its margin is asserted with the other NMS flips) — see _parity.reclassify.
    Returns a dict."""
    num_parts, num_limbs, limb_seq, map_idx = orc.model_tables(model)
    res_r = np.ascontiguousarray(res_r, np.float32).reshape(-1, net_h, net_w)
    res_e = np.ascontiguousarray(res_e, np.float32).reshape(-1, net_h, net_w)
    nthr = thr["nms_threshold"]
    dmap = np.abs(res_e.astype(np.float64) - res_r.astype(np.float64))
    e_heat = float(dmap[:num_parts].max())
    e_paf = float(dmap[num_parts + 1:].max()) if dmap.shape[0] > num_parts + 1 else e_heat
    e_map = max(e_heat, e_paf)
    out = dict(e_map=e_map, e_heat=e_heat, e_paf=e_paf)
    unexplained = []
    # the allowances below are multiples of the MEASURED deviation, so the deviation itself must be inside the tolerance: the low-res maps
    # within tol_c of their maximum (checked by the caller on the low-res maps), hence the bicubic resize of them within 1.375^2 tol_c
    # (sum of |cubic weights| <= 1.375 per axis, imresize_layer.cu:9-18)
    if e_map / c_norm > tol_c * 1.375 ** 2:
        unexplained.append(("map-deviation", e_map / c_norm, tol_c * 1.375 ** 2, "the resized maps deviate by more than the tolerance allows: flips are not near-ties of a sub-tolerance deviation"))
    roots = dict(nms=0, cap=0, accept=0, count=0, rounding=0, inversion=0, keep=0)
    worst = 0.0   # largest margin / allowance ratio of an explained root flip (< 1)

    def check(kind, margin, allow, what):
        nonlocal worst
        roots[kind] += 1
        if not (margin < allow):
            unexplained.append((kind, float(margin), float(allow), what))
        elif allow > 0:
            worst = max(worst, float(margin / allow))

    # ---- NMS ------------------------------------------------------------------------------------------------------------------
    mr, me = nms_margin(res_r, num_parts, nthr), nms_margin(res_e, num_parts, nthr)
    kr, ke = _keys(mr, max_peaks), _keys(me, max_peaks)
    pk_r = orc.nms(res_r, num_parts, max_peaks, nthr)
    pk_e = orc.nms(res_e, num_parts, max_peaks, nthr)
    for p in range(num_parts):
        assert int(pk_r[p, 0, 0]) == len(kr[p]) and int(pk_e[p, 0, 0]) == len(ke[p]), "flag replay disagrees with the oracle's NMS"
    flipped = [set() for _ in range(num_parts)]   # keys that are not the same usable peak on both sides
    for p in range(num_parts):
        sr, se = set(kr[p]), set(ke[p])
        for (y, x) in sr ^ se:
            flipped[p].add((y, x))
            # LOCAL allowance: the margin is min(v - thr, v - max 8-neighbour); each term moves by at most twice the largest
            # deviation inside the pixel's 3x3 neighbourhood, so a true flip has |reference margin| <= 2 x that — the global maximum is not needed
            loc = float(dmap[p, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2].max())
            check("nms", abs(mr[p, y, x]), 2 * loc * (1 + 1e-6) + 1e-12, f"part {p} pixel ({x},{y}): reference margin {mr[p, y, x]:+.3e}, engine {me[p, y, x]:+.3e}, local deviation {loc:.3e}")
        # a maximum both sides have, inside the max_peaks cap on one side only: an earlier flip of this part moved its ordinal
        in_r, in_e = set(kr[p][:max_peaks]), set(ke[p][:max_peaks])
        for k in (sr & se):
            if (k in in_r) != (k in in_e):
                flipped[p].add(k)
                first = min(flipped[p] - {k}, default=None)
                check("cap", 0.0 if (first is not None and first < k) else 1.0, 0.5, f"part {p} pixel {k}: inside the max_peaks cap on one side only")
            elif k not in in_r:
                flipped[p].add(k)     # beyond the cap on both sides: nobody uses it
    idx_r = [{k: i + 1 for i, k in enumerate(kr[p][:max_peaks])} for p in range(num_parts)]
    idx_e = [{k: i + 1 for i, k in enumerate(ke[p][:max_peaks])} for p in range(num_parts)]
    # centroid / score deviation of the common peaks
    e_pos = e_sc = 0.0
    for p in range(num_parts):
        for k, i in idx_r[p].items():
            if k in flipped[p]:
                continue
            j = idx_e[p][k]
            e_pos = max(e_pos, float(np.abs(pk_r[p, i, :2].astype(np.float64) - pk_e[p, j, :2]).max()))
            e_sc = max(e_sc, abs(float(pk_r[p, i, 2]) - float(pk_e[p, j, 2])))
    out.update(e_pos_net_px=e_pos, e_peak_score=e_sc, peaks_ref=sum(len(k) for k in kr), peaks_engine=sum(len(k) for k in ke))
    if e_sc > e_heat * (1 + 1e-6):
        unexplained.append(("peak-score", e_sc, e_heat, "a common peak's score moved by more than the map did"))
    e_pos_disp = e_pos * max(disp_w / net_w, disp_h / net_h)
    out["e_pos_display_px"] = e_pos_disp
    if e_pos_disp > tol_px:
        unexplained.append(("centroid", e_pos_disp, tol_px, "a peak on the same integer maximum moved by more than the position tolerance"))
    if e_sc / c_norm > tol_c:
        unexplained.append(("peak-score", e_sc / c_norm, tol_c, "a peak on the same integer maximum changed its score by more than the confidence tolerance"))

    # ---- connect --------------------------------------------------------------------------------------------------------------
    tr = orc.connect_trace(model, res_r, pk_r, max_peaks, net_w, net_h, disp_w, disp_h, thr)
    te = orc.connect_trace(model, res_e, pk_e, max_peaks, net_w, net_h, disp_w, disp_h, thr)
    out.update(people_ref=tr[0], people_engine=te[0])

    def keyed(cand, keys):
        d = {}
        for row in cand:
            l = int(row[0])
            a, b = limb_seq[2 * l], limb_seq[2 * l + 1]
            d[(l, keys[a][int(row[1]) - 1], keys[b][int(row[2]) - 1])] = row
        return d

    cr, ce = keyed(tr[2], kr), keyed(te[2], ke)
    pmax = {}
    for l in range(num_limbs):
        cx, cy = map_idx[2 * l], map_idx[2 * l + 1]
        pmax[l] = float(max(np.sqrt(res_r[cx].astype(np.float64) ** 2 + res_r[cy].astype(np.float64) ** 2).max(),
                            np.sqrt(res_e[cx].astype(np.float64) ** 2 + res_e[cy].astype(np.float64) ** 2).max()))
    bound_max = 0.0
    limb_roots = {}     # limb -> set of peak keys (part, key) that take part in one of its root flips
    stable = {l: [] for l in range(num_limbs)}   # per limb: (score_ref, score_engine, bound, key) of pairs accepted on both sides with the same samples

    def touch(l, ka, kb):
        limb_roots.setdefault(l, set()).update({(limb_seq[2 * l], ka), (limb_seq[2 * l + 1], kb)})

    for key, rr in cr.items():
        l, ka, kb = key
        a, b = limb_seq[2 * l], limb_seq[2 * l + 1]
        if ka in flipped[a] or kb in flipped[b]:
            continue                                   # downstream of an NMS flip
        re_ = ce.get(key)
        if re_ is None:                                # (norm_vec < 1e-6 on one side only: the two peaks coincide; cannot happen for distinct parts' maxima here)
            unexplained.append(("pair-missing", 0.0, 0.0, f"limb {l} pair {ka}-{kb} has no trace row on the engine side"))
            continue
        norm = max(float(rr[8]), 1e-6)
        bnd = SQRT2 * e_paf + pmax[l] * 2 * SQRT2 * e_pos / norm
        bound_max = max(bound_max, bnd if rr[3] or re_[3] else 0.0)
        rounding = rr[7] < 2 * e_pos
        what = f"limb {l} pair {ka}-{kb}: reference accepted {int(rr[3])} count {int(rr[5])} score {rr[4]:.6f}; engine accepted {int(re_[3])} count {int(re_[5])} score {re_[4]:.6f}"
        if rr[3] != re_[3]:
            touch(l, ka, kb)
            if rounding and not rr[6] < 2 * bnd:
                check("rounding", rr[7], 2 * e_pos, what)
            else:
                check("accept", rr[6], 2 * bnd, what + f"; sample-to-threshold margin {rr[6]:.3e}")
        elif rr[3]:
            if rr[5] != re_[5]:
                touch(l, ka, kb)
                if rounding and not rr[9] < 2 * bnd:
                    check("rounding", rr[7], 2 * e_pos, what)
                else:
                    check("count", rr[9], 2 * bnd, what + f"; nearest sample to the threshold {rr[9]:.3e}")
            elif abs(rr[4] - re_[4]) > bnd:
                touch(l, ka, kb)
                check("rounding", rr[7], 2 * e_pos, what + f"; score moved by {abs(rr[4] - re_[4]):.3e} > {bnd:.3e} with round margin {rr[7]:.3e}")
            else:
                stable[l].append((float(rr[4]), float(re_[4]), bnd, (ka, kb)))
    for key in ce:
        l, ka, kb = key
        a, b = limb_seq[2 * l], limb_seq[2 * l + 1]
        if key not in cr and not (ka in flipped[a] or kb in flipped[b]):
            unexplained.append(("pair-missing", 0.0, 0.0, f"limb {l} pair {ka}-{kb} has no trace row on the reference side"))
    # order inversions among the pairs both sides accept on the same samples
    for l, lst in stable.items():
        if len(lst) < 2:
            continue
        sr_ = np.array([t[0] for t in lst]); se_ = np.array([t[1] for t in lst]); bd = np.array([t[2] for t in lst])
        dr = sr_[:, None] - sr_[None]
        de = se_[:, None] - se_[None]
        inv = np.triu((dr * de < 0) | ((dr == 0) != (de == 0)), 1)
        for i, j in zip(*np.nonzero(inv)):
            touch(l, *lst[i][3]); touch(l, *lst[j][3])
            check("inversion", abs(dr[i, j]), 2 * (bd[i] + bd[j]), f"limb {l}: pairs {lst[i][3]} / {lst[j][3]} swap order: reference scores {sr_[i]:.6f} / {sr_[j]:.6f}, engine {se_[i]:.6f} / {se_[j]:.6f}")
    # keep / drop of identical subset rows
    poff = 3 * (max_peaks + 1)

    def row_keys(rows, keys):
        d = {}
        for row in rows:
            ks = []
            for p in range(num_parts):
                o = int(row[p])
                if o:
                    ks.append((p, keys[p][(o - p * poff - 2) // 3 - 1]))
            d[frozenset(ks)] = row
        return d

    rr_, re2 = row_keys(tr[4], kr), row_keys(te[4], ke)
    keep_allow = 2 * (e_map + bound_max)
    for ks, row in rr_.items():
        other = re2.get(ks)
        if other is not None and row[num_parts + 2] != other[num_parts + 2]:
            cnt = row[num_parts]
            margin = abs(row[num_parts + 1] / cnt - thr["min_subset_score"]) if cnt >= thr["min_subset_cnt"] and other[num_parts] >= thr["min_subset_cnt"] else np.inf
            check("keep", margin, keep_allow, f"person with parts {sorted(p for p, _ in ks)}: score/count {row[num_parts + 1] / cnt:.6f} vs {other[num_parts + 1] / other[num_parts]:.6f}")

    # ---- per-joint attribution (reporting) ---------------------------------------------------------------------------------------
    attr = dict(nms=0, limb=0, propagated=0)
    def peak_key(side, part, x, y):
        pk, idx = (pk_r, idx_r) if side == "ref" else (pk_e, idx_e)
        xn, yn = x * net_w / disp_w, y * net_h / disp_h
        best, bk = None, None
        for k, i in idx[part].items():
            dd = abs(float(pk[part, i, 0]) - xn) + abs(float(pk[part, i, 1]) - yn)
            if best is None or dd < best:
                best, bk = dd, k
        return bk

    for ent in structural:
        side, part = ent[0], ent[2]
        ks = [peak_key("engine" if side != "ref" else "ref", part, ent[3], ent[4])]
        if side == "both":
            ks.append(peak_key("ref", part, ent[6], ent[7]))
        if any(k is not None and k in flipped[part] for k in ks):
            attr["nms"] += 1          # the joint sits on a maximum only one side has
        elif any(k is not None and any((part, k) in s for s in limb_roots.values()) for k in ks):
            attr["limb"] += 1         # its peak takes part in a flipped PAF test / an order inversion of one of its limbs
        else:
            attr["propagated"] += 1   # downstream: a greedy pick or a person row re-routed by flips elsewhere
    # ---- counterfactual replay: the per-joint proof (tests/_replay.py) --------------------------------------------------------------
    import _replay
    rep = _replay.replay(model, res_r, res_e, kr, ke, mr, pk_r, pk_e, tr, te, max_peaks, net_w, net_h, disp_w, disp_h, thr, e_heat, e_paf, pmax,
                         tol_px / max(disp_w / net_w, disp_h / net_h))
    out.update(rep)
    flips = []
    for part, xe, ye, xr, yr, _dc in out_of_tol:
        k_e, k_r = peak_key("engine", part, xe, ye), peak_key("ref", part, xr, yr)
        flips.append(bool(k_e is not None and k_r is not None and k_e != k_r and (k_e in flipped[part] or k_r in flipped[part])))
    attr["nms"] += sum(flips)
    out["out_of_tol_is_flip"] = flips
    n_struct = len(structural) + sum(flips)
    n_roots = sum(roots.values())
    # every structural joint is explained iff (a) every root flip found above is a near-tie, and (b) the reference side replayed with the
    # engine's outcome forced at every decision that differs for the same inputs — each of them a checked near-tie — gives the ENGINE's
    # people exactly (no inference about what is "downstream": it is computed)
    ok = not unexplained and (n_struct == 0 or n_roots > 0) and rep["replay_identical"] and rep["replay_unexplained"] == 0
    out.update(root_flips=roots, root_flips_total=n_roots, unexplained=len(unexplained), unexplained_detail=[f"{k}: margin {m:.3e} >= allowance {a:.3e}: {w}" for k, m, a, w in unexplained[:8]],
               worst_margin_over_allowance=worst, joints_structural=n_struct, structural_explained=n_struct if ok else 0, attribution=attr,
               score_bound_max=bound_max)
    return out


def merge(reports):
    tot = dict(frames=len(reports))
    for k in ("joints_structural", "structural_explained", "unexplained", "root_flips_total"):
        tot[k] = int(sum(r[k] for r in reports))
    tot["root_flips"] = {k: int(sum(r["root_flips"][k] for r in reports)) for k in (reports[0]["root_flips"] if reports else {})}
    tot["attribution"] = {k: int(sum(r["attribution"][k] for r in reports)) for k in (reports[0]["attribution"] if reports else {})}
    for k in ("e_map", "e_pos_net_px", "worst_margin_over_allowance"):
        tot[k] = float(max([r[k] for r in reports], default=0.0))
    tot["unexplained_detail"] = [d for r in reports for d in r["unexplained_detail"]][:8]
    tot["replay_identical"] = bool(all(r["replay_identical"] for r in reports))
    tot["replay_unexplained"] = int(sum(r["replay_unexplained"] for r in reports))
    tot["replay_unexplained_detail"] = [d for r in reports for d in r["replay_unexplained_detail"]][:8]
    tot["replay_forced"] = {k: int(sum(r["replay_forced"][k] for r in reports)) for k in (reports[0]["replay_forced"] if reports else {})}
    tot["replay_worst_margin_over_allowance"] = float(max([r["replay_worst_margin_over_allowance"] for r in reports], default=0.0))
    return tot
