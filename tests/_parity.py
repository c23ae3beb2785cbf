"""Set-based people / joint parity report (SURVEY.md section 7 "hard parts": NMS' strict compares and
connect's greedy picks turn 1-ulp map differences into different peak sets, so engine and reference outputs are compared as SETS
of people, not byte for byte).

A "person" is one row of the joints array the reference emits (rtpose.cpp:1051-1073): num_parts x (x, y, score) in display
coordinates, a part the person does not have is (0, 0, 0).  Test infrastructure + bench.py's `parity` dict; host numpy only.

Round 4: joints are paired as "the same peak" within a radius WIDER than the tolerance (pair_px, 3 display pixels), so that a joint
which moved by more than tol_px is reported as `numeric_out_of_tol` — a FAIL — instead of disappearing into the structural count, and
max_dx_px / max_dy_px / max_dc are taken over EVERY paired joint, inside the tolerance or not.

One thing joints alone cannot tell: whether two joints 1-2 display pixels apart are the same maximum whose centroid moved (numeric) or
two ADJACENT net pixels that swapped the role of the maximum (a flipped NMS compare between neighbours; the 7x7 centroid follows the
integer maximum, so the joint jumps by about one net pixel = 1.95 display pixels).  With a smooth peak (sigma ~7 net pixels) the two
pixels next to its centre differ by less than 1e-3 of the peak whenever the centre lies within ~0.1 pixel of the midpoint between
them: the reference's own output is unstable at the one-net-pixel level under a sub-tolerance deviation.  tests/_explain.py has the maps
and decides by the integer maximum: `reclassify()` moves the out-of-tolerance pairs that are two different maxima (whose NMS margins
_explain asserts to be sub-tolerance) from `numeric_out_of_tol` to `joints_structural`.  Without _explain they stay numeric: a FAIL."""
import numpy as np


def people_parity(je, jr, tol_px=1.0, tol_c=1e-3, c_norm=1.0, pair_px=3.0):
    """je / jr: joints [n][P][3] of the engine / the reference for ONE frame.  Returns a dict:

    people_engine / people_ref      rows on either side
    people_matched                  one-to-one pairs in which EVERY part agrees (both absent, or both present within tol_px in x
                                    and y and within tol_c in score / c_norm)
    joints_ref / joints_matched     present parts of the reference people / how many of them the paired engine person has within tolerance
    numeric_out_of_tol              joints of paired people that ARE the same peak (within pair_px) but deviate by more than
                                    tol_px or tol_c: the conv stack's deviation arriving outside the tolerance — any is a FAIL
    joints_structural               parts of paired people that are a different peak (farther than pair_px) or present on one side
                                    only (a flipped compare upstream, not a numeric deviation); parts of unpaired people count here too
    structural                      the list behind joints_structural, one entry per counted joint: (side, person index, part, x, y,
                                    score[, x_ref, y_ref]) with side "engine" / "ref" (present on that side only) or "both" (two
                                    different peaks; engine coordinates first) — tests/_explain.py traces each to the decision that flipped
    out_of_tol                      the list behind numeric_out_of_tol: (part, x_engine, y_engine, x_ref, y_ref, |d score| / c_norm)
    max_dx_px / max_dy_px / max_dc  over ALL same-peak joints of paired people (can exceed the tolerance; bounded only by pair_px)
    """
    je = np.asarray(je, np.float64).reshape(-1, je.shape[-2], 3) if len(je) else np.zeros((0, jr.shape[-2] if len(jr) else 1, 3))
    jr = np.asarray(jr, np.float64).reshape(-1, jr.shape[-2], 3) if len(jr) else np.zeros((0, je.shape[-2], 3))
    ne, nr = len(je), len(jr)
    out = dict(people_engine=ne, people_ref=nr, people_matched=0, joints_ref=0, joints_matched=0, numeric_out_of_tol=0, joints_structural=0,
               max_dx_px=0.0, max_dy_px=0.0, max_dc=0.0, tol_px=tol_px, tol_c=tol_c, pair_px=pair_px, structural=[], out_of_tol=[])
    pe = (je != 0).any(-1) if ne else np.zeros((0, 1), bool)
    pr = (jr != 0).any(-1) if nr else np.zeros((0, 1), bool)
    out["joints_ref"] = int(pr.sum())

    def lone(side, j, present, idx):
        for p in np.nonzero(present[idx])[0]:
            out["structural"].append((side, int(idx), int(p), float(j[idx, p, 0]), float(j[idx, p, 1]), float(j[idx, p, 2])))

    if ne == 0 or nr == 0:
        for i in range(ne):
            lone("engine", je, pe, i)
        for j in range(nr):
            lone("ref", jr, pr, j)
        out["joints_structural"] = len(out["structural"])
        return out
    P = je.shape[1]
    d = np.abs(je[:, None] - jr[None])                       # [ne][nr][P][3]
    both = pe[:, None] & pr[None]
    neither = ~pe[:, None] & ~pr[None]
    same_peak = both & (d[..., 0] <= pair_px) & (d[..., 1] <= pair_px)
    in_tol = same_peak & (d[..., 0] <= tol_px) & (d[..., 1] <= tol_px) & (d[..., 2] / c_norm <= tol_c)
    agree = neither | in_tol
    score = same_peak.sum(-1) * (P + 1) + agree.sum(-1)      # pair by shared peaks first, full agreement second
    order = np.dstack(np.unravel_index(np.argsort(-score, axis=None, kind="stable"), score.shape))[0]
    used_e, used_r, pairs = set(), set(), []
    for i, j in order:
        if same_peak[i, j].sum() == 0:
            break
        if i in used_e or j in used_r:
            continue
        used_e.add(int(i)); used_r.add(int(j)); pairs.append((int(i), int(j)))
    for i, j in pairs:
        sp = same_peak[i, j]
        if agree[i, j].all():
            out["people_matched"] += 1
        out["joints_matched"] += int(in_tol[i, j].sum())
        out["numeric_out_of_tol"] += int((sp & ~in_tol[i, j]).sum())
        for p in np.nonzero(sp & ~in_tol[i, j])[0]:
            out["out_of_tol"].append((int(p), float(je[i, p, 0]), float(je[i, p, 1]), float(jr[j, p, 0]), float(jr[j, p, 1]), float(d[i, j, p, 2] / c_norm)))
        for p in np.nonzero((pe[i] | pr[j]) & ~sp)[0]:
            if pe[i, p] and pr[j, p]:
                out["structural"].append(("both", i, int(p), float(je[i, p, 0]), float(je[i, p, 1]), float(je[i, p, 2]), float(jr[j, p, 0]), float(jr[j, p, 1])))
            elif pe[i, p]:
                out["structural"].append(("engine", i, int(p), float(je[i, p, 0]), float(je[i, p, 1]), float(je[i, p, 2])))
            else:
                out["structural"].append(("ref", j, int(p), float(jr[j, p, 0]), float(jr[j, p, 1]), float(jr[j, p, 2])))
        out["joints_structural"] += int(((pe[i] | pr[j]) & ~sp).sum())
        it = in_tol[i, j]
        if it.any():   # the same maxima restricted to the joints inside the tolerance (what is left after reclassify() removed flips)
            out["_in_tol_max"] = [max(out.get("_in_tol_max", [0.0, 0.0])[0], float(d[i, j, it, 0].max())), max(out.get("_in_tol_max", [0.0, 0.0])[1], float(d[i, j, it, 1].max()))]
        if sp.any():
            out["max_dx_px"] = max(out["max_dx_px"], float(d[i, j, sp, 0].max()))
            out["max_dy_px"] = max(out["max_dy_px"], float(d[i, j, sp, 1].max()))
            out["max_dc"] = max(out["max_dc"], float(d[i, j, sp, 2].max() / c_norm))
    for i in range(ne):
        if i not in used_e:
            out["joints_structural"] += int(pe[i].sum())
            lone("engine", je, pe, i)
    for j in range(nr):
        if j not in used_r:
            out["joints_structural"] += int(pr[j].sum())
            lone("ref", jr, pr, j)
    return out


def reclassify(rep, flags):
    """flags[i] (from _explain.explain(..., out_of_tol=rep['out_of_tol'])['out_of_tol_is_flip']): pair i of rep['out_of_tol'] is two
    DIFFERENT integer maxima (one of them exists on one side only: an NMS flip between neighbouring pixels), not one maximum that
    moved.  Those pairs become structural differences; max_dx_px / max_dy_px / max_dc are re-taken over what stays numeric."""
    n = int(sum(bool(f) for f in flags))
    rep["numeric_out_of_tol"] -= n
    rep["joints_structural"] += n
    rep["adjacent_pixel_flips"] = rep.get("adjacent_pixel_flips", 0) + n
    for f, (p, xe, ye, xr, yr, _dc) in zip(flags, rep["out_of_tol"]):
        if f:
            rep["structural"].append(("both", -1, p, xe, ye, 0.0, xr, yr))
    rep["out_of_tol"] = [o for f, o in zip(flags, rep["out_of_tol"]) if not f]
    if n:   # the maxima were taken over pairs that turned out to be different peaks: bound them by what is left
        keep = rep["out_of_tol"]
        base = rep.get("_in_tol_max", [0.0, 0.0])
        rep["max_dx_px"] = max([abs(o[1] - o[3]) for o in keep] + [base[0]])
        rep["max_dy_px"] = max([abs(o[2] - o[4]) for o in keep] + [base[1]])
    return rep


def merge(reports):
    """Totals over several frames (maxima of the maxima).  The per-frame `structural` lists stay with their frames."""
    tot = dict(frames=len(reports))
    for k in ("people_engine", "people_ref", "people_matched", "joints_ref", "joints_matched", "numeric_out_of_tol", "joints_structural"):
        tot[k] = int(sum(r[k] for r in reports))
    tot["adjacent_pixel_flips"] = int(sum(r.get("adjacent_pixel_flips", 0) for r in reports))
    for k in ("max_dx_px", "max_dy_px", "max_dc"):
        tot[k] = float(max([r[k] for r in reports], default=0.0))
    if reports:
        tot["tol_px"], tot["tol_c"], tot["pair_px"] = reports[0]["tol_px"], reports[0]["tol_c"], reports[0]["pair_px"]
    return tot


def verdict(tot, map_err=None, post_exact=True, explained=None):
    """The one-word verdict of a merged report.
      FAIL ...        a paired joint outside +-tol_px / +-tol_c, the maps outside tol_c, the post-processing not bit-exact on the
                      engine's own maps, or a structural difference that no sub-tolerance near-tie explains (explained = the
                      number of structural joints tests/_explain.py traced to such a flip; None = not traced -> cannot pass
                      unless there is nothing to explain)
      pass            identical people sets, every joint inside the tolerance
      numeric pass... every paired joint inside the tolerance, every structural difference traced to a near-tie decision"""
    if tot["numeric_out_of_tol"] > 0 or tot["max_dc"] > tot["tol_c"] or tot["max_dx_px"] > tot["tol_px"] or tot["max_dy_px"] > tot["tol_px"]:
        return f"FAIL: {tot['numeric_out_of_tol']} paired joint(s) outside +-{tot['tol_px']} px / +-{tot['tol_c']} (max dx {tot['max_dx_px']:.3f} dy {tot['max_dy_px']:.3f} dc {tot['max_dc']:.2e})"
    if map_err is not None and map_err > tot["tol_c"]:
        return f"FAIL: final maps deviate by {map_err:.2e} of the map maximum (> {tot['tol_c']})"
    if not post_exact:
        return "FAIL: the reference's post-processing applied to the engine's maps does not reproduce the engine's joints"
    if tot["joints_structural"] == 0 and tot["people_matched"] == tot["people_ref"] == tot["people_engine"]:
        return "pass"
    if explained is None or explained != tot["joints_structural"]:
        return f"FAIL: {tot['joints_structural'] - (explained or 0)} of {tot['joints_structural']} structural joint differences are not traced to a sub-tolerance near-tie"
    return "numeric pass; every structural difference traced to a near-tie decision (NMS '>' / PAF threshold / greedy order) whose reference-side margin is below 2x the measured map deviation"
