"""Set-based people / joint parity report (SURVEY.md section 7 "hard parts": NMS' strict compares and
connect's greedy picks turn 1-ulp map differences into different peak sets, so engine and reference outputs are compared as SETS
of people, not byte for byte).

A "person" is one row of the joints array the reference emits (rtpose.cpp:1051-1073): num_parts x (x, y, score) in display
coordinates, a part the person does not have is (0, 0, 0).  Test infrastructure + bench.py's `parity` dict; host numpy only."""
import numpy as np


def people_parity(je, jr, tol_px=1.0, tol_c=1e-3, c_norm=1.0):
    """je / jr: joints [n][P][3] of the engine / the reference for ONE frame.  Returns a dict:

    people_engine / people_ref      rows on either side
    people_matched                  one-to-one pairs in which EVERY part agrees (both absent, or both present within tol_px in x
                                    and y and within tol_c in score / c_norm)
    joints_ref / joints_matched     present parts of the reference people / how many of them the paired engine person has within tolerance
    joints_structural               parts of paired people that are a different peak or present on one side only (a flipped
                                    compare upstream, not a numeric deviation); parts of unpaired people count here too
    max_dx_px / max_dy_px / max_dc  over the corresponding joints of paired people (same peak: position within tol_px): the
                                    numeric deviation of the conv stack as it arrives in the output
    """
    je = np.asarray(je, np.float64).reshape(-1, je.shape[-2], 3) if len(je) else np.zeros((0, jr.shape[-2] if len(jr) else 1, 3))
    jr = np.asarray(jr, np.float64).reshape(-1, jr.shape[-2], 3) if len(jr) else np.zeros((0, je.shape[-2], 3))
    ne, nr = len(je), len(jr)
    out = dict(people_engine=ne, people_ref=nr, people_matched=0, joints_ref=0, joints_matched=0, joints_structural=0,
               max_dx_px=0.0, max_dy_px=0.0, max_dc=0.0, tol_px=tol_px, tol_c=tol_c)
    pe = (je != 0).any(-1) if ne else np.zeros((0, 1), bool)
    pr = (jr != 0).any(-1) if nr else np.zeros((0, 1), bool)
    out["joints_ref"] = int(pr.sum())
    if ne == 0 or nr == 0:
        out["joints_structural"] = int(pe.sum() + pr.sum())
        return out
    P = je.shape[1]
    d = np.abs(je[:, None] - jr[None])                       # [ne][nr][P][3]
    both = pe[:, None] & pr[None]
    neither = ~pe[:, None] & ~pr[None]
    same_peak = both & (d[..., 0] <= tol_px) & (d[..., 1] <= tol_px)
    agree = neither | (same_peak & (d[..., 2] / c_norm <= tol_c))
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
        out["joints_matched"] += int((sp & agree[i, j]).sum())
        out["joints_structural"] += int(((pe[i] | pr[j]) & ~sp).sum())
        if sp.any():
            out["max_dx_px"] = max(out["max_dx_px"], float(d[i, j, sp, 0].max()))
            out["max_dy_px"] = max(out["max_dy_px"], float(d[i, j, sp, 1].max()))
            out["max_dc"] = max(out["max_dc"], float(d[i, j, sp, 2].max() / c_norm))
    out["joints_structural"] += int(sum(pe[i].sum() for i in range(ne) if i not in used_e) + sum(pr[j].sum() for j in range(nr) if j not in used_r))
    return out


def merge(reports):
    """Totals over several frames (maxima of the maxima)."""
    tot = dict(frames=len(reports))
    for k in ("people_engine", "people_ref", "people_matched", "joints_ref", "joints_matched", "joints_structural"):
        tot[k] = int(sum(r[k] for r in reports))
    for k in ("max_dx_px", "max_dy_px", "max_dc"):
        tot[k] = float(max([r[k] for r in reports], default=0.0))
    if reports:
        tot["tol_px"], tot["tol_c"] = reports[0]["tol_px"], reports[0]["tol_c"]
    return tot
