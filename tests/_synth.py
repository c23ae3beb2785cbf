"""Synthetic inputs for parity tests (no images, weights or videos ship with the reference).

* random frames / random low-res maps (seeded)
* analytic low-res heat maps with P planted stick figures: part maps are Gaussians at the joints,
  PAF channels hold the unit limb vector inside a band around each limb.
"""
import numpy as np

# canonical skeletons in a unit box (x right, y down), index = part id
COCO_POSE = {
    0: (0.50, 0.08), 1: (0.50, 0.22), 2: (0.36, 0.22), 3: (0.30, 0.40), 4: (0.27, 0.56),
    5: (0.64, 0.22), 6: (0.70, 0.40), 7: (0.73, 0.56), 8: (0.41, 0.55), 9: (0.40, 0.75),
    10: (0.39, 0.95), 11: (0.59, 0.55), 12: (0.60, 0.75), 13: (0.61, 0.95), 14: (0.46, 0.05),
    15: (0.54, 0.05), 16: (0.41, 0.09), 17: (0.59, 0.09),
}
MPI_POSE = {
    0: (0.50, 0.06), 1: (0.50, 0.20), 2: (0.36, 0.22), 3: (0.30, 0.40), 4: (0.27, 0.56),
    5: (0.64, 0.22), 6: (0.70, 0.40), 7: (0.73, 0.56), 8: (0.41, 0.58), 9: (0.40, 0.77),
    10: (0.39, 0.95), 11: (0.59, 0.58), 12: (0.60, 0.77), 13: (0.61, 0.95), 14: (0.50, 0.40),
}


def random_frame(N, H, W, seed):
    """What process_and_pad_image(normalize=1) produces: u8/256 - 0.5 (rtpose.cpp:259)."""
    rs = np.random.RandomState(seed)
    u8 = rs.randint(0, 256, size=(N, 3, H, W)).astype(np.float32)
    return (u8 / 256.0 - 0.5).astype(np.float32)


def smooth_field(C, h, w, seed, scale=1.0):
    """Band-limited random field in roughly [-scale, scale]: many local maxima, few exact ties."""
    rs = np.random.RandomState(seed)
    f = rs.randn(C, h, w).astype(np.float32)
    k = np.array([1, 4, 6, 4, 1], np.float32) / 16.0
    for ax in (1, 2):
        f = sum(np.roll(f, s - 2, axis=ax) * k[s] for s in range(5))
    f = f / (np.abs(f).max() + 1e-9)
    return (f * scale).astype(np.float32)


def people_lowres(model, tables, P, h, w, seed, N=1, amp=0.9, sigma=0.9, band=0.8, jitter=True):
    """Low-res maps [N][C][h][w] for P planted people.  Channel layout = concat_stage7:
    heat maps (num_parts + background) first, PAFs second (coco prototxt:2966-2975)."""
    num_parts, num_limbs, limb_seq, map_idx = tables
    pose = COCO_POSE if model == 0 else MPI_POSE
    C = (num_parts + 1) + 2 * num_limbs
    rs = np.random.RandomState(seed)
    m = np.zeros((C, h, w), np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    people = []
    for p in range(P):
        size = rs.uniform(0.45, 0.8) * h
        cx = rs.uniform(0.15, 0.85) * w
        cy = rs.uniform(0.05, 0.25) * h
        pts = {}
        for k, (ux, uy) in pose.items():
            jx = rs.uniform(-0.3, 0.3) if jitter else 0.0
            jy = rs.uniform(-0.3, 0.3) if jitter else 0.0
            pts[k] = (cx + (ux - 0.5) * size * 0.6 + jx, cy + uy * size + jy)
        people.append(pts)
        for k, (px, py) in pts.items():
            if not (1.5 < px < w - 2.5 and 1.5 < py < h - 2.5):
                continue
            g = amp * np.exp(-((xx - px) ** 2 + (yy - py) ** 2) / (2 * sigma * sigma))
            m[k] = np.maximum(m[k], g.astype(np.float32))
        for l in range(num_limbs):
            a, b = limb_seq[2 * l], limb_seq[2 * l + 1]
            (ax, ay), (bx, by) = pts[a], pts[b]
            d = np.array([bx - ax, by - ay], np.float32)
            n = float(np.hypot(*d))
            if n < 1e-3:
                continue
            u = d / n
            t = (xx - ax) * u[0] + (yy - ay) * u[1]
            perp = np.abs((xx - ax) * u[1] - (yy - ay) * u[0])
            mask = (t >= -0.5) & (t <= n + 0.5) & (perp <= band)
            cx_, cy_ = map_idx[2 * l], map_idx[2 * l + 1]
            m[cx_][mask] = u[0]
            m[cy_][mask] = u[1]
    m[num_parts] = 1.0 - m[:num_parts].max(axis=0)
    if jitter:  # break exact ties deterministically
        m += (rs.rand(C, h, w).astype(np.float32) - 0.5) * 1e-3
    out = np.repeat(m[None], N, axis=0).astype(np.float32)
    return np.ascontiguousarray(out), people
