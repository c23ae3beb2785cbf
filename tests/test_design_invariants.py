"""Two pieces of index / bound arithmetic that kernels rely on, checked on the CPU against direct computation.

1. The pooled tile walk of conv_ring.hip (POOL): tiles of 2 image rows x 64 pixels walked with an even pitch, per-lane DMA source offsets,
   wrap into the next row pair, zero columns served by the gap pixel.  The kernel's address arithmetic is restated here 1:1 and run against a
   direct 3x3 convolution + 2x2 max pooling on a halo'd flat tensor (one channel is enough: the arithmetic does not depend on channels).
2. The bound behind the exact skip of the ImResize+Nms strip kernel (postproc.hip NMS_BOUND = 1.95): no resized value exceeds
   1.95 x max |4x4 low-res neighbourhood|, borders (negative fraction, imresize_layer.cu:123-128) and several scales included — checked on the
   oracle's ImResize (the reference's kernel semantics, pinned on oracle/_ref)."""
import numpy as np
import pytest

import _oracle as orc


def _pooled_walk(H, W, halo, seed=0):
    Wp, Hp = W + halo, H + 2 * halo
    BM, HALF, KS, PAD = 128, 64, 3, 1
    PHALF = HALF + KS - 1
    Wq = (W + PAD + 1) & ~1                              # engine.cpp: pool_wq
    rs = np.random.RandomState(seed)
    flat = np.zeros(Hp * Wp + 8 * Wp)                    # halo'd tensor, flat (kernels.h), + slack the last tile reads
    img = rs.rand(H, W)
    for y in range(H):
        flat[(y + halo) * Wp + halo:(y + halo) * Wp + halo + W] = img[y]
    wts = rs.rand(3, 3)
    pad = np.zeros((H + 2, W + 2)); pad[1:-1, 1:-1] = img
    conv = sum(wts[r, s] * pad[r:r + H, s:s + W] for r in range(3) for s in range(3))
    ref = conv.reshape(H // 2, 2, W // 2, 2).max(axis=(1, 3))
    out = np.full((H // 2, W // 2), np.nan)
    ntiles = ((H // 2) * Wq + HALF - 1) // HALF          # engine.cpp: tiles_per_img
    for t in range(ntiles):
        q0 = t * HALF
        pair, x0 = divmod(q0, Wq)
        base = halo * Wp + 2 * pair * Wp - PAD * Wp - PAD  # a_ptr (pixels), filter row 0
        acc = np.zeros(BM)
        for r in range(3):
            lds = np.empty(2 * PHALF)
            for row in range(2 * PHALF):                 # conv_ring.hip: a_voff of LDS strip row `row`
                sel = 1 if row >= PHALF else 0
                a = x0 - PAD + (row - sel * PHALF)
                wrap = a >= Wq
                ar = a - Wq if wrap else a
                src = (2 * Wp if wrap else 0) + (ar if ar < W else W) + halo + PAD + sel * Wp
                lds[row] = flat[base + r * Wp + src]
            for s in range(3):                           # consumer: wave row wm multiplies LDS rows wm*PHALF + lrow + s
                for lr in range(BM):
                    wm, rr = divmod(lr, HALF)
                    acc[lr] += wts[r, s] * lds[wm * PHALF + rr + s]
        for k in range(HALF // 2):                       # conv_epilogue_pool
            x, pr = x0 + 2 * k, pair
            if x >= Wq:
                x -= Wq; pr += 1
            if pr >= H // 2 or x >= W:
                continue
            assert np.isnan(out[pr, x // 2])             # every pooled pixel is produced exactly once
            out[pr, x // 2] = max(acc[2 * k], acc[2 * k + 1], acc[HALF + 2 * k], acc[HALF + 2 * k + 1])
    return out, ref


@pytest.mark.parametrize("H,W", [(4, 128), (6, 130), (10, 164), (8, 248), (12, 328), (6, 656), (4, 720), (2, 1312)])
@pytest.mark.parametrize("halo", [1, 3])
def test_pooled_tile_walk_covers_every_pooled_pixel_with_the_right_inputs(H, W, halo):
    out, ref = _pooled_walk(H, W, halo)                  # (the engine fuses only where W >= 128: one wrap per tile at most)
    assert not np.isnan(out).any()
    assert np.abs(out - ref).max() < 1e-12


def _window_bound(low, tw, th, start, gap):
    """max |4x4 low-res neighbourhood| per output pixel, summed over the scales / num (the average of the per-scale bounds)."""
    num, C, h, w = low.shape
    tot = np.zeros((C, th, tw))
    for n in range(num):
        padw = int(np.floor(np.float32(w // 2) * np.float32(1 - start + n * gap)))
        padh = int(np.floor(np.float32(h // 2) * np.float32(1 - start + n * gap)))
        ow, oh = w - 2 * padw, h - 2 * padh

        def nb(size_t, osize):
            off = np.float32(np.float64(np.float32(size_t) / np.float32(osize) / 2) - 0.5)
            on = (np.arange(size_t, dtype=np.float32) - off) * (np.float32(osize) / np.float32(size_t))
            n1 = np.maximum((on.astype(np.float64) + 1e-5).astype(np.int64), 0)
            n0 = np.maximum(n1 - 1, 0)
            n2 = np.minimum(n1 + 1, osize - 1)
            n3 = np.minimum(n2 + 1, osize - 1)
            return n0, n3
        x0, x3 = nb(tw, ow)
        y0, y3 = nb(th, oh)
        a = np.abs(low[n][:, padh:padh + oh, padw:padw + ow])
        for y in range(th):
            rows = a[:, y0[y]:y3[y] + 1, :].max(axis=1)               # [C][ow]
            # running max over the column window [x0, x3] (width <= 4)
            m = np.stack([rows[:, np.minimum(x0 + d, x3)] for d in range(4)], 0).max(axis=0)
            tot[:, y, :] += m
    return tot / num


@pytest.mark.parametrize("num,gap", [(1, 0.3), (3, 0.15), (2, 0.25)])
def test_no_resized_value_exceeds_the_nms_skip_bound(num, gap):
    h, w, tw, th = 46, 82, 656, 368
    rs = np.random.RandomState(5 + num)
    worst = 0.0
    for trial in range(5):
        low = rs.randn(num, 2, h, w).astype(np.float32)
        if trial >= 3:   # adversarial: the sign pattern (-, +, +, -) of the cubic's weights along both axes, at two phases
            px = np.array([-1, 1, 1, -1], np.float32)[(np.arange(w) + trial) % 4]
            py = np.array([-1, 1, 1, -1], np.float32)[(np.arange(h) + trial) % 4]
            low = np.broadcast_to(py[:, None] * px[None, :], low.shape).astype(np.float32).copy()
        if trial == 1:   # isolated spikes: the worst case for the cubic's overshoot
            low = np.zeros_like(low)
            idx = rs.randint(0, low.size, 400)
            low.reshape(-1)[idx] = rs.choice([-1.0, 1.0], 400).astype(np.float32)
        if trial == 2:   # border rows / columns only
            m = np.zeros_like(low); m[:, :, :2, :] = low[:, :, :2, :]; m[:, :, :, :2] = low[:, :, :, :2]; m[:, :, -2:, :] = low[:, :, -2:, :]; m[:, :, :, -2:] = low[:, :, :, -2:]
            low = m
        res = orc.imresize(low, tw, th, 1.0, gap)[0]
        bound = _window_bound(low, tw, th, 1.0, gap)
        ratio = np.abs(res) / np.maximum(bound, 1e-30)
        ratio[bound == 0] = np.where(np.abs(res[bound == 0]) == 0, 0.0, np.inf)
        worst = max(worst, float(ratio.max()))
    print(f"max |resized| / max |neighbourhood| = {worst:.4f} (kernel bound 1.95, analytic 1.375^2 = 1.8906)")
    assert worst <= 1.8906 * 1.001 < 1.95
