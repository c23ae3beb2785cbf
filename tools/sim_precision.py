#!/usr/bin/env python
"""CPU simulation of storage-precision modes of the linevec conv stack (test tooling, torch CPU).

Answers "which roundings cost how much of the final-map error": every layer is an fp32 conv2d
(what fp16 x fp16 products with fp32 accumulation give, up to 1e-7) whose weights and/or input
activations are first rounded the way a storage mode rounds them:
  h  = one fp16 number                       (11 significant bits)
  hh = hi + lo pair of fp16 numbers          (~22 bits: the split mode)
  f  = fp32
Usage: sim_precision.py [--w 160 --h 96] MODE...   with MODE = <weights>:<activations>[:<section-overrides>]
e.g.   h:h   f:h   h:f   hh:h   h:h:s6=hh:hh  h:h:cat=hh
"""
import argparse
import re
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C  # noqa: E402

from caffe_rtpose_amd._lib import lib  # noqa: E402


WSEED = 1


def synth(name, cout, cin, k, seed=None):
    seed = WSEED if seed is None else seed
    w = np.empty((cout, cin, k, k), np.float32)
    b = np.empty((cout,), np.float32)
    fp = C.POINTER(C.c_float)
    lib.rtp_synth_weights(C.c_uint64(seed), name.encode(), cout, cin, k, w.ctypes.data_as(fp), b.ctypes.data_as(fp))
    return torch.from_numpy(w), torch.from_numpy(b)


def rnd(t, mode):
    if mode == "f":
        return t
    hi = t.half().float()
    if mode == "h":
        return hi
    lo = (t - hi).half().float()
    return hi + lo


def f8(t, scale_exp):
    """e4m3fn rounding of t * 2^scale_exp (clamped to +-448 first), returned in t's units."""
    sc = 2.0 ** scale_exp
    return (t * sc).clamp(-448, 448).to(torch.float8_e4m3fn).float() / sc


def conv_h8(x, w, b, pad, comp_a=True, comp_w=True):
    """hi*hi in fp16 + fp8 error compensation: a_lo8 * W8 + a8 * W_lo8 (what the MX-scaled fp8 MFMA passes compute)."""
    xh = x.half().float(); wh = w.half().float()
    y = F.conv2d(xh, wh, b, padding=pad)
    wmax = float(w.abs().max())
    tw = int(np.floor(np.log2(448.0 / wmax)))
    if comp_a:
        xlo = x - xh
        y = y + F.conv2d(f8(xlo, 12), f8(wh, tw), None, padding=pad)
    if comp_w:
        wlo = w - wh
        y = y + F.conv2d(f8(xh, 2), f8(wlo, tw + 11), None, padding=pad)
    return y


MINIFLOAT = {"e2m1": (1, 0, 2, 6.0), "e2m3": (3, 0, 2, 7.5), "e3m2": (2, -2, 4, 28.0)}  # mantissa bits, min normal exponent, max exponent, max value
H4_FMT = "e2m1"


H4_FMT_W = None   # weights' format when it differs from the activations'


def fp4_round(t, fmt=None):
    """nearest value of an MX minifloat format (round half to even), |t| clamped to the format maximum; subnormals kept"""
    m, emin, emax, vmax = MINIFLOAT[fmt or H4_FMT]
    a = t.abs().clamp(max=vmax).double()
    e = torch.floor(torch.log2(a.clamp(min=2.0 ** -60))).clamp(min=emin, max=emax)
    q = 2.0 ** (e - m)
    return (torch.round(a / q) * q).clamp(max=vmax).float() * t.sign()


def blk_exp(x, block):
    """per (pixel, `block` channels) exponent s with max|x| <= 6 * 2^s (smallest such), as a tensor broadcastable to x (channels padded)"""
    n, c, h, w = x.shape
    cp = (c + block - 1) // block * block
    xp = F.pad(x, (0, 0, 0, 0, 0, cp - c))
    m = xp.abs().view(n, cp // block, block, h, w).amax(2, keepdim=True)
    vmax = MINIFLOAT[H4_FMT][3]
    s = torch.ceil(torch.log2((m / vmax).clamp(min=2.0 ** -60)))   # smallest s with max <= vmax * 2^s
    return s.expand(n, cp // block, block, h, w).reshape(n, cp, h, w)[:, :c]


H4_SKIP_FIRST = 0   # Mconv1 experiments: no correction for the first n input channels (the re-injected 57 head maps)


def conv_h4(x, w, b, pad, comp_a=True, comp_w=True, block=32, lo_shift=11):
    if H4_SKIP_FIRST and x.shape[1] > 128:
        n = x.shape[1] - 128
        y0 = F.conv2d(x[:, :n].half().float(), w[:, :n].half().float(), None, padding=pad)
        return y0 + conv_h4(x[:, n:], w[:, n:], b, pad, comp_a, comp_w, block, lo_shift)
    """hi*hi in fp16 + fp4 (e2m1, MX block scales) error compensation: a_lo4 * W4 + a4 * W_lo4.
    activations: one E8M0 scale per (pixel, 32 channels) from the block maximum; weights: one static scale per layer"""
    xh = x.half().float(); wh = w.half().float()
    y = F.conv2d(xh, wh, b, padding=pad)
    s = blk_exp(x, block)
    fw = H4_FMT_W or H4_FMT
    sw = float(np.ceil(np.log2(float(w.abs().max()) / MINIFLOAT[fw][3])))
    if comp_a:
        xlo = x - xh
        sl = blk_exp(xlo, block) if lo_shift < 0 else s - lo_shift   # lo_shift < 0: the lo block gets its own scale from max |lo|
        a4 = fp4_round(xlo / 2.0 ** sl) * 2.0 ** sl
        w4 = fp4_round(wh / 2.0 ** sw, fw) * 2.0 ** sw
        y = y + F.conv2d(a4, w4, None, padding=pad)
    if comp_w:
        wlo = w - wh
        a4 = fp4_round(xh / 2.0 ** s) * 2.0 ** s
        swl = float(np.ceil(np.log2(max(float(wlo.abs().max()), 1e-30) / MINIFLOAT[fw][3]))) if lo_shift < 0 else sw - lo_shift
        w4 = fp4_round(wlo / 2.0 ** swl, fw) * 2.0 ** swl
        y = y + F.conv2d(a4, w4, None, padding=pad)
    return y


def layers(model=0):
    """(name, cin, cout, k, relu, section) in execution order + topology handled in forward()."""
    nL1, nL2 = (38, 19) if model == 0 else (28, 16)
    trunk = [("conv1_1", 3, 64), ("conv1_2", 64, 64), "pool", ("conv2_1", 64, 128), ("conv2_2", 128, 128), "pool",
             ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv3_4", 256, 256), "pool",
             ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3_CPM", 512, 256), ("conv4_4_CPM", 256, 128)]
    return trunk, nL1, nL2


def forward(x, wm, am, over, model=0):
    trunk, nL1, nL2 = layers(model)

    def conv(name, t, cin, cout, k, relu, sec):
        w_mode, a_mode = over.get(sec, (wm, am))
        for pat, v in over.items():
            if pat.startswith("re/") and re.search(pat[3:], name):
                w_mode, a_mode = v
        w, b = synth(name, cout, cin, k)
        if w_mode == "h4" or a_mode == "h4":
            y = conv_h4(t, w, b, k // 2, comp_a=(a_mode == "h4"), comp_w=(w_mode == "h4"), block=H4_BLOCK, lo_shift=H4_LO)
        elif w_mode == "h8" or a_mode == "h8":
            y = conv_h8(t, w, b, k // 2, comp_a=(a_mode == "h8"), comp_w=(w_mode == "h8"))
        else:
            y = F.conv2d(rnd(t, a_mode), rnd(w, w_mode), b, padding=k // 2)
        return F.relu(y) if relu else y

    t = x
    for it in trunk:
        if it == "pool":
            t = F.max_pool2d(t, 2, 2)
        else:
            t = conv(it[0], t, it[1], it[2], 3, True, "trunk")
    feat = t
    outs = []
    for br, n in ((1, nL1), (2, nL2)):
        u = feat
        for i in (1, 2, 3):
            u = conv(f"conv5_{i}_CPM_L{br}", u, 128, 128, 3, True, "s1")
        u = conv(f"conv5_4_CPM_L{br}", u, 128, 512, 1, True, "s1")
        u = conv(f"conv5_5_CPM_L{br}", u, 512, n, 1, False, "s1")
        outs.append(u)
    for s in range(2, 7):
        cat = torch.cat([outs[0], outs[1], feat], 1)
        sec = f"s{s}"
        new = []
        for br, n in ((1, nL1), (2, nL2)):
            u = cat
            cin = cat.shape[1]
            for i in range(1, 6):
                # the concat re-injection can carry its own activation mode ("cat")
                s_eff = "cat" if (i == 1 and "cat" in over) else sec
                if s_eff == "cat" and False:
                    w_mode = over.get(sec, (wm, am))[0]
                    w, b = synth(f"Mconv{i}_stage{s}_L{br}", 128, cin, 7)
                    u = F.relu(F.conv2d(rnd(u, over["cat"][1]), rnd(w, w_mode), b, padding=3))
                else:
                    u = conv(f"Mconv{i}_stage{s}_L{br}", u, cin, 128, 7, True, sec)
                cin = 128
            u = conv(f"Mconv6_stage{s}_L{br}", u, 128, 128, 1, True, sec)
            u = conv(f"Mconv7_stage{s}_L{br}", u, 128, n, 1, False, sec)
            new.append(u)
        outs = new
    return torch.cat([outs[1], outs[0]], 1)  # concat_stage7: heat maps first


H4_BLOCK = 32
H4_LO = 11


def main():
    global H4_BLOCK, H4_LO, H4_FMT, H4_FMT_W, H4_SKIP_FIRST, WSEED
    ap = argparse.ArgumentParser()
    ap.add_argument("--h4_block", type=int, default=32)
    ap.add_argument("--h4_lo", type=int, default=11)
    ap.add_argument("--h4_fmt", default="e2m1", choices=sorted(MINIFLOAT))
    ap.add_argument("--h4_fmt_w", default=None, choices=sorted(MINIFLOAT))
    ap.add_argument("--h4_skip_heads", type=int, default=0)
    ap.add_argument("--wseed", type=int, default=1, help="seed of the synthetic weights (the engine's rtp_config.synthetic_seed)")
    ap.add_argument("--w", type=int, default=160)
    ap.add_argument("--h", type=int, default=96)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("modes", nargs="+")
    a = ap.parse_args()
    H4_BLOCK, H4_LO, H4_FMT, H4_FMT_W, H4_SKIP_FIRST = a.h4_block, a.h4_lo, a.h4_fmt, a.h4_fmt_w, a.h4_skip_heads
    WSEED = a.wseed
    torch.set_num_threads(os.cpu_count())
    rs = np.random.RandomState(a.seed)
    x = torch.from_numpy((rs.randint(0, 256, size=(1, 3, a.h, a.w)).astype(np.float32) / 256.0 - 0.5))
    with torch.no_grad():
        ref = forward(x.double().float(), "f", "f", {})
        mx = ref.abs().max().item()
        print(f"ref {a.w}x{a.h}: max|ref| {mx:.3f} std {ref.std().item():.3f}")
        for m in a.modes:
            parts = m.split(":")
            over = {}
            for p in parts[2:]:
                sec, v = p.split("=")
                v = v.split(",")
                over[sec] = (v[0], v[1] if len(v) > 1 else v[0])
            out = forward(x, parts[0], parts[1], over)
            err = (out - ref).abs()
            print(f"{m:28s} max|err|/max|ref| {err.max().item() / mx:.3e}   rms/max {err.pow(2).mean().sqrt().item() / mx:.3e}")


if __name__ == "__main__":
    main()
