"""Weight families the split set of RTP_PREC_MIXED was NOT tuned on: test infrastructure for
tests/test_calibration.py.  No trained .caffemodel exists offline; what trained weights differ in from the He-scaled Gaussian
synthetic set is their spectrum, so four families with very different ones:

  student_t            heavy tails: t(3) entries (a few weights 10-30x the rms: the fp16 rounding of a weight tile is dominated by them)
  lognormal_channels   per-output-channel scales spanning 100x (log-normal, sigma 1.15): what batch-norm folding leaves in real nets
  decaying_spectrum    every layer = A diag(r^-0.7) B (rank <= 128) + a Gaussian floor of a quarter of its norm: the power-law singular
                       spectrum of trained VGG-like layers (a few directions carry most of the energy) without being rank-deficient
  seed5                the engine's own generator with another seed (rtp_config.synthetic_seed = 5)

Activations must stay in fp16 range through ~50 sequential layers whatever the family, so each layer is rescaled LSUV-style on one
random frame (torch on the CPU, fp32): pre-activation standard deviation 1 per layer, biases U(-0.1, 0.1)."""
import zlib

import numpy as np

TRUNK = ["conv1_1", "conv1_2", "P", "conv2_1", "conv2_2", "P", "conv3_1", "conv3_2", "conv3_3", "conv3_4", "P", "conv4_1", "conv4_2",
         "conv4_3_CPM", "conv4_4_CPM"]
FAMILIES = ("student_t", "lognormal_channels", "decaying_spectrum")


def raw_weights(family, name, cout, cin, k, seed):
    rs = np.random.RandomState((zlib.crc32(f"{family}/{name}".encode()) ^ seed) & 0x7FFFFFFF)
    fan = cin * k * k
    if family == "student_t":
        w = rs.standard_t(3, size=(cout, fan))
    elif family == "lognormal_channels":
        s = np.exp(rs.randn(cout) * 1.15)
        w = rs.randn(cout, fan) * s[:, None]
    elif family == "decaying_spectrum":
        R = min(cout, fan, 128)
        a, b = rs.randn(cout, R), rs.randn(R, fan)
        w = (a * (np.arange(1, R + 1, dtype=np.float64) ** -0.7)[None]) @ b
        w = w / np.sqrt((w ** 2).mean()) + 0.25 * rs.randn(cout, fan)
    else:
        raise ValueError(family)
    w = w / np.sqrt((w ** 2).mean()) * np.sqrt(2.0 / fan)
    b = rs.uniform(-0.1, 0.1, cout)
    return w.reshape(cout, cin, k, k).astype(np.float32), b.astype(np.float32)


def make(layers, family, seed, x):
    """layers: engine.conv_layers() = [(name, cin, cout, k)].  x: one net input [N][3][H][W] (numpy) used for the per-layer rescale.
    Returns ({name: (w, b)}, final maps [N][C][h][w] of the torch-CPU fp32 forward with those weights, reference channel order)."""
    import torch
    import torch.nn.functional as F
    dims = {n: (cin, cout, k) for n, cin, cout, k in layers}
    out = {}

    def conv(name, t, relu=True):
        cin, cout, k = dims[name]
        w, b = raw_weights(family, name, cout, cin, k, seed)
        wt = torch.from_numpy(w)
        y = F.conv2d(t, wt, None, padding=k // 2)
        sd = float(y.std())
        scale = np.float32(1.0 / max(sd, 1e-12))
        w = (w * scale).astype(np.float32)
        out[name] = (w, b)
        y = y * float(scale) + torch.from_numpy(b)[None, :, None, None]
        return F.relu_(y) if relu else y

    nstage = max(int(n.split("_stage")[1].split("_")[0]) for n in dims if n.startswith("Mconv"))
    with torch.no_grad():
        t = torch.from_numpy(np.ascontiguousarray(x, np.float32))
        for nm in TRUNK:
            t = F.max_pool2d(t, 2, 2, ceil_mode=True) if nm == "P" else conv(nm, t)
        feat = t
        br = {}
        for L in (1, 2):
            t = feat
            for k in range(1, 6):
                t = conv(f"conv5_{k}_CPM_L{L}", t, relu=k < 5)
            br[L] = t
        for st in range(2, nstage + 1):
            cat = torch.cat([br[1], br[2], feat], 1)
            nb = {}
            for L in (1, 2):
                t = cat
                for k in range(1, 8):
                    t = conv(f"Mconv{k}_stage{st}_L{L}", t, relu=k < 7)
                nb[L] = t
            br = nb
        final = torch.cat([br[2], br[1]], 1).numpy()   # concat_stage7: heat maps first, PAFs second
    return out, final
