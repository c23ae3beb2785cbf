#!/usr/bin/env python
"""HBM traffic of EVERY launch of the conv plan, in plan order, from rocprofv3 --pmc passes over tools/run_batches.py (whole batches one at a time):
   python tools/pmc_steps.py FETCH_DIR WRITE_DIR B N_SCALES PREC [coco|mpi]
Per plan step: mean FETCH_SIZE / WRITE_SIZE of its dispatches (KiB; FETCH x2 = the guide's gfx950 correction, as in bench.py pmc_traffic), the bytes of the fp16
tensors the step must move at least (input + weights + output), the dispatch duration under the counter pass, GB/s against the 8 TB/s HBM peak.  Dispatches
are matched to steps by their position between two conv1_1 launches; batches whose launch count differs from the plan are skipped (engine creation's dry run)."""
import csv
import glob
import os
import re
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caffe_rtpose_amd as r  # noqa: E402

fetch_dir, write_dir = sys.argv[1], sys.argv[2]
B, N = int(sys.argv[3]), int(sys.argv[4])
prec = {"fp32": r.PREC_FP32, "mixed": r.PREC_MIXED, "f16x3": r.PREC_F16X3, "fp16": r.PREC_FP16}[sys.argv[5]]
model = sys.argv[6] if len(sys.argv) > 6 else "coco"
kw = dict(model=r.MODEL_MPI_15, net_w=496, net_h=368) if model == "mpi" else {}
cfg = r.Config(precision=prec, num_scales=N, scale_gap=0.15 if N > 1 else 0.3, frames_in_flight=B, batch_frames=B, **kw)
plan = [ln for ln in r.plan_summary(cfg).splitlines() if ln.startswith("step ")]
W, H = (496, 368) if model == "mpi" else (656, 368)
PLAN_KERNELS = ("conv_first_kernel", "conv_ring_kernel", "conv_pw2_kernel", "conv_igemm", "maxpool", "pack_input")


def per_step(root, counter):
    rows = []
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for i, row in enumerate(csv.DictReader(open(f))):
            if row["Counter_Name"] != counter or not any(k in row["Kernel_Name"] for k in PLAN_KERNELS):
                continue
            t0, t1 = float(row.get("Start_Timestamp") or 0), float(row.get("End_Timestamp") or 0)
            rows.append((int(row.get("Dispatch_Id") or i), row["Kernel_Name"], float(row["Counter_Value"]), t1 - t0))
    rows.sort()
    batches, cur = [], None
    for _, name, val, dur in rows:
        if "conv_first_kernel" in name or "pack_input" in name:
            if cur:
                batches.append(cur)
            cur = []
        if cur is not None:
            cur.append((name, val, dur))
    if cur:
        batches.append(cur)
    good = [b for b in batches if len(b) == len(plan)]
    out = []
    for si in range(len(plan)):
        vals = [b[si][1] for b in good]
        durs = [b[si][2] for b in good if b[si][2] > 0]
        out.append((sum(vals) / max(len(vals), 1), sum(durs) / max(len(durs), 1) if durs else 0.0, good[0][si][0] if good else ""))
    return out, len(good), len(batches)


def min_bytes(ln):
    """fp16 input + fp16 weights + fp16 output of the step's layers (the q / fp8 copies of compensated layers and halo re-reads are extra)."""
    m = re.match(r"step (\w+) (.*?) k (\d+) cin(?:_p)? (\d+) (?:mid (\d+) )?cout (\d+)", ln)
    if not m:
        return 0.0
    kind, names, k, cin, mid, cout = m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), m.group(5), int(m.group(6))
    n0 = re.findall(r"[A-Za-z0-9_]+", names)[0]
    lvl = 0 if n0.startswith("conv1_") else 1 if n0.startswith("conv2_") else 2 if n0.startswith("conv3_") else 3
    px = (H >> lvl) * (W >> lvl) * B * N
    pair = 2 if " + " in names.split("->")[0] else 1
    shared_in = 1 if (pair == 1 or n0.startswith(("Mconv1_", "conv5_1_"))) else 2     # both branches read the same blob (concat / conv4_4_CPM) or one each
    if kind == "first":
        return px * (3 * 4 + 64 * 2)
    if kind == "pw2":
        mid_c = int(mid)
        return px * (pair * cin * 2 + 57 * 2) + pair * (cin * mid_c + mid_c * cout) * 2   # middle blob stays in LDS; the two maps (38 + 19 channels) as a concat slice
    out_px = px // 4 if "+pool" in ln else px
    return shared_in * px * cin * 2 + pair * out_px * cout * 2 + pair * cout * cin * k * k * 2


fs, ng, nbt = per_step(fetch_dir, "FETCH_SIZE")
ws, ng2, _ = per_step(write_dir, "WRITE_SIZE")
# optional: argv[7] / argv[8] = directories of SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES passes -> chip-wide MFMA pipe utilisation per launch
# (MFMA_BUSY sums the 1024 SIMDs' busy cycles, SQ_BUSY the 32 shader engines' cycles with the launch resident)
mf = per_step(sys.argv[7], "SQ_VALU_MFMA_BUSY_CYCLES")[0] if len(sys.argv) > 8 else None
sq = per_step(sys.argv[8], "SQ_BUSY_CYCLES")[0] if len(sys.argv) > 8 else None
print(f"# {ng} / {ng2} batches of {nbt} matched the plan's {len(plan)} launches (B={B}, scales={N}, {sys.argv[5]}, {model}); FETCH_SIZE x2 + WRITE_SIZE, KiB -> MB")
print(f"# {'fetch MB':>9s} {'write MB':>9s} {'total MB':>9s} {'fp16 min MB':>11s} {'ratio':>6s} {'us(pmc)':>8s} {'GB/s':>7s} {'of 8 TB/s':>9s}{' mfma_busy' if mf else ''}  step")
tot = [0.0, 0.0, 0.0, 0.0]
for si, (ln, (f, d, _), (w, d2, _)) in enumerate(zip(plan, fs, ws)):
    fb, wb = 2 * f * 1024, w * 1024
    mb = min_bytes(ln)
    us = d / 1e3 if d else 0.0
    gbs = (fb + wb) / (us * 1e-6) / 1e9 if us else 0.0
    tot[0] += fb; tot[1] += wb; tot[2] += mb; tot[3] += us
    util = f" {mf[si][0] / (32.0 * sq[si][0]):9.3f}" if mf and sq and sq[si][0] > 0 else ""
    print(f"  {fb / 1e6:9.2f} {wb / 1e6:9.2f} {(fb + wb) / 1e6:9.2f} {mb / 1e6:11.2f} {(fb + wb) / mb if mb else 0:6.2f} {us:8.1f} {gbs:7.0f} {gbs / 8000:9.3f}{util}  {ln[5:110]}")
print(f"# batch: fetch {tot[0] / 1e6:.1f} MB + write {tot[1] / 1e6:.1f} MB = {(tot[0] + tot[1]) / 1e6:.1f} MB against {tot[2] / 1e6:.1f} MB of fp16 tensors; {tot[3]:.0f} us of launches under the counter pass")
