#!/usr/bin/env python
"""GPU occupancy over time from a rocprofv3 kernel trace (rocpd sqlite or *_kernel_trace.csv):
union-busy time vs wall, concurrency histogram, per-kernel totals inside the steady-state window."""
import csv
import re
import sqlite3
import sys


def load(path):
    if path.endswith(".csv"):
        rows = []
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0")))
        return rows
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
    cols = [r[1] for r in db.execute(f"pragma table_info({kt[0]})")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    return [(n, s, e, str(qq)) for n, s, e, qq in db.execute(f"select name, start, end, {q} from {kt[0]}")]


rows = sorted(load(sys.argv[1]), key=lambda r: r[1])
lo_frac, hi_frac = 0.3, 0.9
t0, t1 = rows[0][1], max(r[2] for r in rows)
a, b = t0 + (t1 - t0) * lo_frac, t0 + (t1 - t0) * hi_frac
win = [r for r in rows if r[1] >= a and r[2] <= b]
ev = []
for n, s, e, q in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; cur = 0; last = a; hist = {}
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    if cur > 0:
        busy += t - last
    cur += d; last = t
wall = b - a
print(f"window {wall / 1e6:.2f} ms, {len(win)} dispatches, busy(union) {100 * busy / wall:.1f}%, sum of durations {sum(e - s for _, s, e, _ in win) / wall:.2f}x wall")
print("concurrency histogram (kernels in flight : % of wall):", {k: round(100 * v / wall, 1) for k, v in sorted(hist.items())})
agg = {}
for n, s, e, q in win:
    k = re.sub(r"\(.*", "", n)
    m = re.match(r"_ZN3rtp\d+(conv_\w+?)_kernelI(DF16_|f)((?:Li\d+E)+)", n)
    if m:
        p = re.findall(r"Li(\d+)E", m.group(3))
        k = f"{m.group(1)}<{p[0]}x{p[1]},k{p[5]},chb{p[6]}>"
    x = agg.setdefault(k, [0, 0]); x[0] += 1; x[1] += e - s
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {100 * d / wall:6.1f}% of wall  {c:6d} calls  {d / c / 1e3:8.1f} us avg  {k[:80]}")
print("queues:", sorted({q for *_, q in win}))
