#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel stats table."""
import re
import sqlite3
import sys


def short(name):
    m = re.match(r"void rtp::conv_igemm_kernel<(.*?)>\(", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        return f"conv_igemm<{a[0]},BM{a[1]},BN{a[2]},W{a[3]}x{a[4]},KS{a[5]},k{a[6]},rowb{a[7]}>"
    return re.sub(r"\(.*", "", name)[:90]


db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, (end - start) from kernels").fetchall()
agg = {}
for name, d in rows:
    k = short(name)
    a = agg.setdefault(k, [0, 0, 1 << 62, 0])
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print(f"{'kernel':78s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:78s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100.0 * a[1] / tot:6.2f}")
print(f"total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches")
