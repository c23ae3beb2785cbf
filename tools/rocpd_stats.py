#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel stats table."""
import re
import sqlite3
import sys


def short(name):
    m = re.match(r"_ZN3rtp(\d+)(conv_\w+_kernel)I(DF16_|f)((?:Li\d+E)+)EEvNS_10ConvParamsE", name)
    if m:
        a = re.findall(r"Li(\d+)E", m.group(4))
        t = "f16" if m.group(3) == "DF16_" else "f32"
        tail = f",SB{a[7]}" if len(a) > 7 else ""
        return f"{m.group(2)[:-7]}<{t},{a[0]}x{a[1]},w{a[2]}x{a[3]},ksplit{a[4]},k{a[5]},chb{a[6]}{tail}>"
    m = re.match(r"_ZN3rtp\d+(\w+?)I(DF16_|f)", name)
    if m:
        return f"{m.group(1)}<{'f16' if m.group(2) == 'DF16_' else 'f32'}>"
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
