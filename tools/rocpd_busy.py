"""GPU busy fraction of the steady part of a rocprofv3 kernel trace (rocpd .db): the union of all kernel intervals (any stream) over the
last `frac` of the trace's time span, the idle gaps in it by size class, and which kernels follow the long gaps.
Usage: rocpd_busy.py in.db [frac=0.35]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
rows = db.execute("select start, end, name from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
w0 = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[1] > w0]
busy, cur_end, gaps = 0, w0, []
for s, e, n in rows:
    s = max(s, w0)
    if s > cur_end:
        gaps.append((s - cur_end, n))
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
span = t1 - w0
print(f"window {span/1e6:.1f} ms, kernels {len(rows)}, busy {busy/1e6:.1f} ms = {busy/span:.4f}; idle {(span-busy)/1e6:.2f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 20e3), (20e3, 100e3), (100e3, 1e12)):
    g = [x for x, _ in gaps if lo <= x < hi]
    print(f"  gaps {lo/1e3:6.0f}-{hi/1e3:.0f} us: {len(g):6d}, {sum(g)/1e6:8.3f} ms")
c = collections.Counter()
for x, n in gaps:
    if x >= 5e3:
        c[n.split('(')[0][:70]] += x
for n, x in c.most_common(12):
    print(f"  after gaps >= 5 us: {x/1e6:7.3f} ms before {n}")
