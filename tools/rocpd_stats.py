"""Per-kernel summary (calls, total/avg/min/max duration, % of GPU time) from a rocprofv3 rocpd .db
(`rocprofv3 --kernel-trace --stats` writes SQLite in ROCm 7.2).  Usage: rocpd_stats.py in.db out.csv"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
def short(n):
    n = re.sub(r"\(.*\)$", "", n)
    return n if len(n) < 110 else n[:107] + "..."
with open(sys.argv[2], "w") as f:
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for n, c, t, a, mn, mx in rows:
        f.write(f"\"{short(n)}\",{c},{t},{a:.0f},{100.0*t/tot:.2f},{mn},{mx}\n")
print("columns:", cols)
for n, c, t, a, mn, mx in rows[:25]:
    print(f"{100.0*t/tot:6.2f}% {c:6d} calls avg {a/1e3:9.1f} us  {short(n)[:100]}")
