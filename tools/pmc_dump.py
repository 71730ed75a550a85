"""Print per-kernel PMC counter averages from rocprofv3 rocpd .db files.  usage: pmc_dump.py a.db [b.db ...]"""
import sqlite3, sys
for path in sys.argv[1:]:
    db = sqlite3.connect(path); cur = db.cursor()
    rows = cur.execute("select name, counter_name, avg(counter_value), count(*), avg(duration) from pmc_events group by name, counter_name").fetchall()
    print("==", path.split('/')[-1])
    for n, c, v, cnt, dur in rows:
        if 'conv' in n and 'reduce' not in n:
            print(f"  {n[:100]:100s} {c:28s} {v:16.1f}  n={cnt} dur={dur/1e3:.1f}us")
