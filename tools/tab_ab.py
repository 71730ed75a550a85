"""Tabulate tools/bench_conv.py logs taken under several VQ_TILE modes.  usage: tab_ab.py LOG"""
import re, sys
rows = {}; mode = None; order = []; modes = []
for l in open(sys.argv[1]):
    m = re.match(r'=== VQ_TILE=(\d+)', l)
    if m:
        mode = int(m.group(1)); modes.append(mode); continue
    m = re.search(r'B=\d+\s+(\d+)->\s*(\d+) @\s*(\d+) k(\d) up(\d): fwd\s+([\d.]+) ms\s+([\d.]+) TF \| dgrad\s+([\d.]+) ms\s+([\d.]+) TF', l)
    if m:
        k = tuple(m.group(i) for i in range(1, 6))
        if k not in rows:
            rows[k] = {}; order.append(k)
        rows[k][mode] = (float(m.group(7)), float(m.group(9)))
print("shape (fwd/dgrad TF)".ljust(22) + "".join(f"{m:>11d}" for m in modes))
for k in order:
    print("-".join(k).ljust(22) + "".join(f"{rows[k].get(m, (0, 0))[0]:6.0f}/{rows[k].get(m, (0, 0))[1]:4.0f}" for m in modes))
