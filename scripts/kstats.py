#!/usr/bin/env python
"""Print a compact per-kernel table from a rocprofv3 *_kernel_stats.csv (names shortened)."""
import csv, glob, os, re, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
for r in rows:
    name = r["Name"]
    m = re.search(r"([A-Za-z0-9_]+(<[^>(]*>)?)\(", name)
    short = m.group(1) if m else name[:40]
    print(f"{short:38s} calls={int(r['Calls']):6d} avg={float(r['AverageNs'])/1e3:9.2f}us min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:8.2f} pct={float(r['Percentage']):6.2f}")
