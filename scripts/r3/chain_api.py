#!/usr/bin/env python
"""HIP API calls of the last map update (rocprofv3 --hip-trace CSV of scripts/r2_chain_bench.py) that took longer than 15 us, in order,
with the time since the previous listed call.  usage: chain_api.py <dir>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*hip_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
rows.sort()
# the last update: from the last hipGraphLaunch-free stretch ... take the calls after the last hipGraphLaunch (segment graph of the last registration)
gl = [i for i, r in enumerate(rows) if r[2] == "hipGraphLaunch"]
start = gl[-1] if gl else 0
seg = [r for r in rows[start:] if r[2] not in ("hipStreamQuery", "__hipPushCallConfiguration", "__hipPopCallConfiguration", "hipGetLastError")]
t0 = seg[0][0]
print("calls", len(seg), "span %.1f us" % ((seg[-1][1] - t0) / 1e3))
import collections
agg = collections.Counter(); cnt = collections.Counter()
for s, e, n in seg: agg[n] += e - s; cnt[n] += 1
for n, v in agg.most_common(8): print("  %-28s x%-4d %8.1f us" % (n, cnt[n], v / 1e3))
for s, e, n in seg:
    if e - s > 15000: print("  +%8.1f us  %-24s %7.1f us" % ((s - t0) / 1e3, n, (e - s) / 1e3))
