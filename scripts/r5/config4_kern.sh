#!/bin/bash
# kernel trace of the BASELINE config 4 replay: the launches longer than 150 us, in order (what the first scans of a fresh mapper wait for)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4kern; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o t -- python $R/scripts/r5/config4_scans.py > $O/run.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/trace/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["d"] = int(r["End_Timestamp"]) - r["s"]
rows.sort(key=lambda r: r["s"]); t0 = rows[0]["s"]
for r in rows:
    if r["d"] > 150000: print(f'{(r["s"]-t0)/1e6:9.2f} ms {r["d"]/1e3:9.1f} us  {r["Kernel_Name"][:70]}  grid {r.get("Grid_Size_X","?")}')
m = glob.glob("$O/trace/*memory_copy_trace.csv")
if m:
    cr = list(csv.DictReader(open(m[0])))
    for r in cr:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if d > 150000: print(f'{(int(r["Start_Timestamp"])-t0)/1e6:9.2f} ms {d/1e3:9.1f} us  COPY {r.get("Direction","")} {r.get("Bytes", r.get("Size",""))}')
PY
find $O -name "*.csv" -delete
