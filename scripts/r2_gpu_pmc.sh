#!/bin/bash
# round-2 counter collection on HEAD: FETCH / WRITE / TCC / SQ for the NN kernel on every benchmark chain, the 10 M map and the batch
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT/gpurun_out/r2pmc; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp
pmc() { # name, counters..., -- command
  local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 600 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $R/$name -- "$@" > /dev/null 2>$R/$(echo $name | tr / _).err
}
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --steps 3 --warmup 1"
pmc calib/fetch FETCH_SIZE -- $GRAFT_REPO_ROOT/scripts/pmc_calib.bin
pmc calib/write WRITE_SIZE -- $GRAFT_REPO_ROOT/scripts/pmc_calib.bin
for wl in "p2p:--chain p2p" "p2plane:--chain p2plane" "knn6:--chain docs_knn6" "map10M:--chain p2p --map-points 10000000 --scale 3.16" "batch8:--chain p2p --batch 8"; do
  name=${wl%%:*}; args=${wl#*:}
  pmc $name/fetch FETCH_SIZE -- $B $args
  pmc $name/write WRITE_SIZE -- $B $args
  pmc $name/tcc TCC_HIT_sum TCC_MISS_sum -- $B $args
  pmc $name/sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -- $B $args
done
find $R -name "*.csv" ! -name "*counter_collection.csv" -delete; find $R -name "*.err" -size 0 -delete
cd $GRAFT_REPO_ROOT && python scripts/pmc_collect.py gpurun_out/r2pmc > gpurun_out/r2pmc/summary.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2pmc/summary.json"))
print(json.dumps(d["calibration"], indent=1)); print(d["factors_used"])
for wl in d:
    if wl in ("calibration","factors_used","note"): continue
    for k,v in d[wl].items():
        if "nn" in k: print(wl, k, {c: round(x, 3) for c, x in v.items()})
PY
du -sh gpurun_out/r2pmc
