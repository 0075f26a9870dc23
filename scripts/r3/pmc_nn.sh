#!/bin/bash
# usage: pmc_nn.sh <tag> [ENV=val ...] -- SQ counters of the NN kernel of a short p2p bench run (separate --pmc passes, no trace domains)
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT/gpurun_out/r3/pmc_$1; rm -rf $R; mkdir -p $R; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --steps 3 --warmup 1 --chain ${CHAIN:-p2p}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/p$i -- $B > /dev/null 2> $R/p$i.err
done
cd $GRAFT_REPO_ROOT
python - "$R" "$tag" <<'PY'
import collections, csv, glob, os, sys
root, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    if not any(t in k for t in ("nn1_", "nnk_ml", "accumulate", "solve", "sel2")): continue
    d = {c: x / cnt[(k, c)] for c, x in v.items()}
    w = d.get("SQ_WAVES", 0) or 1
    per = {c: round(x / w, 1) for c, x in d.items() if c.startswith("SQ_") and c != "SQ_WAVES"}
    print(tag, k[:60], "waves", round(w), "per-wave:", per, {c: round(x, 1) for c, x in d.items() if not c.startswith("SQ_")})
PY
find $R -name "*.csv" ! -name "*counter_collection.csv" -delete; find $R -name "*.err" -size 0 -delete
