#!/bin/bash
# Round-6 counter collection: FETCH / WRITE / TCC / SQ for the NN kernel on every benchmark chain, the 10 M map and the batch; every
# --pmc set in its own rocprofv3 run, no trace domains.  Writes gpurun_out/r6pmc/summary.json and gpurun_out/r6pmc/nn_traffic.json.
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT/gpurun_out/r6pmc; rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp
pmc() { local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 600 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $R/$name -- "$@" > /dev/null 2>$R/$(echo $name | tr / _).err; }
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --steps 3 --warmup 1 --min-seconds 0"
pmc calib/fetch FETCH_SIZE -- $GRAFT_REPO_ROOT/scripts/pmc_calib.bin
pmc calib/write WRITE_SIZE -- $GRAFT_REPO_ROOT/scripts/pmc_calib.bin
for wl in "p2p:--chain p2p" "p2plane:--chain p2plane" "knn6:--chain docs_knn6" "map10M:--chain p2p --map-points 10000000 --scale 3.16" "batch8:--chain p2p --batch 8"; do
  if [ -n "$WL" ] && [ "${wl%%:*}" != "$WL" ]; then continue; fi
  name=${wl%%:*}; args=${wl#*:}
  pmc $name/fetch FETCH_SIZE -- $B $args
  pmc $name/write WRITE_SIZE -- $B $args
  pmc $name/tcc TCC_HIT_sum TCC_MISS_sum -- $B $args
  pmc $name/sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -- $B $args
  pmc $name/sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS -- $B $args
  pmc $name/tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -- $B $args
done
find $R -name "*.csv" ! -name "*counter_collection.csv" -delete; find $R -name "*.err" -size 0 -delete
cd $GRAFT_REPO_ROOT && python scripts/pmc_collect.py gpurun_out/r6pmc > gpurun_out/r6pmc/summary.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6pmc/summary.json"))
out = {"source": "profiles/r6_pmc_summary.json (scripts/r6_pmc.sh: rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE, TCC, TCP and SQ sets, each in its own pass, no trace domains; `python bench.py --no-cpu --no-extras --steps 3 --warmup 1` per workload)",
       "calibration": d.get("factors_used"),
       "per_launch": "average over the 20 NN launches of a registration (nn1_wg_kernel for k = 1; knn 6: nnk_wg_kernel, 18 of the 20 launches -- the two nnk_ml_kernel launches of iterations 0 and 1 are NOT in this figure); the bench's roofline.avg_launch_us is the same average",
       "l2_hit_rate": {}, "wave": {}}
keys = {"p2p": "hbm_bytes_per_launch", "p2plane": "hbm_bytes_per_launch_p2plane", "knn6": "hbm_bytes_per_launch_knn6", "map10M": "hbm_bytes_per_launch_10M", "batch8": "hbm_bytes_per_launch_batch8"}
for wl, key in keys.items():
    ks = d.get(wl, {})
    nn = [(k, v) for k, v in ks.items() if k.startswith("nn1_") or k.startswith("nnk_ml") or k.startswith("nnk_wg")]
    if not nn: continue
    # the dominant NN instantiation of the workload: the one with the most waves x launches is what the average is made of
    k, v = max(nn, key=lambda kv: kv[1].get("SQ_WAVES", 0))
    wgk = [kv for kv in nn if kv[0].startswith("nnk_wg")]
    if wgk: k, v = wgk[0]     # knn > 1: the kernel of the 18 seeded launches (nnk_ml_kernel has more waves per launch but runs twice)
    if "hbm_bytes_calibrated" in v: out[key] = int(v["hbm_bytes_calibrated"])
    if wl == "knn6": # the bench's average is over the step's 20 launches: 2 x nnk_ml_kernel (iterations 0 and 1) + 18 x nnk_wg_kernel
        wg = [v2 for k2, v2 in nn if k2.startswith("nnk_wg") and "hbm_bytes_calibrated" in v2]
        ml = [v2 for k2, v2 in nn if k2.startswith("nnk_ml") and "hbm_bytes_calibrated" in v2]
        if wg and ml:
            out[key] = int((18 * wg[0]["hbm_bytes_calibrated"] + 2 * ml[0]["hbm_bytes_calibrated"]) / 20)
            out["knn6_split"] = {"nnk_wg_kernel": int(wg[0]["hbm_bytes_calibrated"]), "nnk_ml_kernel": int(ml[0]["hbm_bytes_calibrated"]), "weights": [18, 2]}
    if "l2_hit_rate" in v: out["l2_hit_rate"][wl] = round(v["l2_hit_rate"], 3)
    out["wave"][wl] = {"kernel": k, "waves": v.get("SQ_WAVES"), "lifetime_quad_cycles": round(v.get("wave_lifetime_quad_cycles", 0)),
                       "wait_any_frac": round(v.get("wait_any_frac", 0), 3),
                       "valu_insts_per_wave": round(v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_WAVES", 1), 1)),
                       "salu_insts_per_wave": round(v.get("SQ_INSTS_SALU", 0) / max(v.get("SQ_WAVES", 1), 1)),
                       "lds_insts_per_wave": round(v.get("SQ_INSTS_LDS", 0) / max(v.get("SQ_WAVES", 1), 1)),
                       "tcp_accesses": v.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "tcp_to_l2_read_requests": v.get("TCP_TCC_READ_REQ_sum")}
    print(wl, k, out.get(key), out["wave"][wl])
import sys
sys.path.insert(0, ".")
import bench
out["kernel_sources_sha"] = bench.kernel_sources_sha()   # bench.py prints roofline.traffic only while this matches the tree it runs from
json.dump(out, open("gpurun_out/r6pmc/nn_traffic.json", "w"), indent=1)
PY
find gpurun_out/r6pmc -name "*.csv" -delete; find gpurun_out/r6pmc -type d -empty -delete   # (the raw per-dispatch counter files: hundreds of MB; the summaries stay)
du -sh gpurun_out/r6pmc
