#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes of scripts/r2_gpu_pmc.sh: per workload and kernel the per-dispatch averages of every
counter, the calibration factors from scripts/pmc_calib.bin, and the calibrated HBM bytes per NN launch.
    python scripts/pmc_collect.py gpurun_out/r2pmc > profiles/r2_pmc_summary.json"""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
N = 64 << 20


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    return {k: {c: x / cnt[(k, c)] for c, x in v.items()} for k, v in agg.items()}


out = {"note": "per-dispatch averages; FETCH_SIZE / WRITE_SIZE in the profiler's own unit (KB), every --pmc set in its own pass without trace domains"}
cal = counters(os.path.join(root, "calib"))
factors = {}
for k, v in cal.items():
    known = {"calib_load16": ("FETCH_SIZE", N * 16), "calib_gather16": ("FETCH_SIZE", N * 16), "calib_store16": ("WRITE_SIZE", N * 16),
             "calib_store4": ("WRITE_SIZE", N * 4), "calib_atomic4": ("WRITE_SIZE", N * 4)}.get(k)
    if known and known[0] in v:
        factors[k] = {"counter": known[0], "reported_KB": v[known[0]], "true_bytes": known[1], "true_over_reported": known[1] / (v[known[0]] * 1024.0)}
    if k == "calib_atomic4" and "FETCH_SIZE" in v:
        factors["calib_atomic4_fetch"] = {"counter": "FETCH_SIZE", "reported_KB": v["FETCH_SIZE"], "true_bytes": N * 4, "true_over_reported": N * 4 / (v["FETCH_SIZE"] * 1024.0)}
out["calibration"] = factors
f_read = factors.get("calib_gather16", factors.get("calib_load16", {})).get("true_over_reported", 2.0)
f_w16 = factors.get("calib_store16", {}).get("true_over_reported", 1.0)
f_w4 = factors.get("calib_store4", {}).get("true_over_reported", 1.0)
out["factors_used"] = {"read": f_read, "write": f_w4, "why": "NN reads are 8-lane groups of consecutive float4 (calib_gather16); its writes are 4-byte and 16-byte "
                       "per-query stores plus histogram atomics: the 4-byte factor is applied to all of WRITE_SIZE (an upper bound if it is the larger one)",
                       "write16": f_w16}
for wl in sorted(os.listdir(root)):
    if wl == "calib" or not os.path.isdir(os.path.join(root, wl)):
        continue
    merged = collections.defaultdict(dict)
    for p in sorted(os.listdir(os.path.join(root, wl))):
        for k, v in counters(os.path.join(root, wl, p)).items():
            merged[k].update(v)
    keep = {k: v for k, v in merged.items() if any(t in k for t in ("nn1_", "nnk_ml", "nnk_wg", "accumulate", "solve", "sel2"))}
    for k, v in keep.items():
        if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
            v["hbm_bytes_calibrated"] = v.get("FETCH_SIZE", 0) * 1024 * f_read + v.get("WRITE_SIZE", 0) * 1024 * f_w4
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v and v["TCC_HIT_sum"] + v["TCC_MISS_sum"] > 0:
            v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        if "SQ_WAVE_CYCLES" in v and v.get("SQ_WAVES"):
            v["wave_lifetime_quad_cycles"] = v["SQ_WAVE_CYCLES"] / v["SQ_WAVES"]
            if "SQ_WAIT_ANY" in v:
                v["wait_any_frac"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    out[wl] = keep
print(json.dumps(out, indent=1, sort_keys=True))
