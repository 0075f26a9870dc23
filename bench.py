#!/usr/bin/env python
"""bench.py -- ICP iterations/s, 100k-point scan vs 1M-point map (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--chain p2p|p2plane|docs_knn6] [--no-extras] [--no-cpu]

One "step" is one registration of the hot path in throughput mode: a fixed 20 ICP iterations
(Counter checker only) of one synthetic 100k-point scan against the 1M-point map,
with the scan already resident in HBM.  value = ICP iterations executed by all ranks / wall time
(max over ranks, barrier + device sync on both sides).  Ranks hold a replica of the map and their own
scan (independent seeds): the path shards across scans with no data-path collective => weak scaling.

Extra objects on the JSON line (rank 0):
  roofline     -- the NN kernel of the headline workload: algorithmic bytes per launch (N*16 query read +
                  M*16 map read + N*8 result write) / mean launch duration measured with HIP
                  events on the library's stream (profile mode), against the 8 TB/s HBM peak.
  step_ms      -- min / median / max over the timed steps (every step ends with a device sync).
  chains       -- (N = 1 only) the other BASELINE configurations measured the same way, each with its own
                  value / step_ms / roofline: `p2plane_filter_normals` (config 3: point-to-plane against normals
                  from SurfaceNormalDataPointsFilter{knn 10} run on the map), `map_10M` (config 5's map on one GPU:
                  10 M points, same density), `batch8` (8 readings through icpmi_register_batch_dev: aggregate
                  throughput of one GPU serving 8 scan streams -- NOT the headline, which is one scan).
  cpu_baseline -- the CPU oracle (a port: libpointmatcher is not installed) timed on this host on a
                  bounded sample of the same workload.
  libpointmatcher -- "present" when oracle/_ref/liboracle_pm.so (the real PM::ICPSequence, `make -C oracle oracle_pm`)
                  exists and was used for pose_err_vs_libpointmatcher; "absent" otherwise.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

ITERS_PER_STEP = 20
M_MAP = 1_000_000
N_SCAN = 100_000
HBM_PEAK_GBS = 8000.0

CHAINS = {
    # config 2 of BASELINE.json: point-to-point
    "p2p": dict(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)]),
    # config 3: point-to-plane (analytic normals on the map)
    "p2plane": dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)]),
    # the documented chain (docs/MapperConfiguration.md:174-189): knn 6, point-to-plane, epsilon 0
    "docs_knn6": dict(minimizer=2, knn=6, max_dist=2.0, outliers=[(4, 0.85)]),
}


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def step_stats(ms):
    return {"min": min(ms), "median": statistics.median(ms), "max": max(ms), "n": len(ms)}


def kernel_sources_sha():
    """Stamp of the kernel sources a counter measurement belongs to: profiles/nn_traffic.json carries the stamp of the tree its PMC
    passes ran on, and `roofline.traffic` is printed only while that stamp is the one of the sources in this tree (a
    constant loaded from a file goes stale silently the next time a kernel changes)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "norlab_icp_mapper_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def map_points_in_reach(np, map4, scan4, max_dist):
    """Map points the search can touch at all: inside the reading's bounding box grown by maxDist.  SURVEY 8d charges `M * 16` for
    "read the map once"; for a map the reading covers (configs 2 / 3: the box is the whole room) this IS M, for the 10 M-point map
    of config 5 it is the part of the map a 100 k-point scan can reach -- charging all 160 MB there printed frac > 1 in r3."""
    lo = scan4[:, :3].min(axis=0) - max_dist
    hi = scan4[:, :3].max(axis=0) + max_dist
    inside = np.ones(map4.shape[0], dtype=bool)
    for a in range(3):
        inside &= (map4[:, a] >= lo[a]) & (map4[:, a] <= hi[a])
    return int(np.count_nonzero(inside))


def nn_roofline(pkg, dev, chain, d_map, d_nrm, d_scan, n_scan, m_map, traffic_key=None, m_reach=None, timed_us=None, timed_cnt=0):
    """The NN launch's duration two ways: (a) `timed_us` -- device clocks summed by the loop itself inside the TIMED graph replays (first
    workgroup of the NN kernel -> first workgroup of the kernel behind it: one kernel boundary included, icpmi.h: nn_ms_avg), the number
    `achieved` is computed from when given; (b) a second, eager handle in profile mode with HIP events on the library's stream around every NN
    launch (r1 - r5's number, kept beside it as avg_launch_us_events)."""
    prof = pkg.ICPSequence(device=dev, max_iterations=ITERS_PER_STEP, profile=1, **chain)
    prof.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr())
    nn_ms, nn_cnt = 0.0, 0
    for r in range(3 + 5):
        prof.registerDev(d_scan.data_ptr(), d_scan.shape[0], fixed_iterations=ITERS_PER_STEP)
        if r >= 3:
            nn_ms += prof.stats.nn_ms_avg * prof.stats.nn_launches
            nn_cnt += prof.stats.nn_launches
    nn_avg_ms_events = nn_ms / max(nn_cnt, 1)
    nn_avg_ms = timed_us * 1e-3 if timed_us else nn_avg_ms_events
    kq = chain.get("knn", 1)
    m_alg = m_map if m_reach is None else min(m_map, m_reach)
    alg_bytes = n_scan * 16 + m_alg * 16 + n_scan * 8 * kq
    achieved = alg_bytes / (nn_avg_ms * 1e-3) / 1e9 if nn_avg_ms > 0 else 0.0
    traffic, traffic_note = None, None
    pmc = os.path.join(ROOT, "profiles", "nn_traffic.json")
    if traffic_key and os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            if rec.get("kernel_sources_sha") == kernel_sources_sha():
                traffic = rec.get(traffic_key)
                # (per-dispatch counter collection does not preserve the L2 from one dispatch to the next, so this is the launch
                #  on cold caches -- an upper bound for the production run, where an XCD's map slice survives the kernel boundary)
                traffic_note = "rocprofv3 --pmc per dispatch: caches cold at every launch (upper bound for the graph-replayed loop, DESIGN 13.1)"
            else:
                traffic_note = "profiles/nn_traffic.json was measured on other kernel sources (stamp mismatch): not reported"
        except Exception:
            traffic = None
    del prof
    nn1 = "nn1_wg_kernel"
    # knn > 1: nnk_ml_kernel serves iterations 0 and 1, nnk_wg_kernel the seeded ones (18 of a step's 20 launches); the average is over all
    nnk = "nnk_wg_kernel (+ nnk_ml_kernel, iterations 0-1)"
    return {"bound": "hbm", "kernel": nn1 if kq == 1 else nnk, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
            "map_points_charged": m_alg, "avg_launch_us": nn_avg_ms * 1e3, "launches_timed": timed_cnt if timed_us else nn_cnt,
            "avg_launch_source": ("device clocks inside the timed graph replays (NN first workgroup -> next kernel's first workgroup)" if timed_us
                                  else "HIP events, eager profile-mode handle"),
            "avg_launch_us_events": nn_avg_ms_events * 1e3,
            # what actually limits the launch (DESIGN: counters + L2 experiments of r5): a dependent chain of ~5 memory trips per workgroup over a map
            # that lives in the L2s / Infinity Cache -- not HBM bandwidth; `bound` names the roofline the fraction is priced against
            "limited_by": "latency: dependent chain of memory trips, map resident in L2 / Infinity Cache (wait_any ~0.55); HBM bandwidth is not the limiter",
            **({"traffic_frac": traffic / (nn_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS} if traffic and nn_avg_ms > 0 else {}),
            **({"traffic_note": traffic_note} if traffic_note else {})}


def config4_replay(np, pkg, with_cpu, passes=3, epsilon_approx=False):
    """BASELINE config 4: the 14 bundled scans (tests/golden/bundled_scans_all.npz, the reference's examples/data as a fixture) through the C++ host
    shell's Mapper::processInput with the shipped configuration (examples/config.yaml with epsilon 0 and PointToPlane: knn 6, Counter 10,
    DynamicPoints + Octree 0.15 m modules, SurfaceNormal knn 10 + CutAtDescriptorThreshold post filters, update every scan), map resident on the
    GPU -- the replay tests/test_gpu_configs.py holds to the oracle's, here on the clock (NIM_TIMING: scans preloaded, processInput timed), with
    the oracle's replay (tests/oracle_mapper.py) timed beside it as this configuration's cpu_baseline."""
    import re
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import config4_data as c4
    exe = os.path.join(ROOT, "norlab_icp_mapper_amd", "build_map_from_scans_and_trajectory")
    if not os.path.exists(exe):
        return {"error": "host harness not built (norlab_icp_mapper_amd/build_map_from_scans_and_trajectory)"}
    z = np.load(os.path.join(ROOT, "tests", "golden", "bundled_scans_all.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        names, traj = c4.write_bundled_dataset(tmp, z)
        cfg = os.path.join(tmp, "config.yaml")
        yaml = c4.CONFIG4_YAML
        env4 = dict(os.environ, NIM_TIMING=str(passes))
        if epsilon_approx:     # the shipped `epsilon: 1` as an approximate search (NIM_EPSILON_APPROX: host/IcpSequence.cpp)
            assert "epsilon: 0" in yaml
            yaml = yaml.replace("epsilon: 0", "epsilon: 1")
            env4["NIM_EPSILON_APPROX"] = "1"
        open(cfg, "w").write(yaml)
        run = subprocess.run([exe, tmp, cfg], capture_output=True, text=True, timeout=600, env=env4)
        if run.returncode != 0:
            return {"error": (run.stderr + run.stdout)[-400:]}
        if os.environ.get("ICPMI_SELF_DIAG"):  # (diagnostic lines of the library's self search travel on the harness's stderr)
            sys.stderr.write("".join(l + "\n" for l in run.stderr.splitlines() if "icpmi" in l)[-4000:])
    num = r"([-+0-9.eE]+)"
    reps = [dict(zip(("pass", "scans", "process_ms", "register_ms", "update_ms", "iterations", "scans_per_s"), map(float, m)))
            for m in re.findall(rf"replay: pass {num} scans {num} process_ms {num} register_ms {num} update_ms {num} iterations {num} scans_per_s {num}", run.stdout)]
    per = [tuple(map(float, m)) for m in re.findall(rf"timing: pass {num} scan {num} points {num} process_ms {num} register_ms {num} update_ms {num} iterations {num} map {num}", run.stdout)]
    if not reps:
        return {"error": "no replay line in the harness output: " + run.stdout[-300:]}
    best = min(reps, key=lambda r: r["process_ms"])      # pass 0 also pays the one-time allocations and graph captures
    mine = [r for r in per if r[0] == best["pass"]]
    pts = [r[2] for r in mine]
    out = {
        "config": "BASELINE config 4: full examples/ trajectory replay (14 bundled scans, lexicographic pairing as the reference's harness), growing map, "
                  "C++ Mapper::processInput over the C ABI; shipped chain with epsilon 0 + PointToPlane: KDTree knn 6 maxDist 2.0, Counter 10; "
                  "DynamicPoints + Octree(0.15) modules; SurfaceNormal knn 10 + CutAtDescriptorThreshold post; update every scan (delay 0.05 s)",
        "value": best["scans_per_s"], "unit": "scans/s", "scans": int(best["scans"]), "passes": len(reps), "pass_reported": int(best["pass"]),
        "process_ms_per_pass": [round(r["process_ms"], 3) for r in reps],
        "ms_per_scan": best["process_ms"] / best["scans"],
        "register_ms_per_scan": step_stats([r[4] for r in mine[1:]]),            # (scan 1 creates the map: no registration)
        "update_ms_per_scan": step_stats([r[5] for r in mine]),
        "icp_iterations": int(best["iterations"]), "iterations_per_s": best["iterations"] / (best["process_ms"] * 1e-3),
        "points_per_scan_after_input_filters": {"min": min(pts), "max": max(pts)}, "map_points_final": int(mine[-1][7]) if mine else None,
        "what_is_timed": "processInput only (registration on the resident map + the map update it starts); VTK parsing and the input filters run before the clock",
    }
    if with_cpu:
        import oracle_mapper as om
        nt = min(16, len(os.sched_getaffinity(0)))
        a_icp, a_mod, a_kw = c4.oracle_mapper_args(nt)
        mapper = om.OracleMapper(a_icp, a_mod, **a_kw)
        clouds = [mapper.apply_input_filters(z[f"scan{i}_xyz"]) for i in range(len(names))]
        t_each = []
        for i in range(len(names)):
            t0 = time.perf_counter()
            mapper.process_input(clouds[i], c4.quat_T(traj[i, 2:]), traj[i, 0] + traj[i, 1] * 1e-9)
            t_each.append((time.perf_counter() - t0) * 1e3)
        cpu_s = sum(t_each) * 1e-3
        out["cpu_baseline"] = {"value": len(names) / cpu_s, "unit": "scans/s", "cores": nt, "kind": "port",
                               "sample": f"the oracle's replay of the same 14 scans (tests/oracle_mapper.py over oracle/liboracle.so, {nt} OpenMP threads in the "
                                         f"kNN / normals loops; the module chain's bookkeeping is numpy): {cpu_s:.2f} s of process_input",
                               "ms_per_scan": step_stats(t_each), "map_points_final": int(mapper.map["xyz1"].shape[0])}
        out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def time_registrations(torch, icp, d_scan, steps, warmup):
    def step():
        return icp.registerDev(d_scan.data_ptr(), d_scan.shape[0], fixed_iterations=ITERS_PER_STEP)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    per = []
    t_all = time.perf_counter()
    for _ in range(steps):
        t0 = time.perf_counter()
        T = step()
        per.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    return T, time.perf_counter() - t_all, per


def minstd_random_sampling_keep(np, n, prob=0.75, seed=1):
    """RandomSamplingDataPointsFilter{prob, randomSamplingMethod 0, seed} as host/IcpSequence.cpp evaluates it (the filter of
    PM::ICPSequence::setDefault()): std::minstd_rand (x <- 48271 x mod 2^31 - 1), one number float(x) / 2147483645.0f per point, kept iff
    number < prob, never more than floor(n prob) + 1 points.  (r3 fed the default chain a numpy default_rng stand-in of the same size.)"""
    x = seed % 2147483647 or 1
    xs = np.empty(n, dtype=np.float32)
    for i in range(n):
        x = (x * 48271) % 2147483647
        xs[i] = x
    keep = (xs / np.float32(2147483645.0)) < np.float32(prob)
    n_out = int(np.float32(n) * np.float32(prob))
    over = np.nonzero(np.cumsum(keep) > n_out + 1)[0]
    if over.size:
        keep[over[0]:] = False
    return keep


def circle_scans(pkg, n_scans, n_points, scale, rank, radius=20.0):
    """BASELINE config 5's streams: scan s of rank r is seeded 100 + 1000 r + s and taken from a sensor on a circle of
    radius 20 m; each is an independent sample of the scene's surfaces within 60 m of its sensor, moved by T_gt^-1 (identity prior)."""
    import math as _m
    out = []
    for sidx in range(n_scans):
        th = 2.0 * _m.pi * (sidx + 0.37 * rank) / max(n_scans, 1)
        sc = pkg.synth.make_scene(m=8, n=n_points, scale=scale, seed_scan=100 + 1000 * rank + sidx,
                                  sensor=(radius * _m.cos(th), radius * _m.sin(th), 1.5))
        out.append(sc["scan"])
    return out


def config5_stream(np, torch, pkg, dev, d_map, d_nrm, d_scans, chain, min_dist, comm, barrier, normals_knn=0, cell_size=20.0):
    """One rank's part of BASELINE config 5: every scan is registered against the shared map (Counter 40 + Differential: what
    Mapper::processInput runs) and followed by ONE map-growth epoch (icpmi_staged_merge_allgather: PointDistance accept against the
    resident map, all-gather of the accepted points, rank-ordered merge, append, incremental index insert on every replica; r6: the merged
    set binned into the mapper's 20 m cells on the device and appended to the replica's cell log, icpmi_staged_bin_cells -- Map.cpp:206-229 /
    RAMCellManager.cpp:13-16) -- registration AND epoch inside the timed region.  normals_knn > 0 (a point-to-plane replica): the epoch also
    recomputes the map's normals, SurfaceNormalDataPointsFilter over the grown map as Map.cpp:524 applies it on every update.
    comm: None (single rank), ("rccl", id, world, rank) or ("loopback", R, shift)."""
    import os as _os
    icp = pkg.ICPSequence(device=dev, max_iterations=40, use_differential=1, **chain)
    assert icp.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr() if d_nrm is not None else None)
    if comm is not None and comm[0] == "loopback":
        _os.environ["ICPMI_COMM_LOOPBACK"] = str(comm[1]); _os.environ["ICPMI_COMM_LOOPBACK_SHIFT"] = repr(comm[2])
        try:
            icp.commInit(bytes(128), 1, 0)
        finally:
            del _os.environ["ICPMI_COMM_LOOPBACK"]; del _os.environ["ICPMI_COMM_LOOPBACK_SHIFT"]
    elif comm is not None and comm[0] == "rccl":
        icp.commInit(comm[1], comm[2], comm[3])
    icp.cellLogConfigure(cell_size)      # the epoch enqueues the binning of its merged set itself; stagedBinCells collects the table
    eye = np.eye(4, dtype=np.float32)
    # warm-up: one scan's registration + epoch on a throw-away handle state would grow the map; instead the first scan is registered once
    # untimed (graphs, allocations) and its epoch is left to the timed loop
    icp.registerWithPriorDev(d_scans[0].data_ptr(), d_scans[0].shape[0], eye)
    icp.stageDiscard()
    barrier()
    reg_ms, ep_ms, bin_ms, its, accepted, appended, cells_binned = [], [], [], 0, [], [], []
    t0 = time.perf_counter()
    for d in d_scans:
        ta = time.perf_counter()
        corr = icp.registerWithPriorDev(d.data_ptr(), d.shape[0], eye)
        its += icp.stats.iterations
        tb = time.perf_counter()
        mine, app, new_m = icp.stagedMergeAllGather(corr, min_dist, normals_knn=normals_knn)
        tbin = time.perf_counter()
        ijk, _off, _cnt = icp.stagedBinCells(cell_size)
        tc = time.perf_counter()
        reg_ms.append((tb - ta) * 1e3); ep_ms.append((tc - tb) * 1e3); bin_ms.append((tc - tbin) * 1e3); accepted.append(mine); appended.append(app)
        cells_binned.append(int(ijk.shape[0]))
    barrier()
    elapsed = time.perf_counter() - t0
    cr, cme, ckind = icp.commInfo()
    dbg = icp.debugCounters()
    fast_ep, slow_ep = int(dbg[14]), int(dbg[15])
    res = {"scans": len(d_scans), "elapsed_s": elapsed, "iterations": its, "register_ms": step_stats(reg_ms), "merge_epoch_ms": step_stats(ep_ms),
           "accepted_per_scan_this_rank": accepted, "appended_per_epoch_all_ranks": appended, "map_points_after": new_m,
           "cells_binned_per_epoch": cells_binned, "cell_binning_ms": step_stats(bin_ms), "cell_log_points": icp.cellLogSize(),
           "cell_binning": f"on the device, inside merge_epoch_ms: icpmi_staged_bin_cells({cell_size:g} m) appends the merged set cell by cell to the replica's "
                           "cell log; the host receives {ijk, offset, count} per touched cell",
           "normals_knn_in_epoch": normals_knn,
           "rccl_ranks": cr, "rccl_rank": cme, "communicator": {0: "none", 1: "rccl", 2: "loopback"}[ckind],
           # r5: what one epoch costs in exchanges (csrc/ops.hip: merge_epoch_one_collective)
           "epochs_one_collective": fast_ep, "epochs_three_collectives": slow_ep,
           "collectives_per_epoch": (fast_ep * 1 + slow_ep * 3) / max(fast_ep + slow_ep, 1),
           "host_waits_before_the_merge_per_epoch": (fast_ep * 1 + slow_ep * 3) / max(fast_ep + slow_ep, 1),
           "epoch_exchange": "one ncclAllGather of fixed-size blocks {count header, <= 32768 accepted points} per scan; the R counts reach the host "
                             "through host-mapped memory behind it (one stream wait); a block overflow falls back to count + ready + points"}
    if comm is not None:
        icp.commDestroy()
    del icp
    return res


def dry_launch(args):
    """The launcher path without GPUs (tests/test_bench_launch.py): every rank joins a gloo group, the ranks are counted by an
    all-reduce and rank 0 prints the one JSON line."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    seen = 1
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
        ranks = [None] * world
        dist.all_gather_object(ranks, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid()})
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranks = [{"rank": 0, "local_rank": 0, "pid": os.getpid()}]
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_seen": seen, "ranks": ranks, "requested_gpus": args.gpus}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # 0.2 s of timed region at ~1 ms per registration: long enough for an outside sampler to see
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the timed --steps block until this much wall time has passed (median block reported)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--chain", default="p2p", choices=list(CHAINS))
    ap.add_argument("--map-points", type=int, default=M_MAP)
    ap.add_argument("--scan-points", type=int, default=N_SCAN)
    ap.add_argument("--scale", type=float, default=1.0,
                    help="scene extent factor (SURVEY.md 8d, config 5: --map-points 10000000 --scale 3.16 keeps the point density of config 2)")
    ap.add_argument("--normals", default="analytic", choices=["analytic", "filter"],
                    help="map normals of the point-to-plane chains: the scene's analytic ones, or SurfaceNormalDataPointsFilter{knn: 10} "
                         "run on the map through icpmi_surface_normals (BASELINE config 3)")
    ap.add_argument("--batch", type=int, default=0, help="headline through icpmi_register_batch_dev with this many readings (0: single registration)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the `chains` object (configs 3 / 5 and the batch of 8)")
    ap.add_argument("--cpu-iters", type=int, default=10, help="iterations of the single-threaded cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work of the multi-threaded cpu_baseline leg")
    ap.add_argument("--workload", default="registration", choices=["registration", "config5"],
                    help="registration (default, the north-star headline): independent 20-iteration registrations; config5: BASELINE config 5 -- "
                         "every rank streams --scans scans against the shared 10 M-point map, one RCCL merge epoch per scan INSIDE the timed region")
    ap.add_argument("--scans", type=int, default=8, help="scans per rank of --workload config5")
    ap.add_argument("--epoch-normals-knn", type=int, default=0,
                    help="--workload config5: SurfaceNormalDataPointsFilter knn over the grown map inside every epoch (a point-to-plane replica, Map.cpp:524); 0: off")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher check without GPUs: start the ranks (gloo), count them with an all-reduce, print one JSON line from rank 0")
    args = ap.parse_args()

    # ---- `python bench.py --gpus N` on its own starts the N ranks (one process per GPU) ------------------------------------
    # Under a launcher (torch.distributed.run sets RANK / WORLD_SIZE) this process IS one of the ranks.  Started bare with
    # --gpus N > 1 it re-executes itself under torch.distributed.run on 127.0.0.1 with a free port, so that the command the
    # driver uses for N = 1 also produces the N > 1 line (the flag used to be parsed and ignored).
    if args.gpus > 1 and "RANK" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; reporting the launcher's count",
              file=sys.stderr)
    if args.dry_launch:
        return dry_launch(args)

    # stdout carries ONE JSON line: libraries that write banners to fd 1 (RCCL prints its version block at the first communicator
    # init) are sent to stderr for the length of the run, the line goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # a launcher (torch.distributed.run) sets RANK / WORLD_SIZE: use the process group then, also for a single rank
    use_pg = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = local_rank if use_pg else 0
    torch.cuda.set_device(dev)

    import norlab_icp_mapper_amd as pkg

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- --workload config5: scan streams against the shared 10 M-point map, one merge epoch per scan in the timed region ----
    if args.workload == "config5":
        m5 = args.map_points if args.map_points != M_MAP else 10_000_000
        scale5 = args.scale if args.scale != 1.0 else 3.16
        chain = dict(CHAINS[args.chain])
        sc5 = pkg.synth.make_scene(m=m5, n=8, scale=scale5)
        d_map = torch.from_numpy(sc5["map"]).cuda()
        d_nrm = torch.from_numpy(sc5["normals"]).cuda() if chain["minimizer"] == 2 else None
        d_scans = [torch.from_numpy(x).cuda() for x in circle_scans(pkg, args.scans, args.scan_points, scale5, rank)]
        comm = None
        if use_pg:
            box = [pkg.ICPSequence.commUniqueId() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = ("rccl", box[0], world, rank)
        res = config5_stream(np, torch, pkg, dev, d_map, d_nrm, d_scans, chain, 0.15, comm, barrier, normals_knn=args.epoch_normals_knn)
        el = torch.tensor([res["elapsed_s"], float(res["iterations"])], dtype=torch.float64, device="cuda")
        each = [torch.zeros_like(el) for _ in range(world)]
        if use_pg:
            dist.all_gather(each, el)
        else:
            each = [el]
        elapsed = max(float(e[0].item()) for e in each)
        its_all = sum(float(e[1].item()) for e in each)
        out = {"metric": f"ICP iterations/sec, {world} stream(s) of {args.scan_points}-pt scans vs shared {m5}-pt map, one map-growth epoch per scan (BASELINE config 5)",
               "value": its_all / elapsed, "unit": "iterations/s", "n_gpus": world,
               "steps": args.scans, "warmup": 1, "ms_per_step": elapsed / args.scans * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"BASELINE config 5: {world} scan stream(s) x {args.scans} synthetic {args.scan_points}-pt scans vs the shared "
                                      f"{m5}-pt map (scene x{scale5}), {args.chain} chain, Counter 40 + Differential, one map-growth epoch per scan "
                                      f"(PointDistance 0.15 m accept + RCCL all-gather + rank-ordered merge + append + incremental index insert + 20 m cell binning on the device"
                                      f"{' + SurfaceNormal knn ' + str(args.epoch_normals_knn) + ' over the grown map' if args.epoch_normals_knn else ''}) inside the timed region",
                          "chain": args.chain, "parallelism": f"scan-sharded x{world}, map replicated, RCCL all-gather of accepted points per scan"},
               "scans_per_s": world * args.scans / elapsed, "per_rank_elapsed_s": [float(e[0].item()) for e in each],
               "per_rank_iterations": [float(e[1].item()) for e in each],
               "merge_epoch": {"ms": res["merge_epoch_ms"]["median"], "min_ms": res["merge_epoch_ms"]["min"], "max_ms": res["merge_epoch_ms"]["max"],
                               "n": res["merge_epoch_ms"]["n"], "inside_timed_region": True},
               "rccl_ranks": res.get("rccl_ranks"), "communicator": res.get("communicator"), "rank0": res}
        if rank == 0:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if use_pg:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- workload: replicated map, one scan stream per rank ----
    sc = pkg.synth.make_scene(m=args.map_points, n=args.scan_points, seed_scan=43 + 1000 * rank, scale=args.scale)
    chain = dict(CHAINS[args.chain])
    icp = pkg.ICPSequence(device=dev, max_iterations=ITERS_PER_STEP, **chain)
    normals_ms = None
    if args.normals == "filter":
        t0 = time.perf_counter()
        sc["normals"] = icp.surfaceNormals(sc["map"], knn=10)
        normals_ms = (time.perf_counter() - t0) * 1e3
    d_map = torch.from_numpy(sc["map"]).cuda()
    d_nrm = torch.from_numpy(sc["normals"]).cuda()
    d_scan = torch.from_numpy(sc["scan"]).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert icp.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr())
    set_map_ms = (time.perf_counter() - t0) * 1e3          # first call on this handle: includes every allocation
    warm = []
    for _ in range(5):                                      # what Map::updateLocalPointCloud pays from the second update on
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert icp.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr())
        warm.append((time.perf_counter() - t0) * 1e3)
    set_map_warm_ms = statistics.median(warm)

    batch_scans = None
    if args.batch > 1:
        batch_scans = [d_scan] + [torch.from_numpy(pkg.synth.make_scene(m=8, n=args.scan_points, seed_scan=43 + 1000 * rank + 7 * b, scale=args.scale)["scan"]).cuda()
                                  for b in range(1, args.batch)]

    def step():
        if batch_scans is not None:
            Ts, _, status = icp.registerBatchDev([d.data_ptr() for d in batch_scans], [d.shape[0] for d in batch_scans], fixed_iterations=ITERS_PER_STEP)
            assert not any(status), status
            return Ts[0]
        return icp.registerDev(d_scan.data_ptr(), d_scan.shape[0], fixed_iterations=ITERS_PER_STEP)

    for _ in range(args.warmup):
        step()

    # The contract's timed region: EXACTLY --steps steps between two barriers.  20 steps are 17 ms here -- too short for the driver's own
    # clock and GPU-busy sampler to see -- so the block is REPEATED until a second of wall time has gone by (every
    # block bracketed the same way), and the line reports the MEDIAN block; `blocks` says how many, `block_ms` their spread.
    def timed_block():
        barrier()
        t0 = time.perf_counter()
        loop_ms, per_step, nn_ms, nn_cnt = 0.0, [], 0.0, 0
        for _ in range(args.steps):
            ts = time.perf_counter()
            T = step()
            per_step.append((time.perf_counter() - ts) * 1e3)
            st = icp.stats if batch_scans is None else icp.batch_stats[0]
            loop_ms += st.loop_ms
            nn_ms += st.nn_ms_avg * st.nn_launches; nn_cnt += st.nn_launches   # device clocks of the NN launches of THIS replay (icpmi.h)
        barrier()
        return dict(elapsed=time.perf_counter() - t0, loop_ms=loop_ms, per_step=per_step, T=T, nn_ms=nn_ms, nn_cnt=nn_cnt)

    blocks, t_all = [], time.perf_counter()
    while True:
        blocks.append(timed_block())
        more = (time.perf_counter() - t_all) < args.min_seconds and len(blocks) < 500
        if use_pg:  # every rank takes the same decision (rank 0's clock)
            flag = torch.tensor([1 if more else 0], dtype=torch.int32, device="cuda")
            dist.broadcast(flag, src=0)
            more = bool(flag.item())
        if not more:
            break
    order = sorted(range(len(blocks)), key=lambda i: blocks[i]["elapsed"])
    med = blocks[order[len(order) // 2]]
    elapsed, loop_ms, per_step, T = med["elapsed"], med["loop_ms"], med["per_step"], med["T"]
    timed_nn_us = (sum(b["nn_ms"] for b in blocks) / max(sum(b["nn_cnt"] for b in blocks), 1)) * 1e3
    timed_nn_cnt = sum(b["nn_cnt"] for b in blocks)
    per_rank_value = [args.steps * ITERS_PER_STEP * max(args.batch, 1) / elapsed]
    if use_pg:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        each = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(each, tt)
        per_rank_value = [args.steps * ITERS_PER_STEP * max(args.batch, 1) / float(e.item()) for e in each]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    readings = max(args.batch, 1)
    iters_total = world * args.steps * ITERS_PER_STEP * readings
    value = iters_total / elapsed

    # ---- the one exchange of the multi-GPU path, outside the timed region: a map-growth epoch through the library's own RCCL
    # communicator (icpmi_staged_merge_allgather: accepted points compacted, all-gathered on the handle's stream, merged in rank
    # order and appended on the device).  Every rank must take the same decisions, or the collective would hang: agree first.
    merge = None
    merge_hung = False
    if use_pg:
        # On a watchdog: a collective that never completes (a rank lost, an RCCL transport problem between two GPUs) must not
        # take the headline measured above with it -- after MERGE_TIMEOUT_S the line is printed without the epoch and every
        # rank leaves through os._exit.
        import threading
        box_out = {}

        def merge_epoch():
            torch.cuda.set_device(local_rank)
            m = None
            try:
                ok = 1
                try:
                    box = [icp.commUniqueId() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    icp.commInit(box[0], world, rank)
                except Exception as e:  # noqa: BLE001
                    ok, m = 0, {"error": repr(e)}
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 1:
                    corr = icp.registerWithPrior(sc["scan"], np.eye(4, dtype=np.float32))
                    barrier()
                    tm = time.perf_counter()
                    mine_n, appended, new_m = icp.stagedMergeAllGather(corr, 0.15, normals_knn=0)
                    barrier()
                    tms = torch.tensor([time.perf_counter() - tm], dtype=torch.float64, device="cuda")
                    dist.all_reduce(tms, op=dist.ReduceOp.MAX)
                    cr, cme, ckind = icp.commInfo()
                    m = {"ms": float(tms.item()) * 1e3, "rccl_ranks": cr, "rccl_rank": cme, "communicator": {0: "none", 1: "rccl", 2: "loopback"}[ckind],
                         "accepted_rank0": mine_n, "appended_all_ranks": appended, "map_points_after": new_m,
                         "what": "PointDistance(0.15 m) accept of one 100k-pt scan per rank + RCCL all-gather + rank-ordered exact merge + append + incremental index insert"}
                elif m is None:
                    m = {"error": "another rank could not create its communicator"}
            except Exception as e:  # noqa: BLE001
                m = {"error": repr(e)}
            box_out["merge"] = m

        MERGE_TIMEOUT_S = 90.0
        th = threading.Thread(target=merge_epoch, daemon=True)
        th.start()
        th.join(MERGE_TIMEOUT_S)
        if th.is_alive():
            merge_hung = True
            merge = {"error": f"no result within {MERGE_TIMEOUT_S:.0f} s (collective did not complete); headline unaffected"}
        else:
            merge = box_out.get("merge")

    out = {
        "metric": "ICP iterations/sec, 100k-pt scan vs 1M-pt map",
        "value": value,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"{readings} synthetic {args.scan_points}-pt scans (one batched launch sequence)" if readings > 1 else
                         f"single synthetic {args.scan_points}-pt scan") + f" vs {args.map_points}-pt map, "
                        f"{'point-to-point' if args.chain == 'p2p' else 'point-to-plane'} ICP, KDTreeMatcher knn {chain.get('knn', 1)} maxDist 2.0 "
                        f"epsilon 0, TrimmedDist 0.85, fixed {ITERS_PER_STEP} iterations per registration",
            "chain": args.chain,
            "iterations_per_step": ITERS_PER_STEP * readings,
            "scene": "box+pillars sigma=0.01 seeds 42/43/44 (SURVEY.md 8d)",
            "parallelism": f"scan-sharded x{world}, map replicated",
        },
    }

    if rank == 0:
        gi = icp.gridInfo()
        out["per_rank_value"] = per_rank_value
        if merge is not None:
            out["merge_epoch"] = merge
        out["step_ms"] = step_stats(per_step)
        out["blocks"] = len(blocks)
        out["block_ms"] = step_stats([b["elapsed"] * 1e3 for b in blocks])
        out["timed_wall_s"] = sum(b["elapsed"] for b in blocks)
        out["build"] = pkg._capi.load().icpmi_build_info().decode()
        out["device_loop_ms_per_step"] = loop_ms / args.steps
        out["set_map_ms"] = set_map_ms
        out["set_map_warm_ms"] = set_map_warm_ms
        if normals_ms is not None:
            out["surface_normals_ms"] = normals_ms
            out["config"]["map_normals"] = "SurfaceNormalDataPointsFilter knn 10 (icpmi_surface_normals, host pointers)"
        out["grid"] = gi
        gt_t, gt_r = pkg.synth.pose_error(T, sc["T_gt"])
        out["pose_err_vs_ground_truth"] = {"m": gt_t, "rad": gt_r,
                                           "note": "after the fixed 20 iterations of throughput mode (Counter checker only); "
                                                   "the point-to-point chain needs more iterations to converge, the CPU oracle "
                                                   "lands on the same pose (pose_err_vs_cpu)"}
        traffic_key = {"p2p": "hbm_bytes_per_launch", "p2plane": "hbm_bytes_per_launch_p2plane", "docs_knn6": "hbm_bytes_per_launch_knn6"}[args.chain]
        if args.map_points != M_MAP or args.scan_points != N_SCAN:
            traffic_key = None
        out["roofline"] = nn_roofline(pkg, dev, chain, d_map, d_nrm, d_scan, args.scan_points, args.map_points, traffic_key,
                                      m_reach=map_points_in_reach(np, sc["map"], sc["scan"], chain["max_dist"]),
                                      timed_us=timed_nn_us if batch_scans is None else None, timed_cnt=timed_nn_cnt)

        # ---- the other BASELINE configurations, same measurement (N = 1 only) ----
        if not args.no_extras and world == 1 and args.batch <= 1:
            extras = {}
            # config 3: point-to-plane against SurfaceNormalDataPointsFilter normals
            c3 = dict(CHAINS["p2plane"])
            icp3 = pkg.ICPSequence(device=dev, max_iterations=ITERS_PER_STEP, **c3)
            t0 = time.perf_counter()
            nrm3 = icp3.surfaceNormals(sc["map"], knn=10)
            nrm_ms = (time.perf_counter() - t0) * 1e3
            d_nrm3 = torch.from_numpy(nrm3).cuda()
            icp3.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm3.data_ptr())
            T3, el3, per3 = time_registrations(torch, icp3, d_scan, args.steps, args.warmup)
            g3t, g3r = pkg.synth.pose_error(T3, sc["T_gt"])
            extras["p2plane_filter_normals"] = {
                "config": "BASELINE config 3: 100k-pt scan vs 1M-pt map, point-to-plane, map normals from SurfaceNormalDataPointsFilter{knn 10}",
                "value": args.steps * ITERS_PER_STEP / el3, "unit": "iterations/s", "step_ms": step_stats(per3), "surface_normals_ms": nrm_ms,
                "pose_err_vs_ground_truth": {"m": g3t, "rad": g3r},
                "roofline": nn_roofline(pkg, dev, c3, d_map, d_nrm3, d_scan, args.scan_points, args.map_points, "hbm_bytes_per_launch_p2plane")}
            if not args.no_cpu:
                import oracle_bindings as ob3
                nt3 = min(16, len(os.sched_getaffinity(0)))
                o3 = ob3.OracleICP(ob3.make_config(max_iterations=ITERS_PER_STEP, nthreads=nt3, minimizer=2, max_dist=2.0, outliers=[(4, 0.85)]))
                o3.setMap(sc["map"], nrm3)
                its3, sec3, T3c = 0, 0.0, None
                while sec3 < 3.0 and its3 < 400:
                    _, T3c = o3(sc["scan"]); its3 += o3.stats.iterations; sec3 += o3.stats.seconds_total
                e3t, e3r = pkg.synth.pose_error(T3, T3c)
                extras["p2plane_filter_normals"]["pose_err_vs_cpu"] = {"m": e3t, "rad": e3r}
                extras["p2plane_filter_normals"]["cpu_baseline"] = {"value": its3 / sec3, "unit": "iterations/s", "cores": nt3, "kind": "port",
                                                                    "sample": f"oracle, same map / filter normals / scan: {its3} iterations in {sec3:.1f} s on {nt3} threads"}
                del o3
            del icp3, d_nrm3
            # the documented chain (docs/MapperConfiguration.md:174-189): knn 6, point-to-plane, epsilon 0
            c6 = dict(CHAINS["docs_knn6"])
            icp6 = pkg.ICPSequence(device=dev, max_iterations=ITERS_PER_STEP, **c6)
            icp6.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr())
            T6, el6, per6 = time_registrations(torch, icp6, d_scan, max(args.steps // 2, 5), args.warmup)
            g6t, g6r = pkg.synth.pose_error(T6, sc["T_gt"])
            dbg6 = icp6.debugCounters()  # (r5) iterations of the last registration whose level 0 came from the NN kernel's window / from the full histogram
            extras["docs_knn6"] = {
                "selection_window": {"iterations_served": int(dbg6[12]), "iterations_missed": int(dbg6[13]), "of": ITERS_PER_STEP,
                                     "note": "speculative level 0 of the quantile selection counted by nnk_wg_kernel (iterations 2..); icpmi_config::sel_window_off switches it off"},
                "config": "docs/MapperConfiguration.md:174-189: KDTreeMatcher knn 6 maxDist 2.0 epsilon 0, TrimmedDist 0.85, PointToPlane; 100k-pt scan vs 1M-pt map",
                "value": max(args.steps // 2, 5) * ITERS_PER_STEP / el6, "unit": "iterations/s", "step_ms": step_stats(per6),
                "pose_err_vs_ground_truth": {"m": g6t, "rad": g6r},
                "roofline": nn_roofline(pkg, dev, c6, d_map, d_nrm, d_scan, args.scan_points, args.map_points, "hbm_bytes_per_launch_knn6")}
            if not args.no_cpu:
                import oracle_bindings as ob6
                o6 = ob6.OracleICP(ob6.make_config(max_iterations=ITERS_PER_STEP, nthreads=min(32, len(os.sched_getaffinity(0))), minimizer=2, knn=6, max_dist=2.0,
                                                    outliers=[(4, 0.85)]))
                o6.setMap(sc["map"], sc["normals"])
                _, T6c = o6(sc["scan"])
                e6t, e6r = pkg.synth.pose_error(T6, T6c)
                extras["docs_knn6"]["pose_err_vs_cpu"] = {"m": e6t, "rad": e6r}
                extras["docs_knn6"]["cpu_iterations_per_s"] = o6.stats.iterations / o6.stats.seconds_total
            del icp6
            # the same chain with the matcher's `epsilon: 1` of the SHIPPED configuration (examples/config.yaml:56-60) honoured as libnabo does
            # (icpmi_config::epsilon_approx): a (1 + epsilon)-approximate search; which valid answer comes back differs from libnabo's
            try:
                icp6e = pkg.ICPSequence(device=dev, max_iterations=ITERS_PER_STEP, epsilon=1.0, epsilon_approx=1, **c6)
                icp6e.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr())
                T6e, el6e, per6e = time_registrations(torch, icp6e, d_scan, max(args.steps // 2, 5), args.warmup)
                g6et, g6er = pkg.synth.pose_error(T6e, sc["T_gt"])
                x6t, x6r = pkg.synth.pose_error(T6e, T6)
                extras["docs_knn6_epsilon1"] = {
                    "config": "as docs_knn6 with KDTreeMatcher epsilon 1 served as an approximate search (epsilon_approx = 1): cells farther than (k-th distance) / 2 are not visited",
                    "value": max(args.steps // 2, 5) * ITERS_PER_STEP / el6e, "unit": "iterations/s", "step_ms": step_stats(per6e),
                    "pose_err_vs_ground_truth": {"m": g6et, "rad": g6er}, "pose_diff_vs_exact_search": {"m": x6t, "rad": x6r}}
                del icp6e
            except Exception as e:  # noqa: BLE001
                extras["docs_knn6_epsilon1"] = {"error": repr(e)}
            # what Mapper::processInput runs (Mapper.cpp:213): Counter 40 + Differential -- a registration of data-dependent length, segment graphs
            icpc = pkg.ICPSequence(device=dev, minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
            icpc.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr())
            for _ in range(5):
                Tc = icpc.registerDev(d_scan.data_ptr(), d_scan.shape[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter(); repsc, itsc, perc = 50, 0, []
            for _ in range(repsc):
                ts = time.perf_counter()
                Tc = icpc.registerDev(d_scan.data_ptr(), d_scan.shape[0]); itsc += icpc.stats.iterations
                perc.append((time.perf_counter() - ts) * 1e3)
            torch.cuda.synchronize()
            elc = time.perf_counter() - t0
            gct, gcr = pkg.synth.pose_error(Tc, sc["T_gt"])
            extras["checked_p2plane"] = {
                "config": "production shape (Mapper.cpp:213): 100k-pt scan vs 1M-pt map, point-to-plane, TrimmedDist 0.85, Counter 40 + Differential(1e-3, 1e-3, 3)",
                "ms_per_registration": elc / repsc * 1e3, "iterations_per_registration": itsc / repsc, "us_per_iteration": elc / max(itsc, 1) * 1e6,
                "value": itsc / elc, "unit": "iterations/s", "step_ms": step_stats(perc), "stop_reason": int(icpc.stats.stop_reason),
                "pose_err_vs_ground_truth": {"m": gct, "rad": gcr}}
            del icpc
            # PM::ICPSequence::setDefault() -- the chain of a configuration without an `icp:` key (Mapper.cpp:74-78): SamplingSurfaceNormal
            # on the reference at every setMap (device: icpmi_sampling_surface_normal), RandomSampling(0.75) on the reading, KDTree knn 1,
            # TrimmedDist 0.85, PointToPlane, Counter 40 + Differential
            try:
                icpd = pkg.ICPSequence(device=dev, minimizer=2, max_dist=float("inf"), outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
                icpd.samplingSurfaceNormal(sc["map"])                                  # warm-up (allocations)
                t0 = time.perf_counter()
                order, nrm_ssn = icpd.samplingSurfaceNormal(sc["map"])
                ssn_ms = (time.perf_counter() - t0) * 1e3
                ref = np.ascontiguousarray(sc["map"][order])
                t0 = time.perf_counter()
                icpd.setMap(ref, nrm_ssn)
                sm_ms = (time.perf_counter() - t0) * 1e3
                keep = minstd_random_sampling_keep(np, sc["scan"].shape[0], 0.75, seed=1)   # the seeded minstd stream of the host filter
                d_read = torch.from_numpy(np.ascontiguousarray(sc["scan"][keep])).cuda()
                for _ in range(3):
                    Td = icpd.registerDev(d_read.data_ptr(), d_read.shape[0])
                torch.cuda.synchronize()
                t0 = time.perf_counter(); reps = 20; its = 0
                for _ in range(reps):
                    Td = icpd.registerDev(d_read.data_ptr(), d_read.shape[0]); its += icpd.stats.iterations
                torch.cuda.synchronize()
                eld = time.perf_counter() - t0
                gdt, gdr = pkg.synth.pose_error(Td, sc["T_gt"])
                extras["default_chain"] = {
                    "config": "PM::ICPSequence::setDefault(): SamplingSurfaceNormal{ratio 0.5, knn 7} on the 1M-pt reference (host pointers in, kept indices + "
                              "normals out), RandomSampling 0.75 on the 100k-pt reading, KDTree knn 1, TrimmedDist 0.85, PointToPlane, Counter 40 + Differential",
                    "sampling_surface_normal_ms": ssn_ms, "reference_points_kept": int(order.shape[0]), "set_map_ms": sm_ms,
                    "ms_per_registration": eld / reps * 1e3, "iterations_per_registration": its / reps, "value": its / eld, "unit": "iterations/s",
                    "pose_err_vs_ground_truth": {"m": gdt, "rad": gdr}}
                del icpd, d_read
            except Exception as e:  # noqa: BLE001
                extras["default_chain"] = {"error": repr(e)}
            # batch of 8 readings: one GPU serving 8 scan streams
            B = 8
            scans8 = [d_scan] + [torch.from_numpy(pkg.synth.make_scene(m=8, n=args.scan_points, seed_scan=43 + 7 * b, scale=args.scale)["scan"]).cuda()
                                 for b in range(1, B)]
            ptrs, ns = [d.data_ptr() for d in scans8], [d.shape[0] for d in scans8]
            for _ in range(args.warmup):
                icp.registerBatchDev(ptrs, ns, fixed_iterations=ITERS_PER_STEP)
            torch.cuda.synchronize()
            per8 = []
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ts = time.perf_counter()
                Ts8, _, status8 = icp.registerBatchDev(ptrs, ns, fixed_iterations=ITERS_PER_STEP)
                per8.append((time.perf_counter() - ts) * 1e3)
            torch.cuda.synchronize()
            el8 = time.perf_counter() - t0
            extras["batch8"] = {
                "config": f"8 x {args.scan_points}-pt scans vs the {args.map_points}-pt map in one launch sequence (icpmi_register_batch_dev), {args.chain} chain",
                "value": args.steps * ITERS_PER_STEP * B / el8, "unit": "iterations/s (aggregate of 8 readings)", "step_ms": step_stats(per8),
                "first_reading_equals_single_registration": bool(np.array_equal(Ts8[0], T)), "errors": int(sum(1 for s in status8 if s))}
            del scans8
            # config 5's map on one GPU: 10 M points at the same density
            try:
                sc10 = pkg.synth.make_scene(m=10_000_000, n=args.scan_points, scale=3.16)
                d_map10 = torch.from_numpy(sc10["map"]).cuda(); d_nrm10 = torch.from_numpy(sc10["normals"]).cuda()
                d_scan10 = torch.from_numpy(sc10["scan"]).cuda()
                icp10 = pkg.ICPSequence(device=dev, max_iterations=ITERS_PER_STEP, **chain)
                t0 = time.perf_counter()
                icp10.setMapDev(d_map10.data_ptr(), d_map10.shape[0], d_nrm10.data_ptr())
                sm10 = (time.perf_counter() - t0) * 1e3
                T10, el10, per10 = time_registrations(torch, icp10, d_scan10, args.steps, args.warmup)
                g10t, g10r = pkg.synth.pose_error(T10, sc10["T_gt"])
                del icp10
                extras["map_10M"] = {
                    "config": f"BASELINE config 5's map on one GPU: 100k-pt scan vs 10M-pt map (scene x3.16, same density), {args.chain} chain",
                    "value": args.steps * ITERS_PER_STEP / el10, "unit": "iterations/s", "step_ms": step_stats(per10), "set_map_ms": sm10,
                    "pose_err_vs_ground_truth": {"m": g10t, "rad": g10r},
                    "roofline": nn_roofline(pkg, dev, chain, d_map10, d_nrm10, d_scan10, args.scan_points, 10_000_000, "hbm_bytes_per_launch_10M",
                                            m_reach=map_points_in_reach(np, sc10["map"], sc10["scan"], chain["max_dist"]))}
                # BASELINE config 5 as far as one GPU goes: a stream of scans against the 10 M-point map, one map-growth epoch per scan
                # (single rank: the epoch is accept + compaction + append + index insert, no exchange)
                try:
                    c5scans = [torch.from_numpy(x).cuda() for x in circle_scans(pkg, 6, args.scan_points, 3.16, 0)]
                    r5 = config5_stream(np, torch, pkg, dev, d_map10, None if chain["minimizer"] != 2 else d_nrm10, c5scans, chain, 0.15, None, barrier)
                    extras["config5_stream_1gpu"] = {
                        "config": "BASELINE config 5 on one GPU: 6 x 100k-pt scans (sensors on a 20 m circle, seeds 100 + s) vs the 10M-pt map, Counter 40 + "
                                  "Differential, one PointDistance(0.15 m) map-growth epoch per scan inside the timed region",
                        "scans_per_s": r5["scans"] / r5["elapsed_s"], "value": r5["iterations"] / r5["elapsed_s"], "unit": "iterations/s", **r5}
                    if not args.no_cpu:
                        # the reference's epoch on the CPU (Map::updateLocalPointCloud with PointDistanceMapperModule, then icp.setMap: a kd-tree of the
                        # map for the module's search and another one for the registration index, rebuilt on every update) -- ONE scan as the sample
                        import oracle_bindings as ob5
                        nt5 = min(16, len(os.sched_getaffinity(0)))
                        o5 = ob5.OracleICP(ob5.make_config(max_iterations=40, use_differential=1, nthreads=nt5, minimizer=chain["minimizer"], max_dist=chain["max_dist"],
                                                           outliers=chain["outliers"]))
                        tb = time.perf_counter(); o5.setMap(sc10["map"], sc10["normals"]); build5 = time.perf_counter() - tb
                        scan5 = c5scans[0].cpu().numpy()
                        ta = time.perf_counter(); _, T5 = o5(scan5); reg5 = time.perf_counter() - ta
                        placed5 = ob5.transform(T5, scan5)
                        ta = time.perf_counter(); keep5 = ob5.point_distance_keep(sc10["map"], placed5, 0.15, nthreads=nt5); pd5 = time.perf_counter() - ta
                        grown = np.concatenate([sc10["map"], placed5[keep5]])
                        ta = time.perf_counter(); o5.setMap(grown, None if chain["minimizer"] != 2 else np.concatenate([sc10["normals"], np.zeros((int(keep5.sum()), 3), np.float32)])); reb5 = time.perf_counter() - ta
                        extras["config5_stream_1gpu"]["cpu_baseline"] = {
                            "value": 1.0 / (reg5 + pd5 + reb5), "unit": "scans/s", "cores": nt5, "kind": "port",
                            "sample": f"oracle, ONE scan of the stream against the 10M-pt map on {nt5} threads: registration {reg5:.2f} s ({o5.stats.iterations} iterations), "
                                      f"PointDistance accept (kd-tree of the map + search) {pd5:.2f} s, index rebuild of the grown map {reb5:.2f} s; first build {build5:.2f} s excluded",
                            "accepted": int(keep5.sum())}
                        extras["config5_stream_1gpu"]["speedup_vs_cpu"] = extras["config5_stream_1gpu"]["scans_per_s"] / extras["config5_stream_1gpu"]["cpu_baseline"]["value"]
                        del o5, grown
                    # a point-to-plane replica: the epoch also recomputes the normals of the grown 10 M-point map (Map.cpp:524 applies the
                    # SurfaceNormal post filter to the whole local map on every update) -- VERDICT r5 weak 11
                    try:
                        pchain = dict(CHAINS["p2plane"])
                        rp = config5_stream(np, torch, pkg, dev, d_map10, d_nrm10, c5scans[:3], pchain, 0.15, None, barrier, normals_knn=10)
                        extras["config5_stream_1gpu_p2plane"] = {
                            "config": "as config5_stream_1gpu with the point-to-plane chain, 3 scans; every epoch ends with SurfaceNormalDataPointsFilter knn 10 over the "
                                      "grown 10M-pt map (sparse block-grid self k-NN + eigen-solve per point) before the index insert",
                            "scans_per_s": rp["scans"] / rp["elapsed_s"], "value": rp["iterations"] / rp["elapsed_s"], "unit": "iterations/s", **rp}
                    except Exception as e:  # noqa: BLE001
                        extras["config5_stream_1gpu_p2plane"] = {"error": repr(e)}
                    del c5scans
                except Exception as e:  # noqa: BLE001
                    extras["config5_stream_1gpu"] = {"error": repr(e)}
                del d_map10, d_nrm10, d_scan10, sc10
            except Exception as e:  # a host without the memory for the 10 M scene: say so instead of failing the headline
                extras["map_10M"] = {"error": repr(e)}
            # the replicated work of the merge as the rank count grows, on ONE GPU through the loopback communicator (csrc/comm.hip: every
            # simulated rank hands in this rank's accepted block, rank r's moved r x 0.4 m along x): R = 1, 2, 4, 8
            try:
                lb = {}
                lscans = [torch.from_numpy(x).cuda() for x in circle_scans(pkg, 5, args.scan_points, args.scale, 0)]
                for R in (1, 2, 4, 8):
                    rr = config5_stream(np, torch, pkg, dev, d_map, None if chain["minimizer"] != 2 else d_nrm, lscans, chain, 0.15,
                                        None if R == 1 else ("loopback", R, 0.4), barrier)
                    lb[f"R{R}"] = {"merge_epoch_ms": rr["merge_epoch_ms"], "appended_per_epoch_all_ranks": rr["appended_per_epoch_all_ranks"],
                                   "accepted_per_scan_this_rank": rr["accepted_per_scan_this_rank"], "scans_per_s": rr["scans"] / rr["elapsed_s"],
                                   "communicator": rr["communicator"], "ranks": rr["rccl_ranks"],
                                   "epochs_one_collective": rr["epochs_one_collective"], "epochs_three_collectives": rr["epochs_three_collectives"]}
                lb["epoch_ms_R8_over_R1"] = lb["R8"]["merge_epoch_ms"]["median"] / lb["R1"]["merge_epoch_ms"]["median"]
                extras["merge_loopback"] = {
                    "config": f"map-growth epoch vs simulated rank count on one GPU (ICPMI_COMM_LOOPBACK): 5 x {args.scan_points}-pt scans vs the "
                              f"{args.map_points}-pt map, PointDistance 0.15 m, blocks of rank r shifted r x 0.4 m", **lb}
                del lscans
            except Exception as e:  # noqa: BLE001
                extras["merge_loopback"] = {"error": repr(e)}
            try:
                extras["config4_replay"] = config4_replay(np, pkg, not args.no_cpu)
            except Exception as e:  # noqa: BLE001
                extras["config4_replay"] = {"error": repr(e)}
            try:
                r4e = config4_replay(np, pkg, False, epsilon_approx=True)
                if "config" in r4e:
                    r4e["config"] = r4e["config"].replace("shipped chain with epsilon 0", "shipped chain with its epsilon 1 served as an approximate search (epsilon_approx)")
                extras["config4_replay_epsilon1"] = r4e
            except Exception as e:  # noqa: BLE001
                extras["config4_replay_epsilon1"] = {"error": repr(e)}
            out["chains"] = extras

        # ---- CPU baseline: the oracle on this host, bounded sample of the same workload ----
        pm_lib = os.path.join(ROOT, "oracle", "_ref", "liboracle_pm.so")
        out["libpointmatcher"] = "present" if os.path.exists(pm_lib) else "absent"
        if not args.no_cpu and world == 1:          # reported on rank 0 at N = 1 only
            import oracle_bindings as ob
            cores = len(os.sched_getaffinity(0))
            okw = dict(minimizer=chain["minimizer"], max_dist=chain["max_dist"], outliers=chain["outliers"], knn=chain.get("knn", 1))
            # the thread count that is fastest on this host (more threads than ~32 run slower here): one
            # registration per candidate, then the long sample with the winner
            best_nt, best_rate, build_s, oicp = 1, 0.0, 0.0, None
            for nt in sorted({min(cores, t) for t in (8, 16, 32, 64)}):
                o = ob.OracleICP(ob.make_config(max_iterations=ITERS_PER_STEP, nthreads=nt, **okw))
                tb = time.perf_counter()
                o.setMap(sc["map"], sc["normals"])
                bs = time.perf_counter() - tb
                o(sc["scan"])
                rate = o.stats.iterations / o.stats.seconds_total
                if rate > best_rate:
                    best_nt, best_rate, build_s, oicp = nt, rate, bs, o
            nthreads = best_nt
            # multi-threaded leg: repeat the 20-iteration registration until ~args.cpu_seconds of work
            mt_iters, mt_secs, mt_regs, mt_knn = 0, 0.0, 0, 0.0
            while mt_secs < args.cpu_seconds and mt_regs < 200:
                err, T_cpu = oicp(sc["scan"])
                mt_iters += oicp.stats.iterations; mt_secs += oicp.stats.seconds_total; mt_knn += oicp.stats.seconds_knn; mt_regs += 1
            mt_its = mt_iters / mt_secs
            # single-threaded leg (libpointmatcher's own loop is single threaded outside libnabo): a few iterations
            o1 = ob.OracleICP(ob.make_config(max_iterations=max(2, args.cpu_iters), nthreads=1, **okw))
            o1.setMap(sc["map"], sc["normals"])
            o1(sc["scan"])
            st_its = o1.stats.iterations / o1.stats.seconds_total
            dt, dr = pkg.synth.pose_error(T, T_cpu)
            out["pose_err_vs_cpu"] = {"m": dt, "rad": dr, "tolerance": {"m": 1e-4, "rad": 1e-4}}
            out["cpu_baseline"] = {
                "value": max(mt_its, st_its), "unit": "iterations/s", "cores": nthreads if mt_its >= st_its else 1, "kind": "port",
                "sample": f"oracle (C restatement, gcc -O3, OpenMP over the kNN queries), same map and scan: {mt_regs} x "
                          f"{ITERS_PER_STEP}-iteration registrations on {nthreads} threads ({mt_secs:.1f} s, {mt_its:.2f} it/s) and "
                          f"{o1.stats.iterations} iterations on 1 thread ({st_its:.2f} it/s); kd-tree build {build_s:.2f} s excluded",
                "value_1thread": st_its, "value_multithread": mt_its, "host_cores": cores,
                "knn_fraction_multithread": mt_knn / max(mt_secs, 1e-9), "knn_fraction_1thread": o1.stats.seconds_knn / max(o1.stats.seconds_total, 1e-9),
                "note": "the port parallelises the kNN loop only (as libnabo does under OpenMP); the gather of the pairs, the quantile selection "
                        "(nth_element) and the pair sums run on one thread, as in libpointmatcher -- with knn_fraction_multithread of the time in "
                        "the kNN at the best thread count the rest is Amdahl's serial part, which is why more threads than that do not help. "
                        "It is a port timed for scale, not libpointmatcher; speedup_vs_cpu is quoted against it and is not a claim about the reference's binary.",
                "cpu_model": cpu_model(), "compiler_flags": "gcc -O3 -march=x86-64-v3 -mfma -ffp-contract=off -fopenmp (oracle/Makefile)",
            }
            out["speedup_vs_cpu"] = value / out["cpu_baseline"]["value"]
            if out["libpointmatcher"] == "present":
                try:
                    import oracle_pm_bindings as opm
                    T_pm = opm.register_default_chain(args.chain, sc["map"], sc["normals"], sc["scan"], ITERS_PER_STEP)
                    pt, pr = pkg.synth.pose_error(T, T_pm)
                    out["pose_err_vs_libpointmatcher"] = {"m": pt, "rad": pr}
                except Exception as e:
                    out["pose_err_vs_libpointmatcher"] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_pg:
        if merge_hung:
            sys.stdout.flush()
            os._exit(0)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
