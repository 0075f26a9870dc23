"""`python bench.py --gpus N` must itself start N ranks (the flag was parsed and ignored, so the driver's SCALE step
could not produce an N > 1 line).  --dry-launch runs the launcher path on CPU: gloo instead of RCCL, no HIP work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                     # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    j = _run(["--gpus", "2", "--dry-launch"])
    assert j["dry_launch"] and j["n_gpus"] == 2 and j["ranks_seen"] == 2
    assert sorted(r["rank"] for r in j["ranks"]) == [0, 1]
    assert sorted(r["local_rank"] for r in j["ranks"]) == [0, 1]
    assert len({r["pid"] for r in j["ranks"]}) == 2        # two processes: one per GPU


def test_gpus_1_stays_in_process():
    j = _run(["--gpus", "1", "--dry-launch"])
    assert j["n_gpus"] == 1 and j["ranks_seen"] == 1 and j["ranks"][0]["pid"] > 0


def test_external_launcher_is_respected():
    """the driver's own form: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2
