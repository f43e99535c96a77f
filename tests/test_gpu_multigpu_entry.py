"""First-contact insurance for the N > 1 entry points (no 8-GPU node was ever available to a
round's driver): the process-group branch of bench.py and tools/bench_extract.py started exactly the way a
multi-GPU driver starts them — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P <script> --gpus N ...` — at N = 1, with the RCCL group and the collective forced.  Each runs in a child
process (the launcher starts before anything touches the GPU; this pytest process keeps its own context)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(cmd, env, timeout):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(lines[-1]), r


def _env(**extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra)
    return env


def test_bench_process_group_branch_under_the_launcher():
    """bench.py as the driver launches it for N > 1, at N = 1: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher's
    environment, nccl group initialised (FNP_BENCH_FORCE_DIST), barrier + MAX all-reduce of the elapsed time, rank 0's line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--reps", "1", "--batch", "4",
           "--no-sweep", "--no-secondary", "--cpu-scenes", "0"]
    j, r = _last_json(cmd, _env(FNP_BENCH_FORCE_DIST="1"), 600)
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["scenes_per_step_per_gpu"] == 4 and len(j["config"]["site_counts"]) == 5
    assert abs(j["value"] - 4 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    assert "roofline" in j and j["roofline"]["step"]["min_traffic_bytes"] > 0


def test_bench_refuses_a_gpu_count_that_is_not_the_world_size():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_extraction_entry_under_the_launcher_with_the_collective():
    """tools/bench_extract.py --gpus 1 --force-collective under torch.distributed.run: the sharded extraction loop with
    one RCCL all_gather_into_tensor per step, files written, recall record merged."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join("tools", "bench_extract.py"), "--gpus", "1", "--force-collective", "--scenes", "32"]
    j, r = _last_json(cmd, _env(), 600)
    assert j["n_gpus"] == 1 and j["scenes"] == 32 and j["collective"].startswith("rccl all_gather_into_tensor")
    assert j["pipeline"] is True and j["scenes_per_s"] > 0 and j["gt"] and j["gt"] > 0
