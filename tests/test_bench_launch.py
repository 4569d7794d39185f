"""bench.py --gpus N starts its own N ranks (VERDICT r04 missing #8): the command line it re-executes under, the refusal
to print a one-rank line for an N > 1 request, and (on the GPU box) a real two-rank run with no outer launcher."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_command_is_one_rank_per_gpu_on_loopback():
    import bench
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "1"], port=29511, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"]     # own arguments pass through unchanged
    # no port given: a free loopback port is picked
    port = int(bench.launcher_command(2, [])[bench.launcher_command(2, []).index("--master-port") + 1])
    assert 1024 < port < 65536


def test_more_ranks_than_gpus_is_refused_not_downgraded():
    """No GPU here: `--gpus 2` must exit non-zero and print no JSON line (never an n_gpus = 1 line for an N = 2 request)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU node: the request is satisfiable")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing to run fewer ranks" in r.stderr
    assert r.stdout.strip() == ""


def test_world_size_must_equal_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_two_ranks_start_themselves_on_a_shared_gpu():
    """`python bench.py --gpus 2` with NO launcher around it: two ranks come up (gloo, both on cuda:0), run the row-range
    sharded Q1 with the all-to-all exchange of partial states, and rank 0 prints one line with n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2",
                        "--warmup", "1", "--rows", "8000000", "--no-q3", "--no-ann", "--no-cpu"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["distinct_devices_seen"] == 1
    assert out["rccl_ranks_seen"] == 0 and out["backend"] == "gloo"         # gloo ranks are not RCCL ranks
    assert out["config"]["groups"] == 4 and out["config"]["rows_total"] == 8000000
