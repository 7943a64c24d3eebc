"""`python bench.py --gpus N` must work however it is launched: without a launcher the
parent starts the N ranks itself through torch.distributed.run before anything touches a GPU.  CPU tier: the
rehearsal mode RZK_BENCH_JOIN_ONLY=1 stops after both ranks have joined the (gloo) process group and reduced one
value, which is everything of the launch path that does not need a device."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(RZK_BENCH_JOIN_ONLY="1", RZK_BENCH_BACKEND="gloo")
    return env


def test_self_launch_two_ranks_join():
    """The driver's 1-GPU command-line form with --gpus 2: one JSON line from rank 0, both ranks joined."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j == {"joined": 2, "max_elapsed": 2.0, "sum": 3, "per_rank": [1, 2], "self_launched": True}


def test_under_torchrun_no_second_launch():
    """Under torch.distributed.run (the driver's N>1 form) bench.py must not start ranks of its own."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["joined"] == 2 and j["self_launched"] is False


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
