"""bench.py's N-rank launch (BASELINE north_star: "reported at 1, 2, 4 and 8 GPUs"): `--gpus N` must really run N
ranks -- started by bench.py itself when no torch.distributed launcher is around it, or by torch.distributed.run as the
driver does -- and refuse to print a line whose n_gpus would not be the number of ranks.  CPU stand-in: the same
rendezvous / barrier / max-over-ranks / score all_gather / rank census on gloo (`--selftest-launch`); no kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return e


def _line(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out           # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-launch"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["ranks_in_gather"] == [0, 1]
    assert d["steps"] == 3 and d["warmup"] == 1
    assert abs(d["max_rank_s"] - 2e-3) < 1e-9          # MAX over ranks of the per-rank time (rank 1 reports 2 ms)


def test_under_torch_distributed_run():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2


def test_mislabelled_runs_are_refused():
    e = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launch"], capture_output=True, text=True, env=e, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and "{" not in r.stdout
    # more GPUs asked for than visible (none here): no silent 1-GPU run
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], capture_output=True, text=True, env=_env(), timeout=120)
    if r.returncode == 0:
        assert _line(r.stdout)["n_gpus"] == 8          # only on a real 8-GPU node
    else:
        assert "visible" in r.stderr and "{" not in r.stdout
