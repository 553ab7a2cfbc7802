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
    # more GPUs asked for than visible (none here): no silent 1-GPU run.  The parent never touches the HIP runtime; the
    # ranks refuse themselves ("needs a GPU" / "rank r needs GPU k, only n visible") and the supervisor stops the rest
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], capture_output=True, text=True, env=_env(), timeout=300)
    if r.returncode == 0:
        assert _line(r.stdout)["n_gpus"] == 8          # only on a real 8-GPU node
    else:
        assert ("visible" in r.stderr or "needs a GPU" in r.stderr) and "{" not in r.stdout
        assert "stopping the other ranks" in r.stderr


def test_spawn_ranks_stops_the_survivors_when_one_rank_dies(tmp_path):
    """parallel.spawn_ranks polls all ranks: rank 1 exits 7 at once, rank 0 would sleep a minute (a stand-in for a rank
    waiting in its next collective) -- the launcher must return 7 within seconds and leave no child behind."""
    import time
    sys.path.insert(0, ROOT)
    from diffsim_amd import parallel
    script = tmp_path / "rank.py"
    pidfile = tmp_path / "pid0"
    script.write_text(
        "import os, sys, time\n"
        "if os.environ['RANK'] == '1':\n"
        "    sys.exit(7)\n"
        f"open({str(pidfile)!r}, 'w').write(str(os.getpid()))\n"
        "time.sleep(60)\n")
    t0 = time.monotonic()
    rc = parallel.spawn_ranks(2, [sys.executable, str(script)], poll_s=0.05, grace_s=5.0)
    assert rc == 7 and time.monotonic() - t0 < 20
    for _ in range(100):                       # rank 0 may not have written its pid before it was stopped
        if pidfile.exists():
            break
        time.sleep(0.01)
    if pidfile.exists():
        pid = int(pidfile.read_text())
        time.sleep(0.2)
        try:
            os.kill(pid, 0)
            alive = True
        except OSError:
            alive = False
        assert not alive


def test_spawn_ranks_forwards_sigterm(tmp_path):
    """SIGTERM to the launching process tears its ranks down (no orphan GPU processes behind a Ctrl-C / kill)."""
    import signal
    import time
    script = tmp_path / "rank.py"
    script.write_text("import time\ntime.sleep(120)\n")
    launcher = tmp_path / "launch.py"
    launcher.write_text(
        f"import sys\nsys.path.insert(0, {ROOT!r})\nfrom diffsim_amd import parallel\n"
        f"sys.exit(parallel.spawn_ranks(2, [sys.executable, {str(script)!r}], poll_s=0.05, grace_s=5.0))\n")
    p = subprocess.Popen([sys.executable, str(launcher)], env=_env())
    time.sleep(3.0)                            # import torch + start the ranks
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=30)
    assert rc == 128 + signal.SIGTERM


def test_self_launch_eight_ranks():
    """The driver's scaling run goes to 8 ranks: the same plumbing at the full node width (gloo on CPU)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--selftest-launch"],
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 8 and d["n_ranks_seen"] == 8 and d["ranks_in_gather"] == list(range(8))
    assert abs(d["max_rank_s"] - 8e-3) < 1e-9          # MAX over ranks: rank 7 reports 8 ms


def _cute_tree(root, n_cls=2, n_inst=3, n_light=2, n_img=3):
    from PIL import Image
    for c in range(n_cls):
        for i in range(n_inst):
            for l in range(n_light):
                d = os.path.join(root, f"cls{c}", f"inst{i}", f"light{l}")
                os.makedirs(d)
                for k in range(n_img):
                    Image.new("RGB", (8, 8), (c * 40, i * 40, k * 40)).save(os.path.join(d, f"im{k}.png"))


def test_cli_triplet_shard_eight_ranks(tmp_path):
    """python -m diffsim_amd --ngpu 8: the CUTE walk sharded over 8 ranks (60 triplets: shards of 8 and 7), both score gathers and
    the printed counts must equal the 1-rank run (stand-in scorer on CPU, gloo)."""
    _cute_tree(str(tmp_path))
    base = [sys.executable, "-m", "diffsim_amd", "--image_path", str(tmp_path), "--target_layer", "0", "--target_step", "600",
            "--similarity", "cosine", "--seed", "2334", "--selftest_shard"]
    r1 = subprocess.run(base, capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r8 = subprocess.run(base + ["--ngpu", "8"], capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=600)
    assert r8.returncode == 0, r8.stderr[-2000:]
    keep = lambda out: [ln for ln in out.splitlines() if ln.startswith(("Total", "Accuracy", "2x", "Final"))]
    assert keep(r1.stdout) and keep(r1.stdout) == keep(r8.stdout)


def test_cli_rank_dying_between_the_collectives_stops_the_launch(tmp_path):
    """Rank 5 of 8 exits between the two score gathers; the other seven are then waiting in the second all_gather.  The
    supervising launcher must stop them and return the failing code within seconds, not at the collective's timeout."""
    import time
    _cute_tree(str(tmp_path))
    cmd = [sys.executable, "-m", "diffsim_amd", "--image_path", str(tmp_path), "--target_layer", "0", "--target_step", "600",
           "--similarity", "cosine", "--selftest_shard", "--ngpu", "8"]
    t0 = time.monotonic()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(_env(), DSIM_SELFTEST_DIE_RANK="5"), cwd=ROOT, timeout=600)
    assert r.returncode == 9, (r.returncode, r.stderr[-2000:])
    assert time.monotonic() - t0 < 180 and "stopping the other ranks" in r.stderr
