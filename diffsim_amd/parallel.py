"""Multi-GPU sharding of DiffSim scoring: one process per GPU, pairs are the shard unit.

The reference is a single-process, single-GPU, batch-of-one loop (cute_main.py:54-132,
``export CUDA_VISIBLE_DEVICES=N`` in the *.sh drivers).  Pairs are independent and weights are
read-only, so the build shards the pair list across ranks with NO data-path collective; the only
communication is one all_gather of the fp32 scores at the end (RCCL over xGMI with the "nccl"
backend on GPUs; gloo on CPU in the tests).  4 bytes per pair: latency-bound, not bandwidth-bound.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Sequence

import torch
import torch.distributed as dist


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int, cmd: Sequence[str], poll_s: float = 0.2, grace_s: float = 10.0) -> int:
    """Start `n` fresh rank processes of `cmd` on this node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, rendezvous on 127.0.0.1) and supervise them; returns the first non-zero exit code.

    All ranks are polled together: when one exits non-zero (a bad image path in its shard, out of memory, a missing
    weight), the others -- which would otherwise sit in their next collective until the RCCL / gloo timeout -- are
    terminated (SIGTERM, then SIGKILL after `grace_s`) and that rank's code is returned at once.  SIGINT / SIGTERM to
    this process tear the ranks down the same way (no orphan GPU processes behind a Ctrl-C).  The caller must not have
    touched the GPU: nothing is exec'ed from a GPU-initialised process, the ranks are plain children."""
    import signal
    import time
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env0["MASTER_ADDR"] = "127.0.0.1"
    env0["MASTER_PORT"] = str(free_port())
    env0["WORLD_SIZE"] = env0["LOCAL_WORLD_SIZE"] = str(n)
    procs = [subprocess.Popen(list(cmd), env=dict(env0, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + grace_s
        for p in procs:
            try:
                p.wait(timeout=max(0.0, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    got = []

    def on_signal(signum, _frame):
        got.append(signum)

    old = {}
    for sig in (signal.SIGINT, signal.SIGTERM):
        try:
            old[sig] = signal.signal(sig, on_signal)
        except ValueError:          # not the main thread: no handlers, the polling below still works
            pass
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            if got:
                print(f"signal {got[0]}: stopping {len(pending)} rank(s)", file=sys.stderr, flush=True)
                stop_all()
                return 128 + got[0]
            for r in sorted(pending):
                c = procs[r].poll()
                if c is None:
                    continue
                pending.discard(r)
                if c != 0:
                    print(f"rank {r} exited with code {c}; stopping the other ranks", file=sys.stderr, flush=True)
                    stop_all()
                    return c
            if pending:
                time.sleep(poll_s)
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc


def pin_to_gpu_numa(local_rank: int) -> bool:
    """Best effort: restrict this rank's host threads (image decode / resize pool included) to the CPUs of the NUMA node its
    GPU hangs off, so that the eight ranks of a node neither migrate across sockets nor crowd one.  Returns whether it
    pinned; any missing sysfs entry leaves the affinity alone."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return False
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        before = os.sched_getaffinity(0)
        cpus &= before
        if not cpus:
            return False
        os.sched_setaffinity(0, cpus)
        # the ranks of this host now share len(cpus) cores per NUMA node, not len(before) cores among all of them: tell the
        # pool sizing (image.host_threads) how many ranks sit on THIS node's cores -- the local world scaled by the share of
        # the host's cores this node holds (2 sockets x 4 GPUs: 8 ranks -> 4 per node)
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        os.environ["DSIM_RANKS_ON_THESE_CPUS"] = str(max(1, round(lw * len(cpus) / max(1, len(before)))))
        return True
    except Exception:
        return False


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Strided shard: rank r owns items r, r+world, ... (balanced to within one item)."""
    return list(range(rank, n_items, world))


def shard_triplets(n_triplets: int, rank: int, world: int) -> List[int]:
    """Triplets (ref, left, right) are sharded whole so the reference image's cached features are
    reused inside one rank."""
    return shard_indices(n_triplets, rank, world)


def gather_scores(local_scores: torch.Tensor, n_items: int, rank: int, world: int) -> torch.Tensor:
    """All-gather the per-rank score vectors of a strided shard back into item order.

    ``local_scores`` holds the scores of ``shard_indices(n_items, rank, world)`` in that order, on
    the device the process group works with.  Returns a length-``n_items`` fp32 tensor on every rank.
    """
    if world == 1:
        return local_scores.float()
    per = (n_items + world - 1) // world
    pad = torch.full((per,), float("nan"), dtype=torch.float32, device=local_scores.device)
    pad[: local_scores.numel()] = local_scores.float()
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = torch.empty(n_items, dtype=torch.float32, device=local_scores.device)
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        out[idx] = bufs[r][: len(idx)]
    return out


def score_pairs_sharded(scorer, latA: torch.Tensor, latB: torch.Tensor, noiseA, noiseB, prompt, rank: int, world: int,
                        **kw) -> torch.Tensor:
    """Score pairs (latA[i], latB[i]) with rank r taking i = r (mod world); every rank returns all
    scores in pair order.  Per-pair results are bit-identical to the single-GPU run because the
    engine's reductions are fixed-order and batch-invariant."""
    n = latA.shape[0]
    idx = shard_indices(n, rank, world)
    if idx:
        sel = torch.tensor(idx, dtype=torch.long)
        # per-pair noise (n,4,s,s) is sharded with the pairs; a shared (1,4,s,s) draw is passed through
        nA = noiseA[sel] if torch.is_tensor(noiseA) and noiseA.shape[0] == n and n > 1 else noiseA
        nB = noiseB[sel] if torch.is_tensor(noiseB) and noiseB.shape[0] == n and n > 1 else noiseB
        local = scorer.score_latent_pairs(latA[sel], latB[sel], nA, nB, prompt, **kw)
    else:
        local = torch.empty(0, dtype=torch.float32, device=scorer.device)
    return gather_scores(local, n, rank, world)
