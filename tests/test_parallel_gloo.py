"""The N>1 path on CPU: world_size-2 gloo process group, strided pair sharding + score gather.
(The scoring kernels need a GPU; a deterministic stand-in scorer plays their part here.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffsim_amd import parallel as P


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StandInScorer:
    device = torch.device("cpu")

    def score_latent_pairs(self, latA, latB, nA, nB, prompt, **kw):
        s = (latA.flatten(1).sum(1) * 0.5 - latB.flatten(1).mean(1)).float()
        for nz in (nA, nB):
            if nz is not None:      # shared (1,...) draw or per-pair (n,...) noise, as DiffSim.score_latent_pairs accepts
                assert nz.shape[0] in (1, latA.shape[0])
                s = s + nz.expand(latA.shape[0], *nz.shape[1:]).flatten(1).sum(1) * 0.25
        return s


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        latA = torch.randn(n, 4, 4, 4, generator=g)
        latB = torch.randn(n, 4, 4, 4, generator=g)
        sc = _StandInScorer()
        got = P.score_pairs_sharded(sc, latA, latB, None, None, "p", rank, world)
        want = sc.score_latent_pairs(latA, latB, None, None, "p")
        ok = bool(torch.equal(got, want))
        # per-pair noise must be sharded with the pairs; a shared (1,...) draw is passed through
        nzA, nzB = torch.randn(n, 4, 4, 4, generator=g), torch.randn(1, 4, 4, 4, generator=g)
        got = P.score_pairs_sharded(sc, latA, latB, nzA, nzB, "p", rank, world)
        ok = ok and bool(torch.equal(got, sc.score_latent_pairs(latA, latB, nzA, nzB, "p")))
        q.put((rank, ok, P.shard_indices(n, rank, world)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 8, 1])
def test_sharded_scores_match_single_process(n):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = []
    for rank, ok, idx in res:
        assert ok, f"rank {rank} gathered scores differ from the single-process run"
        seen += idx
    assert sorted(seen) == list(range(n))          # every pair scored exactly once


def test_sharded_scores_match_single_process_world8():
    """The node's full width: 8 ranks, 13 pairs (shards of 2 and 1) -- every pair scored exactly once, gathered in order."""
    world, port, n = 8, _free_port(), 13
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    seen = []
    for rank, ok, idx in res:
        assert ok, f"rank {rank} gathered scores differ from the single-process run"
        seen += idx
    assert sorted(seen) == list(range(n))


def test_shard_balance():
    for n in (0, 1, 9, 10000):
        for w in (1, 2, 4, 8):
            sizes = [len(P.shard_indices(n, r, w)) for r in range(w)]
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
