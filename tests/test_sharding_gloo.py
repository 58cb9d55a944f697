"""N>1 path on CPU: world_size-2 gloo processes shard a batch by instance, each rank produces its shard's results
(here with the CPU oracle as the stand-in compute; the GPU engine plugs in at the same place) and the shards are
gathered. Covers ragged shards."""
import importlib
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import oracle_lib, synth

    sharding = importlib.import_module("bls-verify-gadget_amd.sharding")
    o = oracle_lib.load()
    pk, msg, sig, expect = synth.make_batch(o, n_total, tamper_every=2)
    lo, hi = sharding.shard_range(n_total, rank, world)
    res, _ = o.witness_batch(pk[lo:hi], msg[lo:hi], sig[lo:hi], threads=1, want_digests=False)
    local = torch.from_numpy(res.astype(np.int32))
    full = sharding.all_gather_results(local, n_total)
    chunk = torch.full((2, 3, 6), rank, dtype=torch.int64)
    gathered = sharding.all_gather_witness_chunk(chunk)
    # the micro-batched gather of bench.py's all-gather leg (same function, gloo instead of RCCL): 5 rows per rank in
    # micro-batches of 2 (ragged tail), every micro-batch consumed before the next; the consumer sees rank-major rows
    local = (torch.arange(5 * 4 * 6, dtype=torch.int64).reshape(5, 4, 6) + 1000 * rank)
    seen = []
    buf = torch.empty((world * 2, 4, 6), dtype=torch.int64)
    nmb = sharding.stream_allgather(local, 2, lambda g, c0, rows: seen.append((c0, rows, g[:, 0, 0].tolist())), buffer=buf)
    # the compact-form gather (bench.py's default all-gather leg): one uint8 batch per rank, expanded rank by rank on the receiver
    # (stand-in expansion on the CPU: bytes -> int64; the GPU engine's expand_compact plugs in at the same place)
    comp = torch.arange(40, dtype=torch.uint8) + 100 * rank
    cseen = []
    nb = sharding.stream_allgather_compact(comp, lambda c: c.to(torch.int64) * 2, lambda w, r: cseen.append((r, int(w[0]), int(w[-1]), w.numel())))
    # the double-buffered order of the all-gather leg (CompactGatherPipeline): the gather of batch k + 1 is issued before batch k
    # is expanded and consumed; three batches through two gathered buffers
    pseen = []
    pipe = sharding.CompactGatherPipeline(world, 16, torch.device("cpu"), lambda c: c.to(torch.int64) + 1, lambda w, r, k: pseen.append((k, r, int(w[0]))))
    hooks = []
    for k in range(3):
        pipe.push(torch.full((16,), 10 * k + rank, dtype=torch.uint8), before=lambda st, k=k: hooks.append(("before", k)), after=lambda st, k=k: hooks.append(("after", k)))
    pipe.flush()
    if rank == 0:
        q.put((full.numpy().astype(bool).tolist(), expect.tolist(), gathered[:, 0, 0].tolist(), nmb, seen, nb, cseen, pseen, pipe.order, hooks))
    dist.destroy_process_group()


def test_shard_ranges():
    sharding = importlib.import_module("bls-verify-gadget_amd.sharding")
    for n in (0, 1, 5, 8, 65536):
        for w in (1, 2, 3, 8):
            r = [sharding.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_total = 5  # ragged: 3 + 2
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, expect, gathered, nmb, seen, nb, cseen, pseen, porder, hooks = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == expect
    assert gathered == [0, 0, 1, 1]
    assert nb == 2 and cseen == [(0, 0, 78, 40), (1, 200, 278, 40)]
    assert nmb == 3 and [(c0, rows) for c0, rows, _ in seen] == [(0, 2), (2, 2), (4, 1)]
    assert seen[0][2] == [0, 24, 1000, 1024] and seen[2][2] == [96, 1096]
    assert pseen == [(0, 0, 1), (0, 1, 2), (1, 0, 11), (1, 1, 12), (2, 0, 21), (2, 1, 22)]  # (batch, rank, first element + 1)
    assert porder == [("gather", 0), ("gather", 1), ("consume", 0), ("gather", 2), ("consume", 1), ("consume", 2)]
    assert hooks == [("before", 0), ("after", 0), ("before", 1), ("after", 1), ("before", 2), ("after", 2)]
