"""Worker of tests/test_multi_gpu_rccl.py (one process per GPU under torch.distributed.run, backend nccl = RCCL): every rank runs a
consumer-mode engine over its own shard, ships each step in compact wire form through the double-buffered all-gather
(sharding.CompactGatherPipeline), expands EVERY rank's batch and digests it; the digests must equal those of a direct-mode engine
run locally on every rank's inputs (the inputs are deterministic, so each rank can mint all of them). Rank 0 prints one JSON line."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("BLSW_TEST_BACKEND", "nccl")  # "gloo": a rehearsal of world > 1 on ONE GPU (every rank on device 0; RCCL refuses that)
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    saved = os.dup(1)  # RCCL prints its banner on stdout
    os.dup2(2, 1)
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        dist.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    pkg = importlib.import_module("bls-verify-gadget_amd")
    sharding = importlib.import_module("bls-verify-gadget_amd.sharding")
    workload = importlib.import_module("bls-verify-gadget_amd.workload")
    n, steps, ring = 64, 3, 2
    assert dist.get_world_size() == world
    # reference digests: direct-mode engine on every rank's batch of every step
    direct = pkg.WitnessEngine(n, 32, max_steps=1, device=dev, n_buffers=1)
    wit = direct.new_witness_tensor()
    res = torch.empty(n, dtype=torch.int32, device=dev)
    want = {}
    inputs = {}
    for k in range(steps):
        for r in range(world):
            pk, msg, sig, _ = workload.make_batch(pkg, n, seed=0x5EED, device=dev, start=(k * world + r) * n)
            inputs[(k, r)] = (pk, msg, sig)
            direct.submit(pk, sig, msg, witness=wit, result=res)
            direct.flush()
            torch.cuda.synchronize()
            want[(k, r)] = pkg.witness_digest(wit).cpu()
    direct.close()
    eng = pkg.WitnessEngine(n, 32, max_steps=2, device=dev, n_buffers=2, consumer_mode=1)
    cbufs = eng.new_compact_buffer(ring)
    results = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(ring)]
    consumer, comm = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    got = {}

    def expand(c):
        eng.expand_compact(c, wit, stream=consumer)
        return wit

    def consume(w, r, k):
        got[(k, r)] = pkg.witness_digest(w, stream=consumer).clone()

    pipe = sharding.CompactGatherPipeline(world, eng.compact_bytes(), dev, expand, consume, comm_stream=comm, consumer_stream=consumer)
    state = {"next": 0}

    def drain():
        while state["next"] < eng.materialised():
            s = state["next"]
            buf = cbufs[s % ring]
            pipe.push(buf, before=lambda st, s=s: eng.wait_step(s, st), after=lambda st, buf=buf: eng.output_consumed(buf, st))
            state["next"] += 1

    for k in range(steps):
        pk, msg, sig = inputs[(k, rank)]
        while True:
            try:
                eng.submit_compact(pk, sig, msg, cbufs[k % ring], result=results[k % ring])
                break
            except pkg.BlswBusy:
                drain()
        drain()
    eng.flush()
    while state["next"] < steps:
        drain()
    pipe.flush()
    torch.cuda.synchronize()
    bad = [key for key in want if not torch.equal(got[key].cpu(), want[key])]
    ok = torch.tensor([0 if bad else 1], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    eng.close()
    if rank == 0:
        print(json.dumps({"world_size": dist.get_world_size(), "backend": dist.get_backend(), "steps": steps, "instances_per_rank_per_step": n,
                          "all_ranks_ok": bool(ok.item()), "mismatches_rank0": [list(b) for b in bad], "order": pipe.order}))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok.item() else 3)


if __name__ == "__main__":
    main()
