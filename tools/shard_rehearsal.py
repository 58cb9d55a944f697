#!/usr/bin/env python3
"""BASELINE configs[2] on the hardware at hand: the procedure of SURVEY.md 8d config 3 for ONE rank's shard.

65 536 instances are sharded contiguously over 8 GPUs (8 192 each). A shard does not fit HBM as witness tensors
(8 192 x 34 MB = 278 GB), so a rank streams it in micro-batches of `batch` instances through a small ring of output
tensors; a consumer (here: the digest kernel blsw_witness_digest, standing in for a per-GPU prover or the all-gather
of a micro-batch) drains every tensor before the engine may overwrite it:

    submit(step k -> ring[k % ring])  ...  engine.wait_step(k, consumer)  ->  digest  ->  engine.output_consumed(ring[k % ring])

The engine runs in consumer mode (options.consumer_mode): groups of `group` steps of chains run ahead into the staging, and a
step is expanded into its ring tensor only once the consumer has released that tensor's previous user — the ring can be much
smaller than a group (`--group 0` = the old scheme: free-running engine, groups of `ring` steps, so that a step's tensor is
released before its group is launched). `check_sample` compares the digests of a fixed-stride sample with the CPU oracle.

Standalone: python tools/shard_rehearsal.py [--rank R --world W --shard 8192 --batch 1024] prints one JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def stream_shard(pkg, n_shard=8192, batch=1024, ring=2, rank=0, world=8, buffers=6, seed=0x5EED, device=None, group=None, ramp=False, keep=None):
    """One rank's shard of a sharded batch (BASELINE configs[2]: 65 536 instances over 8 GPUs = 8 192 per rank) streamed in
    micro-batches of `batch` instances through a ring of `ring` witness tensors: every tensor is drained by a consumer (the digest
    kernel blsw_witness_digest, standing in for a per-GPU prover or the gather of a micro-batch) before the engine may overwrite it.
    group > 0: consumer-mode engine with groups of `group` steps (chains run ahead into the staging, a step is expanded into its
    ring tensor when the consumer has released that tensor's previous user); group = 0: free-running engine with groups of `ring`.
    ramp: the consumer-mode engine starts with groups of 2, 4, 8, ... steps (options.group_ramp); measured useless (a group's chain latency is one
    wave's latency, 59 ms for 2 x 1024 instances and 63 ms for 4 x 1024: profiles/r04_consumer_probe.txt), off by default.
    keep: a dict that carries the engine, its ring and the consumer's stream from one call to the next (same shard geometry): a consumer streams
    many shards through ONE engine, and an engine's fresh streams pay the runtime's one-time costs (per-queue scratch, first dispatches) once.
    The last caller closes keep["eng"].
    Inputs are minted on the GPU before the timed region. -> dict (instances_per_s, digests [n_shard, 2] uint64, ...)."""
    import torch

    workload = importlib.import_module("bls-verify-gadget_amd.workload")
    shard_range = importlib.import_module("bls-verify-gadget_amd.sharding").shard_range
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    lo, hi = shard_range(n_shard * world, rank, world)  # this rank's contiguous block of the global batch
    assert hi - lo == n_shard and n_shard % batch == 0
    steps = n_shard // batch
    if group is None:
        group = max(1, min(16, steps // 2))
    if keep is not None and "eng" in keep:
        eng, outs = keep["eng"], keep["outs"]
    else:
        if group:
            n_groups = (steps + group - 1) // group + (2 if ramp and group > 2 else 0)
            eng = pkg.WitnessEngine(batch, 32, max_steps=group, device=dev, n_buffers=max(2, min(3, n_groups)), consumer_mode=1, group_ramp=1 if ramp else 0)
        else:
            eng = pkg.WitnessEngine(batch, 32, max_steps=ring, device=dev, n_buffers=min(buffers, max(1, steps // ring)))
        outs = [eng.new_witness_tensor() for _ in range(ring)]
    digests = torch.zeros((steps, batch, 2), dtype=torch.int64, device=dev)
    results = torch.zeros((steps, batch), dtype=torch.int32, device=dev)
    # the consumer's stream in the HIGH-priority pool of hardware queues (with the engine's sha / expand / place streams): the runtime backs each
    # priority level with four hardware queues, and a fifth normal-priority stream (null stream + three group buffers' main streams + this one) shares
    # a queue with the null stream — every submit's input-ready marker then queues behind the consumer's waiting digests and the next launch group
    # starts ~70 ms late (profiles/r04_consumer_timeline.txt)
    consumer = keep["consumer"] if keep is not None and "consumer" in keep else torch.cuda.Stream(device=dev, priority=-1)
    if keep is not None:
        keep.update(eng=eng, outs=outs, consumer=consumer)
    base = eng.submitted()  # global step number of this shard's first step (a kept engine has served shards before)
    inputs, expects = [], []
    for k in range(steps):  # inputs are resident before the timed region (minted by the product's signer)
        pk, msg, sig, expect = workload.make_batch(pkg, batch, seed=seed, device=dev, start=lo + k * batch)
        inputs.append((pk, msg, sig))
        expects.append(expect)
    torch.cuda.synchronize(dev)
    state = {"next": 0}

    ev_t0, ev_first = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def drain():
        while state["next"] < eng.materialised() - base:
            s = state["next"]
            eng.wait_step(base + s, consumer)
            if s == 0:
                ev_first.record(consumer)  # the first witness tensor exists: what a consumer waits for before it can start
            pkg.witness_digest(outs[s % ring], out=digests[s], stream=consumer)
            eng.output_consumed(outs[s % ring], consumer)
            state["next"] += 1

    t0 = time.perf_counter()
    ev_t0.record(torch.cuda.current_stream(dev))
    for k in range(steps):
        pk, msg, sig = inputs[k]
        while True:
            try:
                eng.submit(pk, sig, msg, witness=outs[k % ring], result=results[k])
                break
            except pkg.BlswBusy:  # the group buffer still has unwritten steps: their outputs are ours to release
                drain()
        drain()
    eng.flush()
    while state["next"] < steps:
        drain()
    consumer.synchronize()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    res = results.cpu().numpy().astype(bool)
    expect = np.stack(expects)
    if keep is None:
        eng.close()
    return {"rank": rank, "world": world, "first_instance": lo, "n_shard": n_shard, "batch": batch, "ring": ring, "group_steps": group or ring,
            "consumer_mode": bool(group), "group_ramp": bool(group and ramp), "steps": steps, "seconds": dt, "first_step_ms": ev_t0.elapsed_time(ev_first),
            "instances_per_s": n_shard / dt, "results_ok": bool((res == expect).all()), "digests": digests.cpu().numpy().view(np.uint64).reshape(n_shard, 2),
            "inputs": inputs, "sampled": 0}


run_shard = stream_shard


def check_sample(pkg, oracle, out, frac=0.01, threads=8):
    """Digest of every (1 / frac)-th instance of the shard against the oracle's witness vector. -> list of mismatching indices"""
    from concurrent.futures import ThreadPoolExecutor

    n, batch = out["n_shard"], out["batch"]
    stride = max(1, int(round(1.0 / frac)))
    idx = list(range(stride // 2, n, stride))
    host = {}
    for i in idx:
        k = i // batch
        if k not in host:
            pk, msg, sig = out["inputs"][k]
            host[k] = (pk.cpu().numpy().view(np.uint64), msg.cpu().numpy(), sig.cpu().numpy().view(np.uint64))

    def one(i):
        pk, msg, sig = host[i // batch]
        j = i % batch
        _, _, _, w = oracle.witness(pk[j], msg[j].tobytes(), sig[j])  # ctypes releases the GIL
        return i, pkg.witness_digest_reference(w)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        got = list(ex.map(one, idx))
    out["sampled"] = len(idx)
    return [i for i, d in got if out["digests"][i].tolist() != d]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--shard", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--ring", type=int, default=2)
    ap.add_argument("--buffers", type=int, default=6)
    ap.add_argument("--group", type=int, default=None, help="steps per launch group in consumer mode (default min(16, steps / 2); 0 = free-running engine with groups of `ring`)")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    pkg = importlib.import_module("bls-verify-gadget_amd")
    out = run_shard(pkg, args.shard, args.batch, args.ring, args.rank, args.world, args.buffers, group=args.group)
    line = {k: v for k, v in out.items() if k not in ("digests", "inputs")}
    if not args.no_check:
        from tests import oracle_lib  # the CPU restatement: the checker of the sample, nothing else

        bad = check_sample(pkg, oracle_lib.load(), out)
        line.update({"sampled": out["sampled"], "sample_mismatches": len(bad)})
    line["workload"] = "configs[2] rehearsal: one rank's shard (rank %d of %d) streamed through a ring of %d tensors, digest consumer" % (args.rank, args.world, args.ring)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
