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


def run_shard(pkg, *args, **kwargs):
    """the product's shard streamer (bls-verify-gadget_amd/sharding.py: stream_shard)"""
    return importlib.import_module("bls-verify-gadget_amd.sharding").stream_shard(pkg, *args, **kwargs)


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
