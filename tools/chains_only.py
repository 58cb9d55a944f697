#!/usr/bin/env python3
"""Throughput of the chain kernels alone (results only, no witness tensors written): the other side of the balance that
bench.py's placement stream sees. One JSON line."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    dev = torch.device("cuda:0")
    pkg = importlib.import_module("bls-verify-gadget_amd")
    workload = importlib.import_module("bls-verify-gadget_amd.workload")
    n, steps, coalesce, buffers = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 16, 3
    pk, msg, sig, expect = workload.make_batch(pkg, 64, device=dev)
    pk, msg, sig = pk.repeat(16, 1).contiguous(), msg.repeat(16, 1).contiguous(), sig.repeat(16, 1).contiguous()
    eng = pkg.WitnessEngine(n, 32, max_steps=coalesce, device=dev, n_buffers=buffers)
    res = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
    for k in range(48):
        eng.submit(pk, sig, msg, witness=None, result=res[k % 2])
    eng.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        eng.submit(pk, sig, msg, witness=None, result=res[k % 2])
    eng.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "chain kernels only (no witness tensors)", "steps": steps, "instances_per_s": n * steps / dt, "ms_per_step": dt / steps * 1e3}))


if __name__ == "__main__":
    main()
