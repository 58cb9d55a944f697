#!/usr/bin/env python3
"""Chain kernels without the HBM streams: the grouped engine with results only (no witness tensor: no SHA bit pass, expansion or placement),
groups of 10 steps of 1024 instances, 3 groups in flight — what DESIGN calls "chains only". Prints instances/s; run under rocprofv3 --pmc for
per-kernel counters (tools/gpu_call23.sh)."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--coalesce", type=int, default=10)
    ap.add_argument("--buffers", type=int, default=3)
    args = ap.parse_args()
    import torch

    pkg = importlib.import_module("bls-verify-gadget_amd")
    workload = importlib.import_module("bls-verify-gadget_amd.workload")
    dev = torch.device("cuda:0")
    n = 1024
    pk, msg, sig, expect = workload.make_batch(pkg, n, seed=0x5EED, device=dev)
    eng = pkg.WitnessEngine(n, 32, max_steps=args.coalesce, device=dev, n_buffers=args.buffers)
    res = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4)]

    def run(k):
        for i in range(k):
            eng.submit(pk, sig, msg, witness=None, result=res[i % 4])
        eng.flush()
        torch.cuda.synchronize()

    run(args.warmup)
    t0 = time.perf_counter()
    run(args.steps)
    dt = time.perf_counter() - t0
    ok = bool((res[0].cpu().numpy().astype(bool) == expect).all())
    print(json.dumps({"workload": "chains only (results, no witness tensors)", "steps": args.steps, "instances_per_s": args.steps * n / dt, "ms_per_step": 1e3 * dt / args.steps, "results_ok": ok}))
    eng.close()


if __name__ == "__main__":
    main()
