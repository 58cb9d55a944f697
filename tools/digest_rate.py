#!/usr/bin/env python3
"""Rate of the consumer stand-in alone: blsw_witness_digest over one 1024-instance witness tensor (34.8 GB read per launch), HIP events
around 10 launches. Prints one JSON line (GB/s, ms per launch). Under `rocprofv3 --pmc SQ_INSTS_VALU` the same script gives the
kernel's VALU instruction count per launch."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    pkg = importlib.import_module("bls-verify-gadget_amd")
    workload = importlib.import_module("bls-verify-gadget_amd.workload")
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    pk, msg, sig, _ = workload.make_batch(pkg, n, seed=0x5EED, device=dev)
    eng = pkg.WitnessEngine(n, 32, max_steps=2, device=dev, n_buffers=1)
    w = eng.new_witness_tensor()
    eng.submit(pk, sig, msg, witness=w)
    eng.flush()
    torch.cuda.synchronize()
    dig = torch.empty((n, 2), dtype=torch.int64, device=dev)
    pkg.witness_digest(w, out=dig)
    torch.cuda.synchronize()
    ref = pkg.witness_digest_reference(w[3].cpu().numpy().view("uint64"))
    ok = dig[3].cpu().numpy().view("uint64").tolist() == ref
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        pkg.witness_digest(w, out=dig)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    nbytes = n * eng.n_witness * 48
    print(json.dumps({"kernel": "k_digest", "instances": n, "bytes_per_launch": nbytes, "ms_per_launch": ms, "GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000,
                      "matches_host_definition": ok}))
    eng.close()


if __name__ == "__main__":
    main()
