#!/usr/bin/env python3
"""blsw_verify_batch (BLS::verify as values, bls.rs:427-458): verdicts per second at three batch sizes. One JSON line each."""
import importlib, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("bls-verify-gadget_amd")
workload = importlib.import_module("bls-verify-gadget_amd.workload")
dev = torch.device("cuda:0")
for n in (16384, 65536, 262144):
    sk = np.frombuffer(b"".join(workload.secret_keys(0x5EED, 16)[i % 16].to_bytes(32, "little") for i in range(n)), dtype=np.uint8).reshape(n, 32).copy()
    msg = torch.from_numpy(workload.messages(0x5EED, 0, n)).to(dev)
    s = pkg.sign_batch(torch.from_numpy(sk).to(dev), msg)
    r = pkg.verify_batch(s["pk48"], msg, s["sig96"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        r = pkg.verify_batch(s["pk48"], msg, s["sig96"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"n": n, "verdicts_per_s": n / dt, "ms": dt * 1e3, "all_true": bool((r == 1).all().item())}))
