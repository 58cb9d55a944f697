#!/usr/bin/env python3
"""configs[3] under rocprofv3: one warm-up call and one profiled call of blsw_verify_multi_batch (n instances of K pairs, witnesses
written). Run as: rocprofv3 --kernel-trace --output-format csv -d <dir> -o m -- python3 tools/multi_profile.py [n [K]]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    nm = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    Kp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = torch.device("cuda:0")
    pkg = importlib.import_module("bls-verify-gadget_amd")
    workload = importlib.import_module("bls-verify-gadget_amd.workload")
    msk = workload.secret_keys(0x5EED, 16)
    mm = workload.messages(0x5EED, 1000, Kp, tag=b"mm")
    mr = pkg.sign_batch(torch.from_numpy(np.frombuffer(b"".join(msk[j % 16].to_bytes(32, "little") for j in range(Kp)), dtype=np.uint8).reshape(Kp, 32).copy()).to(dev),
                        torch.from_numpy(mm).to(dev))
    mpks = mr["pk_xy"].unsqueeze(0).repeat(nm, 1, 1).contiguous()
    mmsg = torch.from_numpy(mm).to(dev).unsqueeze(0).repeat(nm, 1, 1).contiguous()
    msig = mr["sig_xy"][0:1].repeat(nm, 1).contiguous()
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, wit = pkg.verify_multi(pkg.ParametersVar(), pkg.PublicKeyVar(mpks), mmsg, pkg.SignatureVar(msig), want_witness=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        del wit  # 4.19 GB per instance: one tensor at a time
        torch.cuda.empty_cache()
    print(json.dumps({"instances": nm, "pairs": Kp, "seconds": dt, "instances_per_s": nm / dt}))


if __name__ == "__main__":
    main()
