"""Synthetic workloads of SURVEY.md §8d, minted with the product's own GPU signer (blsw_sign_batch), no CPU oracle.

config 2: sk_k = SHA-256(seed || "sk" || k) mod r for 16 distinct keys, msg_i = SHA-256(seed || "m" || i) (32 bytes),
sig_i = sk_(i mod 16) * H(msg_i); every 16th instance has byte 31 of its message flipped AFTER signing (expected false).
tests/synth.py builds the same batch with the CPU oracle; tests/test_gpu_parity.py checks the two agree bit for bit.
"""
import hashlib

import numpy as np

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def _h(seed, tag, i):
    return hashlib.sha256(seed.to_bytes(8, "little") + tag + i.to_bytes(8, "little")).digest()


def secret_keys(seed, n_keys=16):
    return [int.from_bytes(_h(seed, b"sk", k), "big") % R_MOD or 1 for k in range(n_keys)]


def messages(seed, start, n, tag=b"m"):
    return np.frombuffer(b"".join(_h(seed, tag, start + i) for i in range(n)), dtype=np.uint8).reshape(n, 32).copy()


def make_batch(pkg, n, seed=0x5EED, tamper_every=16, device=None, start=0):
    """-> (pk_xy [n,12] int64, msg [n,32] uint8, sig_xy [n,24] int64) cuda tensors, expect [n] numpy bool"""
    import torch

    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    sks = secret_keys(seed)
    sk = np.frombuffer(b"".join(sks[(start + i) % 16].to_bytes(32, "little") for i in range(n)), dtype=np.uint8).reshape(n, 32).copy()
    msg = messages(seed, start, n)
    r = pkg.sign_batch(torch.from_numpy(sk).to(dev), torch.from_numpy(msg).to(dev), want_bytes=False)
    assert int(r["status"].abs().sum().item()) == 0
    expect = np.ones(n, dtype=bool)
    if tamper_every:
        idx = np.arange(n)
        bad = ((start + idx) % tamper_every) == tamper_every - 1
        msg[bad, 31] ^= 1
        expect[bad] = False
    return r["pk_xy"], torch.from_numpy(msg).to(dev), r["sig_xy"], expect
