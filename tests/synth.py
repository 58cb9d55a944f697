"""Deterministic synthetic instances: 16 distinct keys, uniform 32-byte messages,
sig = sk * H(msg); every 16th instance has its message tampered after signing (expected result false).
Built with the CPU oracle (test infrastructure): signing needs a CPU hash-to-G2."""
import hashlib

import numpy as np

from tests.oracle_lib import R_MOD

_cache = {}


def _h(seed, tag, i):
    return hashlib.sha256(seed.to_bytes(8, "little") + tag + i.to_bytes(8, "little")).digest()


def make_batch(o, n, seed=0x5EED, tamper_every=16):
    key = (n, seed, tamper_every)
    if key in _cache:
        return _cache[key]
    sks = [int.from_bytes(_h(seed, b"sk", k), "big") % R_MOD or 1 for k in range(16)]
    pks = []
    for sk in sks:
        st, xy, _ = o.g1_decompress(o.sk_to_pk(sk))
        assert st == 0
        pks.append(xy)
    pk = np.zeros((n, 12), dtype=np.uint64)
    sig = np.zeros((n, 24), dtype=np.uint64)
    msg = np.zeros((n, 32), dtype=np.uint8)
    expect = np.ones(n, dtype=bool)
    for i in range(n):
        m = _h(seed, b"m", i)
        s = o.sign(sks[i % 16], m)
        st, sxy, _ = o.g2_decompress(s)
        assert st == 0
        pk[i] = pks[i % 16]
        sig[i] = sxy
        mb = bytearray(m)
        if tamper_every and i % tamper_every == tamper_every - 1:
            mb[31] ^= 1
            expect[i] = False
        msg[i] = np.frombuffer(bytes(mb), dtype=np.uint8)
    _cache[key] = (pk, msg, sig, expect)
    return _cache[key]
