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


def make_multi(o, K, seed=0x5EED, msg_len=32, tamper=None, start=0):
    """One instance of the N+1-pair product (BASELINE configs[3]): K distinct (sk_j, msg_j),
    sigma = sum_j sk_j * H(msg_j). `tamper` = index of a message flipped after signing (expected result false).
    -> (pks [K,12] u64, msgs [K,msg_len] u8, sig [24] u64, expect)"""
    key = ("multi", K, seed, msg_len, tamper, start)
    if key in _cache:
        return _cache[key]
    pks = np.zeros((K, 12), dtype=np.uint64)
    msgs = np.zeros((K, msg_len), dtype=np.uint8)
    sigs = []
    for j in range(K):
        sk = int.from_bytes(_h(seed, b"sk", start + j), "big") % R_MOD or 1
        m = (_h(seed, b"mm", start + j) * (msg_len // 32 + 1))[:msg_len]
        st, xy, _ = o.g1_decompress(o.sk_to_pk(sk))
        assert st == 0
        pks[j] = xy
        msgs[j] = np.frombuffer(m, dtype=np.uint8) if msg_len else 0
        sigs.append(o.sign(sk, m))
    st, sxy, inf = o.g2_decompress(o.aggregate_g2(sigs))
    assert st == 0 and not inf
    if tamper is not None:
        msgs[tamper, msg_len - 1] ^= 1
    _cache[key] = (pks, msgs, sxy, tamper is None)
    return _cache[key]


def make_aggregate(o, K, bitmap, seed=0x5EED, start=0, tamper=False):
    """One aggregate_verify instance (constraints.rs:153-167): K keys, one message, sigma = sum over the selected keys of
    sk_j * H(msg). -> (pks [K,12] u64, bitmap [K] u8, msg [32] u8, sig [24] u64, expect)"""
    key = ("agg", K, tuple(bitmap), seed, start, tamper)
    if key in _cache:
        return _cache[key]
    pks = np.zeros((K, 12), dtype=np.uint64)
    m = _h(seed, b"am", start)
    sigs = []
    for j in range(K):
        sk = int.from_bytes(_h(seed, b"sk", start + j), "big") % R_MOD or 1
        st, xy, _ = o.g1_decompress(o.sk_to_pk(sk))
        assert st == 0
        pks[j] = xy
        if bitmap[j]:
            sigs.append(o.sign(sk, m))
    st, sxy, inf = o.g2_decompress(o.aggregate_g2(sigs))
    assert st == 0 and not inf
    msg = np.frombuffer(m, dtype=np.uint8).copy()
    if tamper:
        msg[7] ^= 2
    _cache[key] = (pks, np.array(bitmap, dtype=np.uint8), msg, sxy, not tamper)
    return _cache[key]
