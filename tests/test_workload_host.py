"""Host logic of the product-side synthetic workload (bls-verify-gadget_amd/workload.py): the key and message derivation of
SURVEY.md §8d config 2 must be the one tests/synth.py (oracle-backed) uses; the signatures themselves are compared on the
GPU (tests/test_gpu_parity.py::test_sign_batch_fixtures_and_synthetic_workload)."""
import importlib

import numpy as np

from tests import synth


def test_keys_and_messages_match_the_oracle_backed_generator(oracle):
    workload = importlib.import_module("bls-verify-gadget_amd.workload")
    seed = 0x5EED
    assert workload.R_MOD == synth.R_MOD
    sks = workload.secret_keys(seed)
    assert len(sks) == 16 and all(0 < k < workload.R_MOD for k in sks)
    pk, msg, sig, expect = synth.make_batch(oracle, 20, seed=seed)
    m = workload.messages(seed, 0, 20)
    untampered = expect
    assert (m[untampered] == msg[untampered]).all()
    assert (m[~untampered][:, :31] == msg[~untampered][:, :31]).all() and (m[~untampered][:, 31] ^ 1 == msg[~untampered][:, 31]).all()
    # keys: the oracle's pk for sk_k is what the synthetic batch carries
    for i in (0, 5, 17):
        st, xy, _ = oracle.g1_decompress(oracle.sk_to_pk(sks[i % 16]))
        assert st == 0 and (xy == pk[i]).all()


def test_witness_digest_host_definition():
    """witness_digest_reference (the vectorised numpy statement of include/blsw.h's digest: NH multiply-accumulate with a position-derived key + keyed xor sums)
    against a word-by-word restatement of the header's formula; value- and position-sensitivity: one flipped bit changes both words, two swapped pieces too."""
    pkg = importlib.import_module("bls-verify-gadget_amd")
    rng = np.random.default_rng(3)
    w = rng.integers(0, 2**63, size=(333, 6), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(333, 6), dtype=np.uint64)
    w[5] = 0
    w[7] = np.uint64(0xFFFFFFFFFFFFFFFF)
    got = pkg.witness_digest_reference(w)
    x = w.reshape(-1).view(np.uint32).tolist()
    M, K, A = 0xFFFFFFFF, 0x9E3779B1, 0x85EBCA6B
    d0 = lo = hi = 0
    for q in range(len(x) // 4):
        key = ((q + 1) * K) & M
        a, b, c, d = x[4 * q:4 * q + 4]
        d0 += ((a + key) & M) * ((b + key + A) & M) + ((c + key + 2 * A) & M) * ((d + key + 3 * A) & M)
        lo += (a ^ key) + (c ^ (~key & M))
        hi += (b ^ key) + (d ^ (~key & M))
    assert got == [d0 & (2**64 - 1), (lo & M) | ((hi & M) << 32)]
    w2 = w.copy()
    w2[100, 3] ^= np.uint64(1) << np.uint64(40)
    g2 = pkg.witness_digest_reference(w2)
    assert g2[0] != got[0] and g2[1] != got[1]
    w3 = w.copy().reshape(-1, 2)
    w3[[10, 500]] = w3[[500, 10]]  # two 16-byte pieces swapped
    g3 = pkg.witness_digest_reference(w3)
    assert g3[0] != got[0] and g3[1] != got[1]
