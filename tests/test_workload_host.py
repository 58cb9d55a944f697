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
