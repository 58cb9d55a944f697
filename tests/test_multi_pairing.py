"""N+1-pair product of pairings (BASELINE configs[3]: one signature over K (pk, msg) pairs) without a GPU: the oracle's
generalisation of constraints.rs:90-128 (slices of K + 1 prepared points into product_of_pairings, constraints.rs:121-125)
against the device programs compiled for the host — the single-lane statement and the six-lane team program — plus the
segment table of blsw_layout_multi. K = 1 must reproduce the reference's single-key circuit bit for bit."""
import importlib

import numpy as np
import pytest

from tests import hostsim_lib, synth


def _first_marks(marks):
    out = {}
    for name, start in marks:
        out.setdefault(name, start)
    return out


def test_k1_is_the_single_key_circuit(oracle):
    pk, msg, sig, _ = synth.make_batch(oracle, 16)
    for i in (3, 15):  # valid, tampered
        n, _, res, w = oracle.witness(pk[i], msg[i].tobytes(), sig[i])
        n2, res2, _, w2 = oracle.witness_multi(pk[i:i + 1], msg[i:i + 1], sig[i])
        assert (n, res) == (n2, res2) and np.array_equal(w, w2)


@pytest.mark.parametrize("K,tamper", [(2, None), (3, 1)])
def test_device_programs_on_host_match_oracle(oracle, K, tamper):
    pks, msgs, sig, expect = synth.make_multi(oracle, K, tamper=tamper)
    n, res, marks, w = oracle.witness_multi(pks, msgs, sig)
    assert res == expect
    pkg = importlib.import_module("bls-verify-gadget_amd")
    lay = pkg.layout_multi(32, K)
    assert lay["n_witness"] == n and lay["n_pairs"] == K
    # segment table: first copy of every per-pair segment and the per-pair strides
    m = _first_marks(marks)
    names = {"msg": "off_msg", "pk_alloc": "off_pk_alloc", "sig_alloc": "off_sig_alloc", "verify.pk_not_zero": "off_pk_not_zero", "hash.expand": "off_expand",
             "hash.map0": "off_map0", "hash.map1": "off_map1", "hash.add": "off_add", "hash.clear_cofactor": "off_cofactor", "prepare.h": "off_prep_h",
             "prepare.pk": "off_prep_pk", "prepare.sig": "off_prep_sig", "miller": "off_miller", "final_exp": "off_final_exp", "is_one": "off_is_one"}
    for k, v in names.items():
        assert lay[v] == m[k], k
    expands = [s for nme, s in marks if nme == "hash.expand"]
    assert len(expands) == K and all(expands[j] == lay["off_expand"] + j * lay["stride_hash"] for j in range(K))
    for team in (0, 1):
        hostsim_lib.load().hostsim_use_team(team)
        try:
            r, out, hl = hostsim_lib.witness_multi(pks, msgs, sig)
        finally:
            hostsim_lib.load().hostsim_use_team(0)
        assert hl == lay
        assert r == int(expect)
        bad = np.nonzero((w != out).any(axis=1))[0]
        assert len(bad) == 0, "team=%d: first mismatching witness index %d" % (team, bad[0])


def test_other_message_length(oracle):
    pks, msgs, sig, expect = synth.make_multi(oracle, 2, msg_len=3)
    n, res, _, w = oracle.witness_multi(pks, msgs, sig)
    r, out, lay = hostsim_lib.witness_multi(pks, msgs, sig)
    assert res and r == 1 and lay["n_witness"] == n and np.array_equal(w, out)


@pytest.mark.parametrize("K,chunk,tamper", [(1, 2, None), (3, 2, None), (5, 2, 3), (5, 3, None), (4, 12, None)])
def test_pair_parallel_miller_on_host_matches_oracle(oracle, K, chunk, tamper):
    """miller_par.hpp: the Miller product cut into chunks of pairs (chunk products, prefixes over the chunks, the serial spine of
    squares and ell(sig), the chunks' witnesses from their prefix) gives the serial chain's witnesses bit for bit — chunk
    boundaries inside and at the end of the pair list, one chunk, more pairs than a chunk."""
    pks, msgs, sig, expect = synth.make_multi(oracle, K, tamper=tamper)
    n, res, _, w = oracle.witness_multi(pks, msgs, sig)
    lib = hostsim_lib.load()
    lib.hostsim_use_team(2)
    lib.hostsim_miller_chunk(chunk)
    try:
        r, out, hl = hostsim_lib.witness_multi(pks, msgs, sig)
    finally:
        lib.hostsim_use_team(0)
    assert r == int(expect) == int(res) and hl["n_witness"] == n
    bad = np.nonzero((w != out).any(axis=1))[0]
    assert len(bad) == 0, "first mismatching witness index %d (miller segment starts at %d)" % (bad[0], hl["off_miller"])
