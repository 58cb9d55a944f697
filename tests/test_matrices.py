"""Constraint matrices of the product (blsw_matrices_info / blsw_matrices_fill, csrc/r1cs.cpp: a symbolic synthesis that shares
no code with oracle/) against the oracle's recorded R1CS, row by row, and against witness vectors: A z o B z = C z.
The reference never checks satisfaction (constraints.rs:369-373 only prints the system's size); SURVEY.md 8(f).1."""
import importlib

import numpy as np
import pytest

from tests import hostsim_lib, synth


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("bls-verify-gadget_amd")


def _same(mo, mp):
    return all(np.array_equal(x, y) for x, y in zip(mo, mp))


@pytest.mark.parametrize("shape", [(32, 0, 1), (0, 2, 1), (0, 0, 2)])  # single key; aggregate_verify, 2 keys; N+1-pair product, 2 pairs
def test_product_matrices_equal_the_oracles(pkg, oracle, shape):
    nc, nw, M = oracle.matrices(*shape)
    P = pkg.matrices(*shape)
    assert (P["n_constraints"], P["n_witness"], P["n_instance_vars"]) == (nc, nw, 1)
    lay = pkg.layout(shape[0]) if shape[1:] == (0, 1) else (pkg.layout_aggregate(shape[0], shape[1]) if shape[1] else pkg.layout_multi(shape[0], shape[2]))
    assert lay["n_witness"] == nw
    for m, name in enumerate("ABC"):
        assert _same(M[m], P[name]), "matrix %s differs" % name
        rp, col, val = P[name]
        assert rp[0] == 0 and rp[-1] == len(col) and (np.diff(rp.astype(np.int64)) >= 0).all() and int(col.max()) <= nw
    # CSR rows are sorted by column and carry no zero coefficient
    rp, col, val = P["B"]
    inner = np.ones(len(col), dtype=bool)
    inner[rp[1:-1][rp[1:-1] < len(col)]] = False
    assert (np.diff(col.astype(np.int64))[inner[1:]] > 0).all()
    assert val.any(axis=1).all()


def test_params_witness_matrices_equal_the_oracles(pkg, oracle):
    """ParametersVar::new_variable(Witness) (constraints.rs:198-211): the generator's allocation, prepare_g1(-g1) and the variable-point
    ell of the (-g1, sig) pair, row by row against the oracle; a witness vector of that circuit satisfies the system."""
    nc, nw, M = oracle.matrices(32, params_mode=1)
    P = pkg.matrices(32, params_mode="witness")
    assert (P["n_constraints"], P["n_witness"], P["n_instance_vars"]) == (nc, nw, 1)
    assert pkg.layout(32, params_mode=1)["n_witness"] == nw == pkg.layout(32)["n_witness"] + 1942 + 7 + 67 * 8 + 2
    for m, name in enumerate("ABC"):
        assert _same(M[m], P[name]), "matrix %s differs" % name
    pk, msg, sig, expect = synth.make_batch(oracle, 16)
    r, w = hostsim_lib.witness(pk[5], msg[5].tobytes(), sig[5], params_mode=1)
    assert r == 1 and hostsim_lib.r1cs_check(P, w) == -1
    # the Constant-parameters system is a different one
    assert pkg.matrices(32)["n_constraints"] != nc


@pytest.mark.parametrize("pk_mode,sig_mode", [(1, 0), (0, 1), (1, 1)])
def test_public_input_modes(pkg, oracle, pk_mode, sig_mode):
    """PublicKeyVar / SignatureVar::new_variable with AllocationMode::Input (constraints.rs:214-249 take any mode): the point's x, y, z become
    instance variables (no allocation segment, no in-circuit prime-order check). Layout against the oracle's allocation trace; the device chain
    code on the host harness against the oracle's witness AND instance vectors; the product's matrices (instance columns first, then the
    witnesses: ark-relations' numbering) equal to the oracle's row by row, and satisfied by [instance | witness]."""
    marks, nw, nc = oracle.layout_io(32, pk_mode, sig_mode)
    lay = pkg.layout(32, pk_mode=pk_mode, sig_mode=sig_mode)
    base = pkg.layout(32)
    assert lay["n_witness"] == nw == base["n_witness"] - 1942 * pk_mode - 12413 * sig_mode
    assert lay["n_instance_vars"] == 1 + 3 * pk_mode + 6 * sig_mode and (lay["pk_mode"], lay["sig_mode"]) == (pk_mode, sig_mode)
    m = dict(marks)
    for name, field in (("msg", "off_msg"), ("pk_alloc", "off_pk_alloc"), ("sig_alloc", "off_sig_alloc"), ("verify.pk_not_zero", "off_pk_not_zero"), ("hash.expand", "off_expand"),
                        ("hash.map0", "off_map0"), ("hash.add", "off_add")):
        assert m[name] == lay[field], name
    assert hostsim_lib.layout_io(32, pk_mode, sig_mode) == lay
    pk, msg, sig, expect = synth.make_batch(oracle, 16)
    P = pkg.matrices(32, pk_mode=pk_mode, sig_mode=sig_mode)
    nco, nwo, M = oracle.matrices(32, pk_input=pk_mode, sig_input=sig_mode)
    assert (P["n_constraints"], P["n_witness"], P["n_instance_vars"]) == (nc, nw, lay["n_instance_vars"]) and (nco, nwo) == (nc, nw)
    for k, name in enumerate("ABC"):
        assert _same(M[k], P[name]), "matrix %s differs" % name
    assert int(max(P[n][1].max() for n in "ABC")) == lay["n_instance_vars"] + nw - 1  # the last witness's column
    for i in (3, 15):  # a valid and a tampered instance
        n, ncons, res, w, inst = oracle.witness_io(pk[i], msg[i].tobytes(), sig[i], pk_mode, sig_mode)
        assert res == bool(expect[i]) and (n, ncons) == (nw, nc)
        r, hw, hinst = hostsim_lib.witness_io(pk[i], msg[i].tobytes(), sig[i], pk_mode, sig_mode)
        assert bool(r) == res and np.array_equal(hinst, inst)
        bad = np.nonzero((hw != w).any(axis=1))[0]
        assert len(bad) == 0, "first mismatching witness %d" % bad[0]
        assert hostsim_lib.r1cs_check(P, w, inst) == -1
        # the public inputs ARE the point: x, y, one (and the signature's three Fp2 coordinates)
        if pk_mode:
            assert np.array_equal(inst[1], pk[i][:6]) and np.array_equal(inst[2], pk[i][6:]) and np.array_equal(inst[3], inst[0])
        if sig_mode:
            k0 = 1 + 3 * pk_mode
            assert np.array_equal(inst[k0:k0 + 4].reshape(-1), sig[i]) and np.array_equal(inst[k0 + 4], inst[0]) and not inst[k0 + 5].any()
        # a flipped input breaks the system
        bad_inst = inst.copy()
        bad_inst[1, 0] ^= 1
        assert hostsim_lib.r1cs_check(P, w, bad_inst) >= 0
    # identity inputs keep defined values: (0, 1, 0)
    r, hw, hinst = hostsim_lib.witness_io(np.zeros(12, dtype=np.uint64), msg[0].tobytes(), np.zeros(24, dtype=np.uint64), pk_mode, sig_mode)
    n, _, res, w, inst = oracle.witness_io(np.zeros(12, dtype=np.uint64), msg[0].tobytes(), np.zeros(24, dtype=np.uint64), pk_mode, sig_mode)
    assert bool(r) == res and np.array_equal(hinst, inst) and np.array_equal(hw, w)


def test_witness_vectors_satisfy_the_product_matrices(pkg, oracle):
    P = pkg.matrices(32, 0, 1)
    pk, msg, sig, expect = synth.make_batch(oracle, 16)
    for i in (3, 15):  # a valid and a tampered instance: both assignments satisfy the system, the output Boolean differs
        n, _, res, w = oracle.witness(pk[i], msg[i].tobytes(), sig[i])
        assert res == bool(expect[i])
        assert hostsim_lib.r1cs_check(P, w) == -1
    # the host harness (device chain code compiled for the host) as the witness source
    r, w = hostsim_lib.witness(pk[5], msg[5].tobytes(), sig[5])
    assert hostsim_lib.r1cs_check(P, w) == -1
    # a single flipped witness breaks it, and the evaluator says where
    w2 = w.copy()
    w2[P["n_witness"] // 2, 0] ^= 1
    assert hostsim_lib.r1cs_check(P, w2) >= 0


def test_argument_checks(pkg):
    import ctypes

    info = pkg.blsw_matrices_info_t()
    assert pkg.lib().blsw_matrices_info(32, 2, 2, ctypes.byref(info)) == 1  # aggregate and multi together
    assert pkg.lib().blsw_matrices_info_params(32, 2, ctypes.byref(info)) == 1  # AllocationMode::Input of the PARAMETERS is not produced
    assert pkg.lib().blsw_matrices_info_io(32, 2, 0, ctypes.byref(info)) == 1 and pkg.lib().blsw_matrices_info_io(32, 0, 2, ctypes.byref(info)) == 1  # nor Constant keys / signatures
    assert pkg.lib().blsw_matrices_info(32, 0, 0, ctypes.byref(info)) == 1
    assert pkg.lib().blsw_matrices_info(32, 0, 1, None) == 1
