"""T2': the oracle's constraint system (A, B, C recorded as arkworks would inline them under
OptimizationGoal::Constraints) is satisfied by its own witness, and by the host-simulated device witness; a corrupted
witness is rejected. This is the strongest self-check available without real arkworks."""
import json
import os

import numpy as np

from tests import hostsim_lib
from tests.oracle_lib import GOLDEN

LIT = json.load(open(os.path.join(GOLDEN, "literals.json")))
OPC = json.load(open(os.path.join(GOLDEN, "oracle_opcount.json")))


def test_r1cs_satisfied_and_counts(oracle):
    g = LIT["gadget_verify"]
    _, pk, _ = oracle.g1_decompress(bytes.fromhex(g["pubkey"]))
    _, sig, _ = oracle.g2_decompress(bytes.fromhex(g["signature"]))
    msg = bytes.fromhex(g["messages"][0])
    bad, ncons, nnz = oracle.check_satisfied(pk, msg, sig)
    assert bad == -1
    assert ncons == OPC["n_constraints"]
    # the device-logic witness satisfies the same system
    r, w = hostsim_lib.witness(pk, msg, sig)
    assert w.shape[0] == OPC["n_witness"]
    bad, _, _ = oracle.check_satisfied(pk, msg, sig, witness=w)
    assert bad == -1
    # a single flipped limb is caught
    w2 = w.copy()
    w2[700000, 0] ^= np.uint64(1)
    bad, _, _ = oracle.check_satisfied(pk, msg, sig, witness=w2)
    assert bad >= 0


def test_opcount_frozen(oracle):
    g = LIT["gadget_verify"]
    _, pk, _ = oracle.g1_decompress(bytes.fromhex(g["pubkey"]))
    _, sig, _ = oracle.g2_decompress(bytes.fromhex(g["signature"]))
    oc = oracle.opcount(pk, bytes.fromhex(g["messages"][0]), sig)
    assert oc["fp_mul"] == OPC["fp_mul"] and oc["fp_inv"] == OPC["fp_inv"] and oc["sha_blocks"] == OPC["sha_blocks"]
