"""ctypes loader for tests/hostsim/libhostsim.so: the device headers compiled for the host (TEST HARNESS ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
u64p = ctypes.POINTER(ctypes.c_uint64)

_FIELDS = (
    "msg_len n_instance_vars n_witness sha_bits off_msg off_pk_alloc off_sig_alloc off_pk_not_zero off_expand off_map0 off_map1 "
    "off_add off_cofactor off_prep_h off_prep_pk off_prep_sig off_miller off_final_exp off_is_one n_keys off_keys off_bitmap off_count off_agg "
    "n_pairs stride_msg stride_pk_alloc stride_pk_not_zero stride_hash stride_prep_h stride_prep_pk "
    "params_mode off_params_alloc off_prep_g1 pk_mode sig_mode"
).split()


class Lay(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in _FIELDS]


_lib = None


def load():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostsim")])
        _lib = ctypes.CDLL(os.path.join(HERE, "hostsim", "libhostsim.so"))
    return _lib


def layout(msg_len, params_mode=0):
    L = Lay()
    load().hostsim_layout_params(msg_len, params_mode, ctypes.byref(L))
    return {n: getattr(L, n) for n in _FIELDS}


def witness(pk_xy, msg, sig_xy, params_mode=0):
    """params_mode 1: ParametersVar allocated as witnesses (constraints.rs:198-211 with AllocationMode::Witness)"""
    pk_xy = np.ascontiguousarray(pk_xy, dtype=np.uint64)
    sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
    lay = layout(len(msg), params_mode)
    out = np.zeros((lay["n_witness"], 6), dtype=np.uint64)
    buf = (ctypes.c_uint8 * max(1, len(msg))).from_buffer_copy(bytes(msg) if len(msg) else b"\0")
    r = load().hostsim_witness_params(pk_xy.ctypes.data_as(u64p), buf, len(msg), sig_xy.ctypes.data_as(u64p), params_mode, out.ctypes.data_as(u64p))
    return r, out


def layout_io(msg_len, pk_mode, sig_mode):
    L = Lay()
    load().hostsim_layout_io(msg_len, pk_mode, sig_mode, ctypes.byref(L))
    return {n: getattr(L, n) for n in _FIELDS}


def witness_io(pk_xy, msg, sig_xy, pk_mode=0, sig_mode=0):
    """PublicKeyVar / SignatureVar allocated as public inputs (AllocationMode::Input): -> (result, witness [n_witness, 6], instance [n_instance_vars, 6])"""
    pk_xy = np.ascontiguousarray(pk_xy, dtype=np.uint64)
    sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
    lay = layout_io(len(msg), pk_mode, sig_mode)
    out = np.zeros((lay["n_witness"], 6), dtype=np.uint64)
    inst = np.zeros((lay["n_instance_vars"], 6), dtype=np.uint64)
    buf = (ctypes.c_uint8 * max(1, len(msg))).from_buffer_copy(bytes(msg) if len(msg) else b"\0")
    r = load().hostsim_witness_io(pk_xy.ctypes.data_as(u64p), buf, len(msg), sig_xy.ctypes.data_as(u64p), pk_mode, sig_mode, out.ctypes.data_as(u64p), inst.ctypes.data_as(u64p))
    return r, out, inst


def witness_aggregate(pks_xy, bitmap, msg, sig_xy):
    pks_xy = np.ascontiguousarray(pks_xy, dtype=np.uint64)
    sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
    bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
    K = pks_xy.shape[0]
    L = Lay()
    buf = (ctypes.c_uint8 * max(1, len(msg))).from_buffer_copy(bytes(msg) if len(msg) else b"\0")
    u8p = ctypes.POINTER(ctypes.c_uint8)
    load().hostsim_witness_aggregate(pks_xy.ctypes.data_as(u64p), bitmap.ctypes.data_as(u8p), K, buf, len(msg), sig_xy.ctypes.data_as(u64p), None, None, ctypes.byref(L))
    out = np.zeros((L.n_witness, 6), dtype=np.uint64)
    cnt = ctypes.c_uint32(0)
    r = load().hostsim_witness_aggregate(pks_xy.ctypes.data_as(u64p), bitmap.ctypes.data_as(u8p), K, buf, len(msg), sig_xy.ctypes.data_as(u64p),
                                         out.ctypes.data_as(u64p), ctypes.byref(cnt), ctypes.byref(L))
    return r, cnt.value, out, {n: getattr(L, n) for n in _FIELDS}


def witness_multi(pks_xy, msgs, sig_xy):
    """N+1-pair product circuit on the host harness: pks_xy [K, 12], msgs [K, msg_len] uint8 -> (result, witness, layout)"""
    pks_xy = np.ascontiguousarray(pks_xy, dtype=np.uint64)
    sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
    msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
    K, msg_len = msgs.shape
    u8p = ctypes.POINTER(ctypes.c_uint8)
    L = Lay()
    mp = msgs.ctypes.data_as(u8p) if msg_len else (ctypes.c_uint8 * 1)()
    load().hostsim_witness_multi(pks_xy.ctypes.data_as(u64p), mp, msg_len, K, sig_xy.ctypes.data_as(u64p), None, ctypes.byref(L))
    out = np.zeros((L.n_witness, 6), dtype=np.uint64)
    r = load().hostsim_witness_multi(pks_xy.ctypes.data_as(u64p), mp, msg_len, K, sig_xy.ctypes.data_as(u64p), out.ctypes.data_as(u64p), ctypes.byref(L))
    return r, out, {n: getattr(L, n) for n in _FIELDS}


def hash_to_g2_values(msg):
    """value-only hash_to_g2 of the device headers (vcurve.hpp) on the host: affine (x.c0, x.c1, y.c0, y.c1) as [24] uint64"""
    out = np.zeros(24, dtype=np.uint64)
    buf = (ctypes.c_uint8 * max(1, len(msg))).from_buffer_copy(bytes(msg) if len(msg) else b"\0")
    load().hostsim_hash_to_g2_values(buf, len(msg), out.ctypes.data_as(u64p))
    return out


def verify_values(pk48, sig96, msg):
    """blsw_verify_batch's per-instance logic on the host: (verdict, status_pk, status_sig)"""
    st = (ctypes.c_int32 * 2)()
    pk = (ctypes.c_uint8 * 48).from_buffer_copy(bytes(pk48).ljust(48, b"\0")[:48])
    sg = (ctypes.c_uint8 * 96).from_buffer_copy(bytes(sig96).ljust(96, b"\0")[:96])
    m = (ctypes.c_uint8 * max(1, len(msg))).from_buffer_copy(bytes(msg) if len(msg) else b"\0")
    r = load().hostsim_verify_values(pk, sg, m, len(msg), st)
    return r, st[0], st[1]


def sign(sk_le32, h_xy):
    """device signer logic on the host: (status, sig96, pk48)"""
    h_xy = np.ascontiguousarray(h_xy, dtype=np.uint64)
    sk = (ctypes.c_uint8 * 32).from_buffer_copy(bytes(sk_le32))
    sig = (ctypes.c_uint8 * 96)()
    pk = (ctypes.c_uint8 * 48)()
    st = load().hostsim_sign(sk, h_xy.ctypes.data_as(u64p), sig, pk)
    return st, bytes(sig), bytes(pk)


def r1cs_check(mats, witness, instance=None):
    """A z o B z = C z for the matrices dict of pkg.matrices() and a witness array [n_witness, 6] uint64 -> first bad row or -1.
    instance: [n_instance_vars, 6] (element 0 = one) for circuits with public inputs: z = [instance | witness]."""
    witness = np.ascontiguousarray(witness, dtype=np.uint64)
    if instance is not None:
        instance = np.ascontiguousarray(instance, dtype=np.uint64)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        rp = (u64p * 3)(*[mats[k][0].ctypes.data_as(u64p) for k in "ABC"])
        col = (u32p * 3)(*[mats[k][1].ctypes.data_as(u32p) for k in "ABC"])
        val = (u64p * 3)(*[mats[k][2].ctypes.data_as(u64p) for k in "ABC"])
        fn = load().hostsim_r1cs_check_io
        fn.restype = ctypes.c_int64
        return fn(ctypes.c_uint64(mats["n_constraints"]), rp, col, val, witness.ctypes.data_as(u64p), ctypes.c_uint64(witness.shape[0]), instance.ctypes.data_as(u64p),
                  ctypes.c_uint64(instance.shape[0]))
    u32p = ctypes.POINTER(ctypes.c_uint32)
    rp = (u64p * 3)(*[mats[k][0].ctypes.data_as(u64p) for k in "ABC"])
    col = (u32p * 3)(*[mats[k][1].ctypes.data_as(u32p) for k in "ABC"])
    val = (u64p * 3)(*[mats[k][2].ctypes.data_as(u64p) for k in "ABC"])
    fn = load().hostsim_r1cs_check
    fn.restype = ctypes.c_int64
    return fn(ctypes.c_uint64(mats["n_constraints"]), rp, col, val, witness.ctypes.data_as(u64p), ctypes.c_uint64(witness.shape[0]))
