"""ctypes loader for oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
ETH = os.path.join(GOLDEN, "ethereum_bls12_381_v0.1.2")

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB

u8p = ctypes.POINTER(ctypes.c_uint8)
u64p = ctypes.POINTER(ctypes.c_uint64)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_witness.restype = ctypes.c_uint64
        lib.orc_witness_aggregate.restype = ctypes.c_uint64
        lib.orc_witness_multi.restype = ctypes.c_uint64
        lib.orc_layout.restype = ctypes.c_uint64
        lib.orc_check_satisfied.restype = ctypes.c_int64

    # ---- helpers
    @staticmethod
    def _buf(b):
        return (ctypes.c_uint8 * len(b)).from_buffer_copy(bytes(b)) if len(b) else (ctypes.c_uint8 * 1)()

    def selfcheck(self):
        return self.lib.orc_selfcheck()

    def g1_decompress(self, b):
        out = (ctypes.c_uint64 * 12)()
        inf = ctypes.c_int(0)
        st = self.lib.orc_g1_decompress(self._buf(b), ctypes.c_size_t(len(b)), out, ctypes.byref(inf))
        return st, np.array(out, dtype=np.uint64), bool(inf.value)

    def g2_decompress(self, b):
        out = (ctypes.c_uint64 * 24)()
        inf = ctypes.c_int(0)
        st = self.lib.orc_g2_decompress(self._buf(b), ctypes.c_size_t(len(b)), out, ctypes.byref(inf))
        return st, np.array(out, dtype=np.uint64), bool(inf.value)

    def g1_compress(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.uint64)
        out = (ctypes.c_uint8 * 48)()
        self.lib.orc_g1_compress(xy.ctypes.data_as(u64p), out)
        return bytes(out)

    def g2_compress(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.uint64)
        out = (ctypes.c_uint8 * 96)()
        self.lib.orc_g2_compress(xy.ctypes.data_as(u64p), out)
        return bytes(out)

    @staticmethod
    def _sk_limbs(sk):
        sk %= R_MOD
        return (ctypes.c_uint64 * 4)(*[(sk >> (64 * i)) & (2**64 - 1) for i in range(4)])

    def sk_to_pk(self, sk):
        out = (ctypes.c_uint8 * 48)()
        self.lib.orc_sk_to_pk(self._sk_limbs(sk), out)
        return bytes(out)

    def sign(self, sk, msg):
        out = (ctypes.c_uint8 * 96)()
        rc = self.lib.orc_sign(self._sk_limbs(sk), self._buf(msg), ctypes.c_size_t(len(msg)), out)
        return None if rc else bytes(out)

    def hash_to_g2(self, msg):
        out = (ctypes.c_uint8 * 96)()
        aff = (ctypes.c_uint64 * 24)()
        self.lib.orc_hash_to_g2(self._buf(msg), ctypes.c_size_t(len(msg)), out, aff)
        return bytes(out), np.array(aff, dtype=np.uint64)

    def expand(self, msg, dst, n):
        out = (ctypes.c_uint8 * n)()
        self.lib.orc_expand(self._buf(msg), ctypes.c_size_t(len(msg)), self._buf(dst), ctypes.c_size_t(len(dst)), ctypes.c_size_t(n), out)
        return bytes(out)

    def verify_bytes(self, pk, msg, sig):
        return bool(
            self.lib.orc_verify_bytes(self._buf(pk), ctypes.c_size_t(len(pk)), self._buf(msg), ctypes.c_size_t(len(msg)), self._buf(sig), ctypes.c_size_t(len(sig)))
        )

    def aggregate_g1(self, pks):
        out = (ctypes.c_uint8 * 48)()
        rc = self.lib.orc_aggregate_g1(self._buf(b"".join(pks)), ctypes.c_size_t(len(pks)), out)
        return None if rc else bytes(out)

    def aggregate_g2(self, sigs):
        out = (ctypes.c_uint8 * 96)()
        rc = self.lib.orc_aggregate_g2(self._buf(b"".join(sigs)), ctypes.c_size_t(len(sigs)), out)
        return None if rc else bytes(out)

    def witness(self, pk_xy, msg, sig_xy, want_vector=True, params_mode=0):
        pk_xy = np.ascontiguousarray(pk_xy, dtype=np.uint64)
        sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
        ncons = ctypes.c_uint64(0)
        res = ctypes.c_int(0)
        if params_mode:  # ParametersVar allocated as witnesses (constraints.rs:198-211 with AllocationMode::Witness)
            self.lib.orc_witness_params.restype = ctypes.c_uint64
            call = lambda w, cap: self.lib.orc_witness_params(pk_xy.ctypes.data_as(u64p), self._buf(msg), ctypes.c_size_t(len(msg)), sig_xy.ctypes.data_as(u64p),
                                                              ctypes.c_int(params_mode), w, ctypes.c_uint64(cap), ctypes.byref(ncons), ctypes.byref(res))
            n = call(None, 0)
            if not want_vector:
                return n, ncons.value, bool(res.value), None
            w = np.zeros((n, 6), dtype=np.uint64)
            call(w.ctypes.data_as(u64p), n)
            return n, ncons.value, bool(res.value), w
        n = self.lib.orc_witness(pk_xy.ctypes.data_as(u64p), self._buf(msg), ctypes.c_size_t(len(msg)), sig_xy.ctypes.data_as(u64p), None, ctypes.c_uint64(0), ctypes.byref(ncons), ctypes.byref(res))
        if not want_vector:
            return n, ncons.value, bool(res.value), None
        w = np.zeros((n, 6), dtype=np.uint64)
        self.lib.orc_witness(pk_xy.ctypes.data_as(u64p), self._buf(msg), ctypes.c_size_t(len(msg)), sig_xy.ctypes.data_as(u64p), w.ctypes.data_as(u64p), ctypes.c_uint64(n), ctypes.byref(ncons), ctypes.byref(res))
        return n, ncons.value, bool(res.value), w

    def witness_io(self, pk_xy, msg, sig_xy, pk_input=False, sig_input=False):
        """the single-key circuit with the key / the signature allocated as PUBLIC INPUTS (constraints.rs:214-249 with AllocationMode::Input):
        -> (n_witness, n_constraints, result, witness [n_witness, 6], instance [1 + 3 pk_input + 6 sig_input, 6] = instance_assignment incl. the leading one)"""
        pk_xy = np.ascontiguousarray(pk_xy, dtype=np.uint64)
        sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
        ncons, res = ctypes.c_uint64(0), ctypes.c_int(0)
        self.lib.orc_witness_io.restype = ctypes.c_uint64
        call = lambda w, cap, inst: self.lib.orc_witness_io(pk_xy.ctypes.data_as(u64p), self._buf(msg), ctypes.c_size_t(len(msg)), sig_xy.ctypes.data_as(u64p), ctypes.c_int(int(pk_input)),
                                                            ctypes.c_int(int(sig_input)), w, ctypes.c_uint64(cap), inst, ctypes.byref(ncons), ctypes.byref(res))
        n = call(None, 0, None)
        w = np.zeros((n, 6), dtype=np.uint64)
        inst = np.zeros((1 + 3 * int(pk_input) + 6 * int(sig_input), 6), dtype=np.uint64)
        call(w.ctypes.data_as(u64p), n, inst.ctypes.data_as(u64p))
        return n, ncons.value, bool(res.value), w, inst

    def layout_io(self, msg_len=32, pk_input=False, sig_input=False):
        starts = (ctypes.c_uint64 * 64)()
        names = ctypes.create_string_buffer(4096)
        nw, nc = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self.lib.orc_layout_io.restype = ctypes.c_uint64
        k = self.lib.orc_layout_io(ctypes.c_size_t(msg_len), ctypes.c_int(int(pk_input)), ctypes.c_int(int(sig_input)), starts, ctypes.c_uint64(64), names, ctypes.c_size_t(4096),
                                   ctypes.byref(nw), ctypes.byref(nc))
        nm = names.value.decode().split("\n")[:k]
        return [(nm[i], starts[i]) for i in range(k)], nw.value, nc.value

    def witness_aggregate(self, pks_xy, bitmap, msg, sig_xy, want_vector=True):
        pks_xy = np.ascontiguousarray(pks_xy, dtype=np.uint64)
        sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
        k = pks_xy.shape[0]
        ncons, res, cnt = ctypes.c_uint64(0), ctypes.c_int(0), ctypes.c_uint32(0)
        starts = (ctypes.c_uint64 * 64)()
        names = ctypes.create_string_buffer(4096)
        args = lambda w, cap: (pks_xy.ctypes.data_as(u64p), bitmap.ctypes.data_as(u8p), ctypes.c_uint64(k), self._buf(msg), ctypes.c_size_t(len(msg)),
                               sig_xy.ctypes.data_as(u64p), w, ctypes.c_uint64(cap), ctypes.byref(ncons), ctypes.byref(res), ctypes.byref(cnt), starts,
                               ctypes.c_uint64(64), names, ctypes.c_size_t(4096))
        n = self.lib.orc_witness_aggregate(*args(None, 0))
        marks = dict(zip(names.value.decode().split("\n"), list(starts)))
        w = None
        if want_vector:
            w = np.zeros((n, 6), dtype=np.uint64)
            self.lib.orc_witness_aggregate(*args(w.ctypes.data_as(u64p), n))
        return n, bool(res.value), cnt.value, marks, w

    def witness_multi(self, pks_xy, msgs, sig_xy, want_vector=True):
        """N+1-pair product circuit: pks_xy [K, 12], msgs [K, msg_len] -> (n_witness, result, marks [(name, start)], witness)"""
        pks_xy = np.ascontiguousarray(pks_xy, dtype=np.uint64)
        sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        k, msg_len = msgs.shape
        ncons, res = ctypes.c_uint64(0), ctypes.c_int(0)
        cap = 16 + 8 * k
        starts = (ctypes.c_uint64 * cap)()
        names = ctypes.create_string_buffer(64 * cap)
        mp = msgs.ctypes.data_as(u8p) if msg_len else (ctypes.c_uint8 * 1)()
        args = lambda w, n: (pks_xy.ctypes.data_as(u64p), mp, ctypes.c_size_t(msg_len), ctypes.c_uint64(k), sig_xy.ctypes.data_as(u64p), w, ctypes.c_uint64(n),
                             ctypes.byref(ncons), ctypes.byref(res), starts, ctypes.c_uint64(cap), names, ctypes.c_size_t(64 * cap))
        n = self.lib.orc_witness_multi(*args(None, 0))
        marks = list(zip(names.value.decode().split("\n"), list(starts)))
        w = None
        if want_vector:
            w = np.zeros((n, 6), dtype=np.uint64)
            self.lib.orc_witness_multi(*args(w.ctypes.data_as(u64p), n))
        return n, bool(res.value), marks, w

    def matrices(self, msg_len=32, n_keys=0, n_pairs=1, params_mode=0, pk_input=False, sig_input=False):
        """(n_constraints, n_witness, [(row_ptr, col, val) for A, B, C]) of the oracle's recorded R1CS for a circuit shape"""
        nnz = (ctypes.c_uint64 * 3)()
        nw = ctypes.c_uint64(0)
        self.lib.orc_matrices.restype = ctypes.c_uint64
        self.lib.orc_matrices_params.restype = ctypes.c_uint64
        self.lib.orc_matrices_io.restype = ctypes.c_uint64
        if pk_input or sig_input:
            assert n_keys == 0 and n_pairs == 1 and not params_mode
            run = lambda a, b, c: self.lib.orc_matrices_io(ctypes.c_size_t(msg_len), ctypes.c_int(int(pk_input)), ctypes.c_int(int(sig_input)), nnz, ctypes.byref(nw), a, b, c)
        elif params_mode:
            assert n_keys == 0 and n_pairs == 1
            run = lambda a, b, c: self.lib.orc_matrices_params(ctypes.c_size_t(msg_len), ctypes.c_int(params_mode), nnz, ctypes.byref(nw), a, b, c)
        else:
            run = lambda a, b, c: self.lib.orc_matrices(ctypes.c_size_t(msg_len), ctypes.c_uint64(n_keys), ctypes.c_uint64(n_pairs), nnz, ctypes.byref(nw), a, b, c)
        nc = run(None, None, None)
        rp = [np.zeros(nc + 1, dtype=np.uint64) for _ in range(3)]
        col = [np.zeros(nnz[m], dtype=np.uint32) for m in range(3)]
        val = [np.zeros((nnz[m], 6), dtype=np.uint64) for m in range(3)]
        u32p = ctypes.POINTER(ctypes.c_uint32)
        a_rp = (u64p * 3)(*[x.ctypes.data_as(u64p) for x in rp])
        a_col = (u32p * 3)(*[x.ctypes.data_as(u32p) for x in col])
        a_val = (u64p * 3)(*[x.ctypes.data_as(u64p) for x in val])
        run(a_rp, a_col, a_val)
        return nc, nw.value, list(zip(rp, col, val))

    def witness_batch(self, pk_xy, msgs, sig_xy, threads=1, want_digests=True):
        pk_xy = np.ascontiguousarray(pk_xy, dtype=np.uint64)
        sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        n, msg_len = msgs.shape
        results = np.zeros(n, dtype=np.int32)
        digests = np.zeros(n, dtype=np.uint64)
        self.lib.orc_witness_batch(
            pk_xy.ctypes.data_as(u64p), msgs.ctypes.data_as(u8p), ctypes.c_size_t(msg_len), sig_xy.ctypes.data_as(u64p), ctypes.c_uint64(n), ctypes.c_int(threads),
            results.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), digests.ctypes.data_as(u64p) if want_digests else None)
        return results.astype(bool), digests

    def layout(self, msg_len=32, params_mode=0):
        starts = (ctypes.c_uint64 * 64)()
        names = ctypes.create_string_buffer(4096)
        nw = ctypes.c_uint64(0)
        nc = ctypes.c_uint64(0)
        self.lib.orc_layout_params.restype = ctypes.c_uint64
        k = self.lib.orc_layout_params(ctypes.c_size_t(msg_len), ctypes.c_int(params_mode), starts, ctypes.c_uint64(64), names, ctypes.c_size_t(4096), ctypes.byref(nw), ctypes.byref(nc))
        nm = names.value.decode().split("\n")[:k]
        return [(nm[i], starts[i]) for i in range(k)], nw.value, nc.value

    def check_satisfied(self, pk_xy, msg, sig_xy, witness=None):
        pk_xy = np.ascontiguousarray(pk_xy, dtype=np.uint64)
        sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
        nc = ctypes.c_uint64(0)
        nnz = ctypes.c_uint64(0)
        wp, nw = None, 0
        if witness is not None:
            witness = np.ascontiguousarray(witness, dtype=np.uint64)
            wp, nw = witness.ctypes.data_as(u64p), witness.shape[0]
        bad = self.lib.orc_check_satisfied(pk_xy.ctypes.data_as(u64p), self._buf(msg), ctypes.c_size_t(len(msg)), sig_xy.ctypes.data_as(u64p), wp, ctypes.c_uint64(nw), ctypes.byref(nc), ctypes.byref(nnz))
        return bad, nc.value, nnz.value

    def trace(self, pk_xy, msg, sig_xy):
        pk_xy = np.ascontiguousarray(pk_xy, dtype=np.uint64)
        sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
        out = np.zeros(24 + 96 + 144, dtype=np.uint64)
        self.lib.orc_trace(pk_xy.ctypes.data_as(u64p), self._buf(msg), ctypes.c_size_t(len(msg)), sig_xy.ctypes.data_as(u64p), out.ctypes.data_as(u64p))
        return {"u0": out[0:12], "u1": out[12:24], "q0": out[24:48], "q1": out[48:72], "r": out[72:96], "h": out[96:120], "f_miller": out[120:192], "f_final": out[192:264]}

    def opcount(self, pk_xy, msg, sig_xy):
        pk_xy = np.ascontiguousarray(pk_xy, dtype=np.uint64)
        sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
        out = (ctypes.c_uint64 * 3)()
        self.lib.orc_opcount(pk_xy.ctypes.data_as(u64p), self._buf(msg), ctypes.c_size_t(len(msg)), sig_xy.ctypes.data_as(u64p), out)
        return {"fp_mul": out[0], "fp_inv": out[1], "sha_blocks": out[2]}


_cached = None


def load():
    global _cached
    if _cached is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            build()
        _cached = Oracle(ctypes.CDLL(so))
    return _cached


def eth_cases(kind):
    d = os.path.join(ETH, kind)
    out = []
    for f in sorted(os.listdir(d)):
        if f.endswith(".json"):
            with open(os.path.join(d, f)) as fh:
                out.append((f, json.load(fh)))
    return out


def unhex(s):
    return bytes.fromhex(s[2:] if s.startswith("0x") else s)
