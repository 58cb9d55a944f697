"""bls-verify-gadget_amd — MI355X-native batched witness generation for the BLS12-381 signature-verify gadget.

Host-side mirror (Python over the C ABI of include/blsw.h) of the reference's gadget surface:
    BlsSignatureVerifyGadget::verify(&ParametersVar, &PublicKeyVar, &[UInt8], &SignatureVar) -> Boolean
        (/root/reference/src/constraints.rs:79-128)
    AllocVar::new_variable(.., mode) for ParametersVar / PublicKeyVar / SignatureVar (constraints.rs:194-249)
The product path is the HIP library only: importing works without a GPU (layout is host logic), but every
compute entry point raises if libblsw.so or a HIP device is missing. Nothing here touches oracle/.
"""
import ctypes
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# BLSW_LIB: an alternative build of the same library (A/B runs of compile-time choices, tools/ab_build.sh); default: the in-tree one
LIB_PATH = os.environ.get("BLSW_LIB") or os.path.join(HERE, "libblsw.so")

FP_BYTES = 48
_LAYOUT_FIELDS = (
    "msg_len n_instance_vars n_witness sha_bits off_msg off_pk_alloc off_sig_alloc off_pk_not_zero off_expand off_map0 off_map1 "
    "off_add off_cofactor off_prep_h off_prep_pk off_prep_sig off_miller off_final_exp off_is_one n_keys off_keys off_bitmap off_count off_agg "
    "n_pairs stride_msg stride_pk_alloc stride_pk_not_zero stride_hash stride_prep_h stride_prep_pk "
    "params_mode off_params_alloc off_prep_g1 pk_mode sig_mode"
).split()


class blsw_layout_t(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in _LAYOUT_FIELDS]


class blsw_engine_options_t(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32)] + [(n, ctypes.c_uint32) for n in "n_keys pairing_mode g2_mode expand_variant expand_store prio_mode place_lds consumer_mode output_form chain_variant n_pairs cofactor_mode params_mode group_ramp latency_mode pk_mode sig_mode".split()]


class blsw_matrices_info_t(ctypes.Structure):
    _fields_ = [("n_constraints", ctypes.c_uint64), ("n_instance_vars", ctypes.c_uint64), ("n_witness", ctypes.c_uint64), ("nnz", ctypes.c_uint64 * 3)]


class blsw_matrices_t(ctypes.Structure):
    _fields_ = [("row_ptr", ctypes.POINTER(ctypes.c_uint64) * 3), ("col", ctypes.POINTER(ctypes.c_uint32) * 3), ("val", ctypes.POINTER(ctypes.c_uint64) * 3)]


ERR_BUSY = 6


class BlswError(RuntimeError):
    pass


class BlswBusy(BlswError):
    """consumer_mode engines: the call would have to wait for outputs the caller still holds — drain (wait_step / output_consumed)
    and call again (BLSW_ERR_BUSY)."""


_lib = None


def _load_build_module():
    spec = importlib.util.spec_from_file_location("blsw_build", os.path.join(HERE, "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def build(force=False, verbose=False):
    return _load_build_module().build(force=force, verbose=verbose)


def lib():
    """Loads libblsw.so; fails loudly when the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BlswError("libblsw.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
        L = ctypes.CDLL(LIB_PATH)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.blsw_layout.argtypes = [u32, ctypes.POINTER(blsw_layout_t)]
        L.blsw_engine_workspace_bytes.argtypes = [u64, u32, u32, u32, ctypes.POINTER(u64)]
        L.blsw_engine_create.argtypes = [ctypes.POINTER(vp), u64, u32, u32, u32, vp, u64]
        L.blsw_engine_workspace_bytes_ex.argtypes = [u64, u32, u32, u32, ctypes.POINTER(blsw_engine_options_t), ctypes.POINTER(u64)]
        L.blsw_engine_submit_aggregate.argtypes = [vp, vp, vp, vp, vp, vp, u64, vp, vp, vp]
        L.blsw_engine_create_ex.argtypes = [ctypes.POINTER(vp), u64, u32, u32, u32, ctypes.POINTER(blsw_engine_options_t), vp, u64]
        L.blsw_engine_options_default.argtypes = [ctypes.POINTER(blsw_engine_options_t)]
        L.blsw_engine_submitted.argtypes = [vp, ctypes.POINTER(u64)]
        L.blsw_engine_launched.argtypes = [vp, ctypes.POINTER(u64)]
        L.blsw_engine_materialised.argtypes = [vp, ctypes.POINTER(u64)]
        L.blsw_engine_wait_step.argtypes = [vp, u64, vp]
        L.blsw_engine_output_consumed.argtypes = [vp, vp, vp]
        L.blsw_engine_compact_bytes.argtypes = [vp, ctypes.POINTER(u64)]
        L.blsw_engine_submit_compact.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.blsw_engine_expand_compact.argtypes = [vp, vp, vp, u64, vp]
        L.blsw_engine_submit_aggregate_compact.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.blsw_witness_digest.argtypes = [vp, u64, u64, u32, vp, vp]
        L.blsw_matrices_info.argtypes = [u32, u32, u32, ctypes.POINTER(blsw_matrices_info_t)]
        L.blsw_matrices_fill.argtypes = [u32, u32, u32, ctypes.POINTER(blsw_matrices_info_t), ctypes.POINTER(blsw_matrices_t)]
        L.blsw_matrices_info_params.argtypes = [u32, u32, ctypes.POINTER(blsw_matrices_info_t)]
        L.blsw_matrices_fill_params.argtypes = [u32, u32, ctypes.POINTER(blsw_matrices_info_t), ctypes.POINTER(blsw_matrices_t)]
        L.blsw_layout_params.argtypes = [u32, u32, ctypes.POINTER(blsw_layout_t)]
        L.blsw_layout_multi.argtypes = [u32, u32, ctypes.POINTER(blsw_layout_t)]
        L.blsw_verify_multi_workspace_bytes.argtypes = [u64, u32, u32, ctypes.POINTER(u64)]
        L.blsw_verify_multi_batch.argtypes = [vp, vp, u32, u32, vp, u64, vp, u64, vp, vp, u64, vp]
        L.blsw_engine_destroy.argtypes = [vp]
        L.blsw_engine_submit.argtypes = [vp, vp, vp, vp, vp, u64, vp, vp]
        L.blsw_engine_submit_multi.argtypes = [vp, vp, vp, vp, vp, u64, vp, vp]
        L.blsw_engine_submit_io.argtypes = [vp, vp, vp, vp, vp, vp, u64, vp, vp]
        L.blsw_engine_submit_multi_compact.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.blsw_engine_submit_bytes.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u64, vp, vp]
        L.blsw_engine_flush.argtypes = [vp, vp]
        L.blsw_engine_expand_stats.argtypes = [vp, ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_float)]
        L.blsw_hash_to_g2_workspace_bytes.argtypes = [u64, u32, ctypes.POINTER(u64)]
        L.blsw_hash_to_g2_batch.argtypes = [vp, u32, u64, vp, vp, u64, vp]
        L.blsw_layout_aggregate.argtypes = [u32, u32, ctypes.POINTER(blsw_layout_t)]
        L.blsw_aggregate_workspace_bytes.argtypes = [u64, u32, u32, ctypes.POINTER(u64)]
        L.blsw_aggregate_verify_batch.argtypes = [vp, vp, u32, vp, vp, u32, u64, vp, u64, vp, vp, vp, u64, vp]
        L.blsw_decode_batch.argtypes = [vp, vp, u64, vp, vp, vp, vp]
        L.blsw_aggregate_points_workspace_bytes.argtypes = [u32, u64, u32, ctypes.POINTER(u64)]
        L.blsw_aggregate_points_batch.argtypes = [u32, vp, u32, u64, vp, vp, vp, u64, vp]
        L.blsw_sign_batch.argtypes = [vp, vp, u32, u64, vp, vp, vp, vp, vp, vp, u64, vp]
        L.blsw_verify_workspace_bytes.argtypes = [u64, u32, ctypes.POINTER(u64)]
        L.blsw_verify_batch.argtypes = [vp, vp, vp, u32, u64, vp, vp, vp, u64, vp]
        L.blsw_microbench.argtypes = [ctypes.c_int, u32, u32, ctypes.POINTER(ctypes.c_double)]
        L.blsw_fill_rate.argtypes = [vp, u64, u32, ctypes.POINTER(ctypes.c_double)]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = ["blsw_version", "blsw_layout", "blsw_engine_options_default", "blsw_engine_workspace_bytes", "blsw_engine_workspace_bytes_ex", "blsw_engine_create",
                    "blsw_engine_create_ex", "blsw_engine_destroy", "blsw_engine_submit", "blsw_engine_submit_bytes", "blsw_engine_submit_multi", "blsw_engine_submit_multi_compact", "blsw_engine_submit_aggregate", "blsw_engine_flush", "blsw_engine_submitted", "blsw_engine_launched", "blsw_engine_materialised", "blsw_engine_wait_step",
                    "blsw_engine_output_consumed", "blsw_engine_compact_bytes", "blsw_engine_submit_compact", "blsw_engine_submit_aggregate_compact", "blsw_engine_expand_compact", "blsw_engine_expand_stats", "blsw_witness_digest", "blsw_hash_to_g2_workspace_bytes", "blsw_hash_to_g2_batch",
                    "blsw_decode_batch", "blsw_layout_aggregate", "blsw_aggregate_workspace_bytes", "blsw_aggregate_verify_batch", "blsw_layout_multi",
                    "blsw_verify_multi_workspace_bytes", "blsw_verify_multi_batch", "blsw_matrices_info", "blsw_matrices_fill", "blsw_sign_batch", "blsw_microbench", "blsw_fill_rate", "blsw_layout_io", "blsw_engine_submit_io", "blsw_verify_workspace_bytes", "blsw_verify_batch", "blsw_matrices_info_io", "blsw_matrices_fill_io",
                    "blsw_layout_params", "blsw_matrices_info_params", "blsw_matrices_fill_params", "blsw_aggregate_points_workspace_bytes", "blsw_aggregate_points_batch"]


PARAMS_MODES = {"constant": 0, "witness": 1}
IO_MODES = {"witness": 0, "input": 1, "Witness": 0, "Input": 1}


def layout(msg_len=32, params_mode=0, pk_mode=0, sig_mode=0):
    """Segment table of the witness vector (host logic; replaces cs.num_witness_variables(), constraints.rs:369-373).
    params_mode 1 / "witness": ParametersVar::new_variable with AllocationMode::Witness (constraints.rs:198-211).
    pk_mode / sig_mode 1 / "input": PublicKeyVar / SignatureVar::new_variable with AllocationMode::Input (constraints.rs:214-249): the point's
    coordinates are public inputs (n_instance_vars > 1), its allocation segment is empty."""
    L = blsw_layout_t()
    params_mode = PARAMS_MODES.get(params_mode, params_mode)
    pk_mode, sig_mode = IO_MODES.get(pk_mode, pk_mode), IO_MODES.get(sig_mode, sig_mode)
    if pk_mode or sig_mode:
        if params_mode:
            raise BlswError("pk_mode / sig_mode Input apply to the circuit with Constant parameters")
        rc = lib().blsw_layout_io(msg_len, pk_mode, sig_mode, ctypes.byref(L))
    else:
        rc = lib().blsw_layout_params(msg_len, params_mode, ctypes.byref(L)) if params_mode else lib().blsw_layout(msg_len, ctypes.byref(L))
    if rc:
        raise BlswError("blsw_layout failed: %d" % rc)
    return {n: getattr(L, n) for n in _LAYOUT_FIELDS}


def engine_workspace_bytes(n, msg_len=32, max_steps=1, n_buffers=1):
    b = ctypes.c_uint64(0)
    rc = lib().blsw_engine_workspace_bytes(n, msg_len, max_steps, n_buffers, ctypes.byref(b))
    if rc:
        raise BlswError("blsw_engine_workspace_bytes failed: %d" % rc)
    return b.value


def _require_cuda():
    import torch

    if not torch.cuda.is_available():
        raise BlswError("no HIP device visible: the witness path runs only on the GPU (there is no CPU fallback)")
    return torch


def check_capacity(device, what, *byte_counts):
    """Refuses (BlswError) before anything is allocated when the buffers a direct call needs — its workspace and, with want_witness, n witness
    vectors (a 128-pair instance is 4.19 GB) — exceed the free HBM of `device`: a clear error instead of an allocator exception half-way."""
    torch = _require_cuda()
    need = int(sum(byte_counts))
    free_b, _ = torch.cuda.mem_get_info(device)
    # memory torch has cached but not in use is reusable by the allocations that follow
    reusable = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    if need > free_b + reusable:
        raise BlswError("%s needs %.2f GB of HBM (workspace + witness vectors) and %.2f GB are free on %s: pass fewer instances per call, "
                        "want_witness=False, or stream the batch through a WitnessEngine with a small ring of outputs" % (what, need / 1e9, (free_b + reusable) / 1e9, device))


def engine_options(**overrides):
    """blsw_engine_options_default with keyword overrides: device, pairing_mode ("team"/"lane" or 0/1), g2_mode ("lane"/"team"
    or 0/1), expand_variant, expand_store, prio_mode, place_lds, consumer_mode, output_form, n_keys. The library itself reads no
    environment for its options (one diagnostic: BLSW_TRACE_GROUP=1 prints every launch group's stage times at engine destruction); for A/B runs of measurement scripts THIS function applies BLSW_PAIRING=lane, BLSW_G2=team, BLSW_EXPAND_VARIANT,
    BLSW_EXPAND_NT, BLSW_PRIO_MODE, BLSW_PLACE_LDS (explicit keyword arguments win)."""
    o = blsw_engine_options_t()
    rc = lib().blsw_engine_options_default(ctypes.byref(o))
    if rc:
        raise BlswError("blsw_engine_options_default failed: %d" % rc)
    env = os.environ
    if env.get("BLSW_PAIRING", "")[:1] == "l":
        o.pairing_mode = 1
    if env.get("BLSW_G2", "")[:1] == "t" and o.pairing_mode == 0:
        o.g2_mode = 1
    for var, field in (("BLSW_CHAIN_VARIANT", "chain_variant"), ("BLSW_COFACTOR_MODE", "cofactor_mode"), ("BLSW_EXPAND_VARIANT", "expand_variant"), ("BLSW_EXPAND_NT", "expand_store"), ("BLSW_PRIO_MODE", "prio_mode"), ("BLSW_PLACE_LDS", "place_lds"), ("BLSW_GROUP_RAMP", "group_ramp"), ("BLSW_LATENCY_MODE", "latency_mode")):
        if env.get(var):
            setattr(o, field, int(env[var]))
    names = {"pairing_mode": {"team": 0, "lane": 1}, "g2_mode": {"lane": 0, "team": 1}, "params_mode": PARAMS_MODES, "pk_mode": IO_MODES, "sig_mode": IO_MODES}
    for k, v in overrides.items():
        if v is None:
            continue
        if not hasattr(o, k):
            raise BlswError("unknown engine option %r" % k)
        setattr(o, k, names.get(k, {}).get(v, v))
    return o


class WitnessEngine:
    """Thin wrapper of blsw_engine_*: submit batches, flush, read results. max_steps batches are fused per launch group.
    Streaming consumers use the step numbers returned by submit(): wait_step(seq) / output_consumed(tensor)."""

    def __init__(self, n, msg_len=32, max_steps=1, device=None, n_buffers=None, reserve_bytes=0, **options):
        """options: fields of blsw_engine_options_t; n_keys=K makes it an aggregate_verify engine (submit_aggregate).
        reserve_bytes: bytes the caller will allocate next to the workspace (its witness tensors): the capacity check, made with the byte count
        blsw_engine_workspace_bytes_ex returns for THESE options and this n_buffers, covers them too (BlswError instead of an allocator exception)."""
        torch = _require_cuda()
        self.torch = torch
        self.n, self.msg_len, self.max_steps = int(n), int(msg_len), int(max_steps)
        self.n_buffers = int(n_buffers) if n_buffers is not None else (3 if self.max_steps > 1 else 1)
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        opt = engine_options(**options)
        opt.device = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.n_keys = int(opt.n_keys)
        self.n_pairs = int(opt.n_pairs) if opt.n_pairs > 1 else 1
        self.layout = layout_aggregate(msg_len, self.n_keys) if self.n_keys else (layout_multi(msg_len, self.n_pairs) if self.n_pairs > 1 else layout(msg_len, int(opt.params_mode), int(opt.pk_mode), int(opt.sig_mode)))
        self.n_witness = self.layout["n_witness"]
        self.n_instance_vars = self.layout["n_instance_vars"]
        wb = ctypes.c_uint64(0)
        rc = lib().blsw_engine_workspace_bytes_ex(self.n, self.msg_len, self.max_steps, self.n_buffers, ctypes.byref(opt), ctypes.byref(wb))
        if rc:
            raise BlswError("blsw_engine_workspace_bytes_ex failed: %d" % rc)
        check_capacity(self.device, "WitnessEngine (n = %d, max_steps = %d, n_buffers = %d)" % (self.n, self.max_steps, self.n_buffers), wb.value, reserve_bytes)
        self.workspace = torch.empty(wb.value, dtype=torch.uint8, device=self.device)
        self._e = ctypes.c_void_p()
        rc = lib().blsw_engine_create_ex(ctypes.byref(self._e), self.n, self.msg_len, self.max_steps, self.n_buffers, ctypes.byref(opt), self.workspace.data_ptr(),
                                         self.workspace.numel())
        if rc:
            self._e = None
            raise BlswError("blsw_engine_create_ex failed: %d" % rc)
        self._keep = []

    def close(self):
        if getattr(self, "_e", None):
            lib().blsw_engine_destroy(self._e)
            self._e = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def new_witness_tensor(self):
        return self.torch.empty((self.n, self.n_witness, 6), dtype=self.torch.int64, device=self.device)

    def new_instance_tensor(self):
        """[n, n_instance_vars, 6]: instance_assignment of every instance (submit(..., instance=...); element 0 = one)"""
        return self.torch.empty((self.n, self.n_instance_vars, 6), dtype=self.torch.int64, device=self.device)

    def _stream(self, stream):
        return (stream if stream is not None else self.torch.cuda.current_stream(self.device)).cuda_stream

    def submit(self, pk_xy, sig_xy, msg, witness=None, result=None, stream=None, instance=None):
        """-> step number (0, 1, 2, ... in submission order). instance: [n, n_instance_vars, 6] tensor that receives instance_assignment
        (blsw_engine_submit_io; pk_mode / sig_mode Input engines: the public inputs an arkworks verifier takes)."""
        assert pk_xy.is_cuda and sig_xy.is_cuda and msg.is_cuda
        assert pk_xy.shape == (self.n, 12) and sig_xy.shape == (self.n, 24) and msg.shape == (self.n, self.msg_len)
        assert pk_xy.is_contiguous() and sig_xy.is_contiguous() and msg.is_contiguous()
        if witness is not None:
            assert witness.is_contiguous() and witness.shape[0] == self.n and witness.shape[1] >= self.n_witness
        if result is not None:
            assert result.is_contiguous() and result.numel() >= self.n
        seq = self.submitted()
        if instance is not None:
            assert instance.is_contiguous() and tuple(instance.shape) == (self.n, self.n_instance_vars, 6)
            rc = lib().blsw_engine_submit_io(self._e, pk_xy.data_ptr(), sig_xy.data_ptr(), msg.data_ptr() if self.msg_len else None, instance.data_ptr(),
                                             witness.data_ptr() if witness is not None else None, witness.shape[1] if witness is not None else 0,
                                             result.data_ptr() if result is not None else None, self._stream(stream))
        else:
            rc = lib().blsw_engine_submit(self._e, pk_xy.data_ptr(), sig_xy.data_ptr(), msg.data_ptr() if self.msg_len else None,
                                          witness.data_ptr() if witness is not None else None, witness.shape[1] if witness is not None else 0,
                                          result.data_ptr() if result is not None else None, self._stream(stream))
        if rc:
            raise (BlswBusy if rc == ERR_BUSY else BlswError)("blsw_engine_submit failed: %d" % rc)
        self._keep.append((pk_xy, sig_xy, msg, witness, result))
        self._keep = self._keep[-(self.n_buffers + 1) * self.max_steps:]
        return seq

    def submit_bytes(self, pk48, sig96, msg, witness=None, result=None, stream=None):
        """blsw_engine_submit_bytes: compressed points [n, 48] / [n, 96] uint8 -> (step number, pk_xy, sig_xy, status [n, 2]);
        result[i] = 1 iff both points decode to non-identity subgroup points and the gadget's Boolean is true (tests.rs:244-263)."""
        torch = self.torch
        assert pk48.shape == (self.n, 48) and sig96.shape == (self.n, 96) and msg.shape == (self.n, self.msg_len)
        assert pk48.is_contiguous() and sig96.is_contiguous() and msg.is_contiguous() and pk48.dtype == torch.uint8 and sig96.dtype == torch.uint8
        if witness is not None:
            assert witness.is_contiguous() and witness.shape[0] == self.n and witness.shape[1] >= self.n_witness
        pk_xy = torch.empty((self.n, 12), dtype=torch.int64, device=self.device)
        sig_xy = torch.empty((self.n, 24), dtype=torch.int64, device=self.device)
        status = torch.empty((self.n, 2), dtype=torch.int32, device=self.device)
        seq = self.submitted()
        rc = lib().blsw_engine_submit_bytes(self._e, pk48.data_ptr(), sig96.data_ptr(), msg.data_ptr() if self.msg_len else None, pk_xy.data_ptr(), sig_xy.data_ptr(),
                                            status.data_ptr(), witness.data_ptr() if witness is not None else None, witness.shape[1] if witness is not None else 0,
                                            result.data_ptr() if result is not None else None, self._stream(stream))
        if rc:
            raise (BlswBusy if rc == ERR_BUSY else BlswError)("blsw_engine_submit_bytes failed: %d" % rc)
        self._keep.append((pk48, sig96, msg, pk_xy, sig_xy, status, witness, result))
        self._keep = self._keep[-(self.n_buffers + 1) * self.max_steps:]
        return seq, pk_xy, sig_xy, status

    def submit_multi(self, pks_xy, msgs, sig_xy, witness=None, result=None, stream=None):
        """N+1-pair product batch (engine created with n_pairs=K): pks_xy [n, K, 12] int64, msgs [n, K, msg_len] uint8, sig_xy [n, 24] -> step number"""
        K = self.n_pairs
        assert K > 1 and pks_xy.shape == (self.n, K, 12) and msgs.shape == (self.n, K, self.msg_len) and sig_xy.shape == (self.n, 24)
        assert pks_xy.is_contiguous() and msgs.is_contiguous() and sig_xy.is_contiguous()
        if witness is not None:
            assert witness.is_contiguous() and witness.shape[0] == self.n and witness.shape[1] >= self.n_witness
        seq = self.submitted()
        rc = lib().blsw_engine_submit_multi(self._e, pks_xy.data_ptr(), msgs.data_ptr() if self.msg_len else None, sig_xy.data_ptr(),
                                            witness.data_ptr() if witness is not None else None, witness.shape[1] if witness is not None else 0,
                                            result.data_ptr() if result is not None else None, self._stream(stream))
        if rc:
            raise (BlswBusy if rc == ERR_BUSY else BlswError)("blsw_engine_submit_multi failed: %d" % rc)
        self._keep.append((pks_xy, msgs, sig_xy, witness, result))
        self._keep = self._keep[-(self.n_buffers + 1) * self.max_steps:]
        return seq

    def submit_multi_compact(self, pks_xy, msgs, sig_xy, compact, result=None, stream=None):
        """submit_multi with the step's compact wire form in `compact` (uint8 tensor of compact_bytes()) as its output -> step number"""
        K = self.n_pairs
        assert K > 1 and pks_xy.shape == (self.n, K, 12) and msgs.shape == (self.n, K, self.msg_len) and sig_xy.shape == (self.n, 24)
        assert pks_xy.is_contiguous() and msgs.is_contiguous() and sig_xy.is_contiguous()
        assert compact.is_cuda and compact.is_contiguous() and compact.dtype.itemsize == 1 and compact.numel() >= self.compact_bytes()
        seq = self.submitted()
        rc = lib().blsw_engine_submit_multi_compact(self._e, pks_xy.data_ptr(), msgs.data_ptr() if self.msg_len else None, sig_xy.data_ptr(), compact.data_ptr(),
                                                    result.data_ptr() if result is not None else None, self._stream(stream))
        if rc:
            raise (BlswBusy if rc == ERR_BUSY else BlswError)("blsw_engine_submit_multi_compact failed: %d" % rc)
        self._keep.append((pks_xy, msgs, sig_xy, compact, result))
        self._keep = self._keep[-(self.n_buffers + 1) * self.max_steps:]
        return seq

    def submit_aggregate(self, pks_xy, bitmap, sig_xy, msg, witness=None, result=None, count=None, stream=None):
        """aggregate_verify batch (engine created with n_keys=K): pks_xy [n, K, 12] int64, bitmap [n, K] uint8 -> step number"""
        K = self.n_keys
        assert K and pks_xy.shape == (self.n, K, 12) and bitmap.shape == (self.n, K) and sig_xy.shape == (self.n, 24) and msg.shape == (self.n, self.msg_len)
        assert pks_xy.is_contiguous() and bitmap.is_contiguous() and sig_xy.is_contiguous() and msg.is_contiguous()
        if witness is not None:
            assert witness.is_contiguous() and witness.shape[0] == self.n and witness.shape[1] >= self.n_witness
        seq = self.submitted()
        rc = lib().blsw_engine_submit_aggregate(self._e, pks_xy.data_ptr(), bitmap.data_ptr(), sig_xy.data_ptr(), msg.data_ptr() if self.msg_len else None,
                                                witness.data_ptr() if witness is not None else None, witness.shape[1] if witness is not None else 0,
                                                result.data_ptr() if result is not None else None, count.data_ptr() if count is not None else None, self._stream(stream))
        if rc:
            raise (BlswBusy if rc == ERR_BUSY else BlswError)("blsw_engine_submit_aggregate failed: %d" % rc)
        self._keep.append((pks_xy, bitmap, sig_xy, msg, witness, result, count))
        self._keep = self._keep[-(self.n_buffers + 1) * self.max_steps:]
        return seq

    def compact_bytes(self):
        """Bytes of one batch in compact wire form (bit-packed SHA witnesses + staged field witnesses, ~2.6 MB per instance)."""
        return self._counter(lib().blsw_engine_compact_bytes)

    def new_compact_buffer(self, batches=1):
        import torch

        return torch.empty((batches, self.compact_bytes()), dtype=torch.uint8, device=self.device)

    def submit_compact(self, pk_xy, sig_xy, msg, compact, result=None, stream=None):
        """As submit(), but the step's output is its compact wire form in `compact` (uint8 tensor of compact_bytes()) -> step number"""
        assert pk_xy.shape == (self.n, 12) and sig_xy.shape == (self.n, 24) and msg.shape == (self.n, self.msg_len)
        assert pk_xy.is_contiguous() and sig_xy.is_contiguous() and msg.is_contiguous()
        assert compact.is_cuda and compact.is_contiguous() and compact.dtype.itemsize == 1 and compact.numel() >= self.compact_bytes()
        seq = self.submitted()
        rc = lib().blsw_engine_submit_compact(self._e, pk_xy.data_ptr(), sig_xy.data_ptr(), msg.data_ptr() if self.msg_len else None, compact.data_ptr(),
                                              result.data_ptr() if result is not None else None, self._stream(stream))
        if rc:
            raise (BlswBusy if rc == ERR_BUSY else BlswError)("blsw_engine_submit_compact failed: %d" % rc)
        self._keep.append((pk_xy, sig_xy, msg, compact, result))
        self._keep = self._keep[-(self.n_buffers + 1) * self.max_steps:]
        return seq

    def submit_aggregate_compact(self, pks_xy, bitmap, sig_xy, msg, compact, result=None, count=None, stream=None):
        """submit_aggregate with the step's compact wire form in `compact` as its output -> step number"""
        K = self.n_keys
        assert K and pks_xy.shape == (self.n, K, 12) and bitmap.shape == (self.n, K) and sig_xy.shape == (self.n, 24) and msg.shape == (self.n, self.msg_len)
        assert pks_xy.is_contiguous() and bitmap.is_contiguous() and sig_xy.is_contiguous() and msg.is_contiguous()
        assert compact.is_cuda and compact.is_contiguous() and compact.dtype.itemsize == 1 and compact.numel() >= self.compact_bytes()
        seq = self.submitted()
        rc = lib().blsw_engine_submit_aggregate_compact(self._e, pks_xy.data_ptr(), bitmap.data_ptr(), sig_xy.data_ptr(), msg.data_ptr() if self.msg_len else None,
                                                        compact.data_ptr(), result.data_ptr() if result is not None else None,
                                                        count.data_ptr() if count is not None else None, self._stream(stream))
        if rc:
            raise (BlswBusy if rc == ERR_BUSY else BlswError)("blsw_engine_submit_aggregate_compact failed: %d" % rc)
        self._keep.append((pks_xy, bitmap, sig_xy, msg, compact, result, count))
        self._keep = self._keep[-(self.n_buffers + 1) * self.max_steps:]
        return seq

    def expand_compact(self, compact, witness, stream=None):
        """Receiver side: one batch in compact form (this engine's or another rank's) -> its n witness vectors in `witness`."""
        assert compact.is_cuda and compact.is_contiguous() and compact.numel() >= self.compact_bytes()
        assert witness.is_contiguous() and witness.shape[0] == self.n and witness.shape[1] >= self.n_witness
        rc = lib().blsw_engine_expand_compact(self._e, compact.data_ptr(), witness.data_ptr(), witness.shape[1], self._stream(stream))
        if rc:
            raise BlswError("blsw_engine_expand_compact failed: %d" % rc)

    def flush(self, stream=None):
        rc = lib().blsw_engine_flush(self._e, self._stream(stream))
        if rc:
            raise BlswError("blsw_engine_flush failed: %d" % rc)

    def _counter(self, fn):
        v = ctypes.c_uint64(0)
        rc = fn(self._e, ctypes.byref(v))
        if rc:
            raise BlswError("engine counter failed: %d" % rc)
        return v.value

    def submitted(self):
        return self._counter(lib().blsw_engine_submitted)

    def launched(self):
        return self._counter(lib().blsw_engine_launched)

    def materialised(self):
        """Steps whose output writes have been issued (= launched() unless consumer_mode holds steps back for their outputs)."""
        return self._counter(lib().blsw_engine_materialised)

    def wait_step(self, seq, stream=None):
        """Makes `stream` (default: the current stream) wait for step `seq`'s witness tensor and results (seq < launched())."""
        rc = lib().blsw_engine_wait_step(self._e, seq, self._stream(stream))
        if rc:
            raise (BlswBusy if rc == ERR_BUSY else BlswError)("blsw_engine_wait_step failed: %d" % rc)

    def output_consumed(self, witness, stream=None):
        """The consumer is done with `witness` once `stream` reaches this point: the next step submitted with the same tensor
        does not overwrite it earlier."""
        rc = lib().blsw_engine_output_consumed(self._e, witness.data_ptr(), self._stream(stream))
        if rc:
            raise (BlswBusy if rc == ERR_BUSY else BlswError)("blsw_engine_output_consumed failed: %d" % rc)

    def expand_stats(self):
        """(number of k_sha_expand launches since the last call, their average duration in ms); synchronises with them."""
        ms, cnt = ctypes.c_float(0), ctypes.c_uint32(0)
        rc = lib().blsw_engine_expand_stats(self._e, ctypes.byref(cnt), ctypes.byref(ms))
        if rc:
            raise BlswError("blsw_engine_expand_stats failed: %d" % rc)
        return cnt.value, ms.value


class ParametersVar:
    """constraints.rs:23-28, AllocVar at :194-212: the default generator, allocated as a Constant (every circuit of the reference) or
    as a Witness (new_witness: the generator goes through G1Var::new_variable like a public key). AllocationMode::Input would put it
    into instance_assignment, which the engine does not produce."""

    def __init__(self, mode="Constant"):
        if mode not in ("Constant", "Witness"):
            raise BlswError("ParametersVar: AllocationMode %r is not on the GPU path (Constant or Witness)" % (mode,))
        self.mode = mode

    @classmethod
    def new_constant(cls):
        return cls("Constant")

    @classmethod
    def new_witness(cls):
        return cls("Witness")


class PublicKeyVar:
    """constraints.rs:39-44, AllocVar at :214-232: Witness (the reference's circuits) or Input (new_input: the key's x, y, z are public inputs,
    no in-circuit subgroup check). `xy`: [n, 12] int64 tensor (u64 limbs: x, y Montgomery). AllocationMode::Constant is not on the GPU path."""

    def __init__(self, xy, mode="Witness"):
        if mode not in ("Witness", "Input"):
            raise BlswError("PublicKeyVar: AllocationMode %r is not on the GPU path (Witness or Input)" % (mode,))
        self.xy = xy
        self.mode = mode

    @classmethod
    def new_witness(cls, xy):
        return cls(xy)

    @classmethod
    def new_input(cls, xy):
        return cls(xy, "Input")


class SignatureVar:
    """constraints.rs:55-60, AllocVar at :234-249: Witness or Input (new_input). `xy`: [n, 24] int64 tensor (x.c0, x.c1, y.c0, y.c1)."""

    def __init__(self, xy, mode="Witness"):
        if mode not in ("Witness", "Input"):
            raise BlswError("SignatureVar: AllocationMode %r is not on the GPU path (Witness or Input)" % (mode,))
        self.xy = xy
        self.mode = mode

    @classmethod
    def new_witness(cls, xy):
        return cls(xy)

    @classmethod
    def new_input(cls, xy):
        return cls(xy, "Input")


class BlsSignatureVerifyGadget:
    """Batched counterpart of constraints.rs:79-128. One call = n independent circuits (direct mode engine, one batch)."""

    def __init__(self, n, msg_len=32, device=None, want_witness=True, max_steps=1, **options):
        """options: blsw_engine_options_t fields; params_mode="witness" builds the circuit for ParametersVar.new_witness(), pk_mode / sig_mode="input" the
        one for PublicKeyVar.new_input / SignatureVar.new_input (self.instance then holds every instance's instance_assignment after verify)."""
        reserve = 0
        if want_witness and not options.get("n_keys") and not options.get("n_pairs"):
            reserve = n * layout(msg_len, PARAMS_MODES.get(options.get("params_mode"), options.get("params_mode") or 0), options.get("pk_mode") or 0, options.get("sig_mode") or 0)["n_witness"] * 48
        self.engine = WitnessEngine(n, msg_len, max_steps=max_steps, device=device, reserve_bytes=reserve, **options)
        torch = self.engine.torch
        self.torch = torch
        self.n, self.msg_len, self.device = self.engine.n, self.engine.msg_len, self.engine.device
        self.layout = self.engine.layout
        self.n_witness = self.engine.n_witness
        self.result = torch.empty(self.n, dtype=torch.int32, device=self.device)
        self.witness = self.engine.new_witness_tensor() if want_witness else None
        self.instance = self.engine.new_instance_tensor() if self.layout["n_instance_vars"] > 1 else None

    def verify(self, parameters, public_key, message, signature, witness=None, stream=None):
        """message: [n, msg_len] uint8 tensor. Returns the int32 result tensor (gadget Boolean per instance); the witness
        vectors are in self.witness (or the tensor passed as `witness`)."""
        assert isinstance(parameters, ParametersVar)
        if (parameters.mode == "Witness") != bool(self.layout["params_mode"]):
            raise BlswError("ParametersVar mode %s does not match the circuit this gadget was built for (params_mode=%d)" % (parameters.mode, self.layout["params_mode"]))
        for var, field in ((public_key, "pk_mode"), (signature, "sig_mode")):
            if (getattr(var, "mode", "Witness") == "Input") != bool(self.layout[field]):
                raise BlswError("%s allocated as %s does not match the circuit this gadget was built for (%s=%d)" % (type(var).__name__, var.mode, field, self.layout[field]))
        w = witness if witness is not None else self.witness
        self.engine.submit(public_key.xy, signature.xy, message, witness=w, result=self.result, stream=stream, instance=self.instance)
        self.engine.flush(stream=stream)
        return self.result


def verify_mixed_lengths(parameters, public_key, messages, signature, want_witness=True, **options):
    """`verify` for a batch whose messages differ in LENGTH (constraints.rs:90-95 takes any `&[UInt8]` per call; the circuit — its SHA-256
    block count, hence n_witness and the matrices — is a function of the length, so a batch shares a layout only per length): the
    instances are grouped by message length and each group goes through its own gadget (one engine per distinct length).
    public_key.xy [n, 12], signature.xy [n, 24] cuda tensors; messages: a sequence of n bytes-like objects or 1-D uint8 tensors.
    Returns (result int32 [n], witnesses): witnesses[i] is instance i's [n_witness(len_i), 6] int64 vector (a view of its group's tensor),
    or None with want_witness=False. layout(len(messages[i])) / matrices(len(messages[i])) describe instance i's system."""
    torch = _require_cuda()
    assert isinstance(parameters, ParametersVar)
    pk, sig = public_key.xy, signature.xy
    n = pk.shape[0]
    if len(messages) != n or sig.shape[0] != n:
        raise BlswError("verify_mixed_lengths: one message and one signature per key")
    dev = pk.device
    groups = {}
    for i, m in enumerate(messages):
        b = bytes(m.cpu().numpy().tobytes()) if hasattr(m, "cpu") else bytes(m)
        groups.setdefault(len(b), []).append((i, b))
    result = torch.empty(n, dtype=torch.int32, device=dev)
    witnesses = [None] * n
    for msg_len, items in sorted(groups.items()):
        idx = torch.tensor([i for i, _ in items], dtype=torch.long, device=dev)
        msg = torch.frombuffer(bytearray(b"".join(b for _, b in items)), dtype=torch.uint8).reshape(len(items), msg_len).to(dev) if msg_len else \
            torch.empty((len(items), 0), dtype=torch.uint8, device=dev)
        g = BlsSignatureVerifyGadget(len(items), msg_len, device=dev, want_witness=want_witness, **options)
        res = g.verify(parameters, PublicKeyVar.new_witness(pk[idx].contiguous()), msg, SignatureVar.new_witness(sig[idx].contiguous()))
        torch.cuda.synchronize(dev)
        result[idx] = res
        if want_witness:
            for k, (i, _) in enumerate(items):
                witnesses[i] = g.witness[k]
        g.engine.close()
    return result, (witnesses if want_witness else None)


ST_OK, ST_BAD_ENCODING, ST_NOT_ON_CURVE, ST_NOT_IN_SUBGROUP, ST_IDENTITY = 0, 1, 2, 3, 4


def decode_batch(pk48, sig96):
    """PublicKey::try_from / Signature::try_from for a batch (bls.rs:219-242, 316-339): uint8 cuda tensors [n,48], [n,96] ->
    (pk_xy [n,12] int64, sig_xy [n,24] int64, status [n,2] int32)."""
    torch = _require_cuda()
    n = pk48.shape[0]
    assert pk48.shape == (n, 48) and sig96.shape == (n, 96) and pk48.is_contiguous() and sig96.is_contiguous()
    pk_xy = torch.empty((n, 12), dtype=torch.int64, device=pk48.device)
    sig_xy = torch.empty((n, 24), dtype=torch.int64, device=pk48.device)
    status = torch.empty((n, 2), dtype=torch.int32, device=pk48.device)
    rc = lib().blsw_decode_batch(pk48.data_ptr(), sig96.data_ptr(), n, pk_xy.data_ptr(), sig_xy.data_ptr(), status.data_ptr(),
                                 torch.cuda.current_stream(pk48.device).cuda_stream)
    if rc:
        raise BlswError("blsw_decode_batch failed: %d" % rc)
    return pk_xy, sig_xy, status


def aggregate_points(group, points):
    """Signature::aggregate (group 2: [n, k, 96] uint8) / PublicKey::aggregate (group 1: [n, k, 48]) for n lists of k compressed points
    (bls.rs:288-300, 183-195; tests/tests.rs:270-294): returns (sum [n, 96 or 48] uint8, status [n] int32 — ST_OK or the status of the first
    point of the list that does not decode). An empty list (k == 0) is None, as in the reference."""
    torch = _require_cuda()
    nbytes = {1: 48, 2: 96}[group]
    n, k = points.shape[0], points.shape[1]
    assert points.shape == (n, k, nbytes) and points.dtype == torch.uint8 and points.is_contiguous()
    if k == 0:
        return None
    wb = ctypes.c_uint64(0)
    rc = lib().blsw_aggregate_points_workspace_bytes(group, n, k, ctypes.byref(wb))
    if rc:
        raise BlswError("blsw_aggregate_points_workspace_bytes failed: %d" % rc)
    ws = torch.empty(wb.value, dtype=torch.uint8, device=points.device)
    out = torch.empty((n, nbytes), dtype=torch.uint8, device=points.device)
    status = torch.empty(n, dtype=torch.int32, device=points.device)
    rc = lib().blsw_aggregate_points_batch(group, points.data_ptr(), k, n, out.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(),
                                           torch.cuda.current_stream(points.device).cuda_stream)
    if rc:
        raise BlswError("blsw_aggregate_points_batch failed: %d" % rc)
    torch.cuda.synchronize(points.device)
    return out, status


def aggregate_signatures(sig96):
    return aggregate_points(2, sig96)


def aggregate_public_keys(pk48):
    return aggregate_points(1, pk48)


_VERIFY_WS = {}


def verify_batch(pk48, msg, sig96, want_status=False):
    """BLS::verify (bls.rs:427-458) for a batch as VALUES (blsw_verify_batch: the native algorithm — decode with subgroup checks, hash to G2, a two-pair
    Miller loop over projective lines, final exponentiation — no circuit): uint8 cuda tensors pk48 [n, 48], msg [n, msg_len], sig96 [n, 96] ->
    int32 [n] verdicts (1 / 0; every Err of the reference's verify counts as false), optionally with the decode statuses [n, 2]."""
    torch = _require_cuda()
    n, msg_len = pk48.shape[0], msg.shape[1]
    assert pk48.shape == (n, 48) and sig96.shape == (n, 96) and msg.shape[0] == n and pk48.is_contiguous() and sig96.is_contiguous() and msg.is_contiguous()
    dev = pk48.device
    wb = ctypes.c_uint64(0)
    rc = lib().blsw_verify_workspace_bytes(n, msg_len, ctypes.byref(wb))
    if rc:
        raise BlswError("blsw_verify_workspace_bytes failed: %d" % rc)
    key = (str(dev), wb.value)
    ws = _VERIFY_WS.get(key)
    if ws is None:  # one workspace per (device, size): repeated calls of one batch shape (a verifier's loop) do not reallocate
        _VERIFY_WS.clear()
        ws = _VERIFY_WS[key] = torch.empty(wb.value, dtype=torch.uint8, device=dev)
    res = torch.empty(n, dtype=torch.int32, device=dev)
    status = torch.empty((n, 2), dtype=torch.int32, device=dev)
    rc = lib().blsw_verify_batch(pk48.data_ptr(), sig96.data_ptr(), msg.data_ptr() if msg_len else None, msg_len, n, res.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(),
                                 torch.cuda.current_stream(dev).cuda_stream)
    if rc:
        raise BlswError("blsw_verify_batch failed: %d" % rc)
    return (res, status) if want_status else res


def fast_aggregate_verify_batch(pks48, msg, sig96):
    """tests/tests.rs:296-334: n instances of k compressed keys each signing ONE message: PublicKey::aggregate (blsw_aggregate_points_batch), then
    BLS::verify on the aggregate (blsw_verify_batch). A key that does not decode, an empty list or an identity aggregate is false. -> int32 [n]"""
    torch = _require_cuda()
    n = pks48.shape[0]
    if pks48.shape[1] == 0:
        return torch.zeros(n, dtype=torch.int32, device=pks48.device)
    agg, st = aggregate_points(1, pks48)
    res = verify_batch(agg, msg, sig96)
    return torch.where(st == 0, res, torch.zeros_like(res))


def verify_bytes_batch(pk48, msg, sig96):
    """tests/tests.rs:239-268 semantics on the GPU in one ABI call (blsw_engine_submit_bytes): decode, run the gadget, accept iff
    both points decode to non-identity subgroup points and the in-circuit result is true. Returns a bool tensor [n]."""
    torch = _require_cuda()
    eng = WitnessEngine(pk48.shape[0], msg.shape[1], device=pk48.device)
    res = torch.empty(pk48.shape[0], dtype=torch.int32, device=pk48.device)
    eng.submit_bytes(pk48.contiguous(), sig96.contiguous(), msg.contiguous(), witness=None, result=res)
    eng.flush()
    torch.cuda.synchronize(pk48.device)
    eng.close()
    return res == 1


def layout_aggregate(msg_len, n_keys):
    L = blsw_layout_t()
    rc = lib().blsw_layout_aggregate(msg_len, n_keys, ctypes.byref(L))
    if rc:
        raise BlswError("blsw_layout_aggregate failed: %d" % rc)
    return {n: getattr(L, n) for n in _LAYOUT_FIELDS}


def aggregate_verify(parameters, public_keys, bitmap, message, signature, want_witness=True):
    """BlsSignatureVerifyGadget::aggregate_verify (constraints.rs:153-167) for n instances: public_keys.xy [n, K, 12] int64,
    bitmap [n, K] uint8 (0/1), message [n, msg_len] uint8, signature.xy [n, 24]. Returns (result int32 [n], count int32 [n], witness)."""
    torch = _require_cuda()
    assert isinstance(parameters, ParametersVar)
    pks, sig, bitmap, message = public_keys.xy.contiguous(), signature.xy.contiguous(), bitmap.contiguous(), message.contiguous()
    n, K = pks.shape[0], pks.shape[1]
    assert K >= 1 and bitmap.shape == (n, K)  # constraints.rs:160-162: equal lengths, at least one key
    msg_len = message.shape[1]
    lay = layout_aggregate(msg_len, K)
    wb = ctypes.c_uint64(0)
    lib().blsw_aggregate_workspace_bytes(n, msg_len, K, ctypes.byref(wb))
    dev = pks.device
    check_capacity(dev, "aggregate_verify (n = %d, %d keys)" % (n, K), wb.value, n * lay["n_witness"] * 48 if want_witness else 0)
    ws = torch.empty(wb.value, dtype=torch.uint8, device=dev)
    res = torch.empty(n, dtype=torch.int32, device=dev)
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    wit = torch.empty((n, lay["n_witness"], 6), dtype=torch.int64, device=dev) if want_witness else None
    assert sig.shape == (n, 24) and message.shape[0] == n
    rc = lib().blsw_aggregate_verify_batch(pks.data_ptr(), bitmap.data_ptr(), K, sig.data_ptr(), message.data_ptr(), msg_len, n,
                                           wit.data_ptr() if wit is not None else None, lay["n_witness"], res.data_ptr(), cnt.data_ptr(), ws.data_ptr(),
                                           ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    if rc:
        raise BlswError("blsw_aggregate_verify_batch failed: %d" % rc)
    torch.cuda.synchronize(dev)
    return res, cnt, wit


def matrices(msg_len=32, n_keys=0, n_pairs=1, params_mode=0, pk_mode=0, sig_mode=0):
    """Constraint matrices of a circuit shape (host only; blsw_matrices_info + blsw_matrices_fill): the R1CS an arkworks prover
    takes next to the witness vectors, in ConstraintMatrices shape. Returns dict(n_constraints, n_instance_vars, n_witness,
    A / B / C = (row_ptr uint64 [n_constraints + 1], col uint32 [nnz], val uint64 [nnz, 6] Montgomery limbs)).
    params_mode 1 / "witness" (single-key circuit): the system of layout(msg_len, params_mode=1)."""
    import numpy as np

    params_mode = PARAMS_MODES.get(params_mode, params_mode)
    pk_mode, sig_mode = IO_MODES.get(pk_mode, pk_mode), IO_MODES.get(sig_mode, sig_mode)
    info = blsw_matrices_info_t()
    if pk_mode or sig_mode:  # columns: 0 = one, 1 .. n_instance_vars - 1 = the public inputs, n_instance_vars + k = witness k
        if n_keys or n_pairs != 1 or params_mode:
            raise BlswError("pk_mode / sig_mode apply to the single-key circuit with Constant parameters")
        rc = lib().blsw_matrices_info_io(msg_len, pk_mode, sig_mode, ctypes.byref(info))
    elif params_mode:
        if n_keys or n_pairs != 1:
            raise BlswError("params_mode applies to the single-key circuit")
        rc = lib().blsw_matrices_info_params(msg_len, params_mode, ctypes.byref(info))
    else:
        rc = lib().blsw_matrices_info(msg_len, n_keys, n_pairs, ctypes.byref(info))
    if rc:
        raise BlswError("blsw_matrices_info failed: %d" % rc)
    u64p, u32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
    rp = [np.zeros(info.n_constraints + 1, dtype=np.uint64) for _ in range(3)]
    col = [np.zeros(info.nnz[m], dtype=np.uint32) for m in range(3)]
    val = [np.zeros((info.nnz[m], 6), dtype=np.uint64) for m in range(3)]
    out = blsw_matrices_t()
    for m in range(3):
        out.row_ptr[m] = rp[m].ctypes.data_as(u64p)
        out.col[m] = col[m].ctypes.data_as(u32p)
        out.val[m] = val[m].ctypes.data_as(u64p)
    if pk_mode or sig_mode:
        rc = lib().blsw_matrices_fill_io(msg_len, pk_mode, sig_mode, ctypes.byref(info), ctypes.byref(out))
    elif params_mode:
        rc = lib().blsw_matrices_fill_params(msg_len, params_mode, ctypes.byref(info), ctypes.byref(out))
    else:
        rc = lib().blsw_matrices_fill(msg_len, n_keys, n_pairs, ctypes.byref(info), ctypes.byref(out))
    if rc:
        raise BlswError("blsw_matrices_fill failed: %d" % rc)
    return {"n_constraints": info.n_constraints, "n_instance_vars": info.n_instance_vars, "n_witness": info.n_witness,
            "A": (rp[0], col[0], val[0]), "B": (rp[1], col[1], val[1]), "C": (rp[2], col[2], val[2])}


def layout_multi(msg_len, n_pairs):
    L = blsw_layout_t()
    rc = lib().blsw_layout_multi(msg_len, n_pairs, ctypes.byref(L))
    if rc:
        raise BlswError("blsw_layout_multi failed: %d" % rc)
    return {n: getattr(L, n) for n in _LAYOUT_FIELDS}


def verify_multi(parameters, public_keys, messages, signature, want_witness=True):
    """N+1-pair product of pairings: one signature over K (pk_j, msg_j) pairs per instance, i.e. constraints.rs:90-128 with
    product_of_pairings over slices of K + 1 prepared points. public_keys.xy [n, K, 12] int64, messages [n, K, msg_len] uint8,
    signature.xy [n, 24]. Returns (result int32 [n], witness [n, n_witness, 6] int64 or None)."""
    torch = _require_cuda()
    assert isinstance(parameters, ParametersVar)
    pks, sig, messages = public_keys.xy.contiguous(), signature.xy.contiguous(), messages.contiguous()
    n, K = pks.shape[0], pks.shape[1]
    assert K >= 1 and messages.shape[:2] == (n, K) and sig.shape == (n, 24) and pks.shape == (n, K, 12)
    msg_len = messages.shape[2]
    lay = layout_multi(msg_len, K)
    wb = ctypes.c_uint64(0)
    rc = lib().blsw_verify_multi_workspace_bytes(n, msg_len, K, ctypes.byref(wb))
    if rc:
        raise BlswError("blsw_verify_multi_workspace_bytes failed: %d" % rc)
    dev = pks.device
    check_capacity(dev, "verify_multi (n = %d, %d pairs)" % (n, K), wb.value, n * lay["n_witness"] * 48 if want_witness else 0)
    ws = torch.empty(wb.value, dtype=torch.uint8, device=dev)
    res = torch.empty(n, dtype=torch.int32, device=dev)
    wit = torch.empty((n, lay["n_witness"], 6), dtype=torch.int64, device=dev) if want_witness else None
    rc = lib().blsw_verify_multi_batch(pks.data_ptr(), messages.data_ptr() if msg_len else None, msg_len, K, sig.data_ptr(), n, wit.data_ptr() if wit is not None else None,
                                       lay["n_witness"], res.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    if rc:
        raise BlswError("blsw_verify_multi_batch failed: %d" % rc)
    torch.cuda.synchronize(dev)
    return res, wit


DIGEST_KEY, DIGEST_A = 0x9E3779B1, 0x85EBCA6B  # include/blsw.h: blsw_witness_digest


def witness_digest(witness, n_witness=None, out=None, stream=None):
    """blsw_witness_digest: [n, stride, 6] int64 cuda tensor -> [n, 2] int64 (two u64 sums, see include/blsw.h)."""
    torch = _require_cuda()
    assert witness.is_cuda and witness.is_contiguous() and witness.dim() == 3 and witness.shape[2] == 6
    n, stride = witness.shape[0], witness.shape[1]
    if out is None:
        out = torch.empty((n, 2), dtype=torch.int64, device=witness.device)
    s = stream if stream is not None else torch.cuda.current_stream(witness.device)
    rc = lib().blsw_witness_digest(witness.data_ptr(), stride, n, n_witness if n_witness is not None else stride, out.data_ptr(), s.cuda_stream)
    if rc:
        raise BlswError("blsw_witness_digest failed: %d" % rc)
    return out


def witness_digest_reference(words):
    """The same digest in numpy (host-side definition used by consumers / tests; include/blsw.h): `words` = the instance's u64 words."""
    import numpy as np

    x = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1).view(np.uint32).reshape(-1, 4)  # 16-byte pieces of little-endian u32 words
    key = ((np.arange(1, x.shape[0] + 1, dtype=np.uint64) * np.uint64(DIGEST_KEY)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    a = np.uint32(DIGEST_A)
    with np.errstate(over="ignore"):
        t = [(x[:, 0] + key), (x[:, 1] + key + a), (x[:, 2] + key + np.uint32(2) * a), (x[:, 3] + key + np.uint32(3) * a)]
        prod = t[0].astype(np.uint64) * t[1].astype(np.uint64) + t[2].astype(np.uint64) * t[3].astype(np.uint64)
        d0 = int(prod.sum(dtype=np.uint64))
        lo = int(((x[:, 0] ^ key) + (x[:, 2] ^ ~key)).sum(dtype=np.uint32))
        hi = int(((x[:, 1] ^ key) + (x[:, 3] ^ ~key)).sum(dtype=np.uint32))
    return [d0, lo | (hi << 32)]


def microbench(which, iters=4096, blocks=4096):
    """Measured device rates for the VALU roofline: which=0 v_mad_u64_u32/s, which=1 Fp products/s."""
    _require_cuda()
    v = ctypes.c_double(0)
    rc = lib().blsw_microbench(which, iters, blocks, ctypes.byref(v))
    if rc:
        raise BlswError("blsw_microbench failed: %d" % rc)
    return v.value


def fill_rate(tensor, reps=2):
    """blsw_fill_rate: bytes/s of a plain fill of `tensor` (a cuda tensor; overwritten) in the expansion's store geometry — the same-box HBM yardstick."""
    _require_cuda()
    v = ctypes.c_double(0)
    rc = lib().blsw_fill_rate(ctypes.c_void_p(tensor.data_ptr()), ctypes.c_uint64(tensor.numel() * tensor.element_size()), reps, ctypes.byref(v))
    if rc:
        raise BlswError("blsw_fill_rate failed: %d" % rc)
    return v.value


def hash_to_g2_batch(message, out=None):
    """Batched hash_to_g2_with_cons values (hasher.rs:727-740): message [n, msg_len] uint8 cuda tensor -> [n, 24] int64 affine."""
    torch = _require_cuda()
    n, msg_len = message.shape
    wb = ctypes.c_uint64(0)
    lib().blsw_hash_to_g2_workspace_bytes(n, msg_len, ctypes.byref(wb))
    ws = torch.empty(wb.value, dtype=torch.uint8, device=message.device)
    if out is None:
        out = torch.empty((n, 24), dtype=torch.int64, device=message.device)
    rc = lib().blsw_hash_to_g2_batch(message.data_ptr(), msg_len, n, out.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream(message.device).cuda_stream)
    if rc:
        raise BlswError("blsw_hash_to_g2_batch failed: %d" % rc)
    return out


ST_INVALID_SECRET_KEY = 5


def sign_batch(sk32_le, message, want_bytes=True):
    """BLS::sign + PublicKey::from(&sk) for a batch (bls.rs:411-425, 183-195): sk32_le [n, 32] uint8 (little-endian Fr, as
    PrivateKey::try_from takes it), message [n, msg_len] uint8, cuda tensors.
    Returns dict(sig96, pk48 (uint8, None unless want_bytes), sig_xy [n,24], pk_xy [n,12] int64 Montgomery, status [n] int32)."""
    torch = _require_cuda()
    n, msg_len = message.shape
    assert sk32_le.shape == (n, 32) and sk32_le.is_contiguous() and message.is_contiguous()
    dev = message.device
    wb = ctypes.c_uint64(0)
    lib().blsw_hash_to_g2_workspace_bytes(n, msg_len, ctypes.byref(wb))
    ws = torch.empty(wb.value, dtype=torch.uint8, device=dev)
    sig96 = torch.empty((n, 96), dtype=torch.uint8, device=dev) if want_bytes else None
    pk48 = torch.empty((n, 48), dtype=torch.uint8, device=dev) if want_bytes else None
    sig_xy = torch.empty((n, 24), dtype=torch.int64, device=dev)
    pk_xy = torch.empty((n, 12), dtype=torch.int64, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    rc = lib().blsw_sign_batch(sk32_le.data_ptr(), message.data_ptr(), msg_len, n, sig96.data_ptr() if want_bytes else None, sig_xy.data_ptr(),
                               pk48.data_ptr() if want_bytes else None, pk_xy.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(),
                               torch.cuda.current_stream(dev).cuda_stream)
    if rc:
        raise BlswError("blsw_sign_batch failed: %d" % rc)
    torch.cuda.synchronize(dev)
    return {"sig96": sig96, "pk48": pk48, "sig_xy": sig_xy, "pk_xy": pk_xy, "status": status}
