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
LIB_PATH = os.path.join(HERE, "libblsw.so")

FP_BYTES = 48
_LAYOUT_FIELDS = (
    "msg_len n_instance_vars n_witness sha_bits off_msg off_pk_alloc off_sig_alloc off_pk_not_zero off_expand off_map0 off_map1 "
    "off_add off_cofactor off_prep_h off_prep_pk off_prep_sig off_miller off_final_exp off_is_one"
).split()


class blsw_layout_t(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in _LAYOUT_FIELDS]


class BlswError(RuntimeError):
    pass


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
        L.blsw_layout.argtypes = [ctypes.c_uint32, ctypes.POINTER(blsw_layout_t)]
        L.blsw_workspace_bytes.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
        L.blsw_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        L.blsw_ctx_destroy.argtypes = [ctypes.c_void_p]
        L.blsw_ctx_last_expand_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        L.blsw_microbench.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]
        L.blsw_witness_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        L.blsw_hash_to_g2_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = ["blsw_version", "blsw_layout", "blsw_workspace_bytes", "blsw_ctx_create", "blsw_ctx_destroy", "blsw_ctx_last_expand_ms",
                    "blsw_witness_batch", "blsw_hash_to_g2_batch", "blsw_microbench"]


def layout(msg_len=32):
    """Segment table of the witness vector (host logic; replaces cs.num_witness_variables(), constraints.rs:369-373)."""
    L = blsw_layout_t()
    rc = lib().blsw_layout(msg_len, ctypes.byref(L))
    if rc:
        raise BlswError("blsw_layout failed: %d" % rc)
    return {n: getattr(L, n) for n in _LAYOUT_FIELDS}


def workspace_bytes(n, msg_len=32):
    b = ctypes.c_uint64(0)
    rc = lib().blsw_workspace_bytes(n, msg_len, ctypes.byref(b))
    if rc:
        raise BlswError("blsw_workspace_bytes failed: %d" % rc)
    return b.value


def _require_cuda():
    import torch

    if not torch.cuda.is_available():
        raise BlswError("no HIP device visible: the witness path runs only on the GPU (there is no CPU fallback)")
    return torch


class ParametersVar:
    """constraints.rs:23-28: g1_generator; only AllocationMode::Constant (the default generator) is on the GPU path."""

    def __init__(self):
        self.mode = "Constant"


class PublicKeyVar:
    """constraints.rs:39-44, AllocVar at :214-232 (Witness mode). `xy`: [n, 12] int64 tensor (u64 limbs: x, y Montgomery)."""

    def __init__(self, xy):
        self.xy = xy

    @classmethod
    def new_witness(cls, xy):
        return cls(xy)


class SignatureVar:
    """constraints.rs:55-60, AllocVar at :234-249 (Witness mode). `xy`: [n, 24] int64 tensor (x.c0, x.c1, y.c0, y.c1)."""

    def __init__(self, xy):
        self.xy = xy

    @classmethod
    def new_witness(cls, xy):
        return cls(xy)


class BlsSignatureVerifyGadget:
    """Batched counterpart of constraints.rs:79-128. One call = n independent circuits."""

    def __init__(self, n, msg_len=32, device=None, want_witness=True):
        torch = _require_cuda()
        self.torch = torch
        self.n = int(n)
        self.msg_len = int(msg_len)
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.layout = layout(msg_len)
        self.n_witness = self.layout["n_witness"]
        self.workspace = torch.empty(workspace_bytes(self.n, msg_len), dtype=torch.uint8, device=self.device)
        self.result = torch.empty(self.n, dtype=torch.int32, device=self.device)
        self.witness = torch.empty((self.n, self.n_witness, 6), dtype=torch.int64, device=self.device) if want_witness else None
        self._ctx = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = lib().blsw_ctx_create(ctypes.byref(self._ctx))
        if rc:
            raise BlswError("blsw_ctx_create failed: %d" % rc)

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                lib().blsw_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def last_expand_ms(self):
        ms = ctypes.c_float(0)
        rc = lib().blsw_ctx_last_expand_ms(self._ctx, ctypes.byref(ms))
        if rc:
            raise BlswError("blsw_ctx_last_expand_ms failed: %d" % rc)
        return ms.value

    def verify(self, parameters, public_key, message, signature, witness=None, stream=None):
        """message: [n, msg_len] uint8 tensor. Returns the int32 result tensor (gadget Boolean per instance); the witness
        vectors are in self.witness (or the tensor passed as `witness`)."""
        torch = self.torch
        assert isinstance(parameters, ParametersVar)
        pk, sig = public_key.xy, signature.xy
        assert pk.is_cuda and sig.is_cuda and message.is_cuda
        assert pk.shape == (self.n, 12) and sig.shape == (self.n, 24) and message.shape == (self.n, self.msg_len)
        assert pk.is_contiguous() and sig.is_contiguous() and message.is_contiguous()
        w = witness if witness is not None else self.witness
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        rc = lib().blsw_witness_batch(self._ctx, pk.data_ptr(), sig.data_ptr(), message.data_ptr(), self.msg_len, self.n, w.data_ptr() if w is not None else None,
                                      self.n_witness, self.result.data_ptr(), self.workspace.data_ptr(), self.workspace.numel(), s.cuda_stream)
        if rc:
            raise BlswError("blsw_witness_batch failed: %d" % rc)
        return self.result


def microbench(which, iters=4096, blocks=4096):
    """Measured device rates for the VALU roofline: which=0 v_mad_u64_u32/s, which=1 Fp products/s."""
    _require_cuda()
    v = ctypes.c_double(0)
    rc = lib().blsw_microbench(which, iters, blocks, ctypes.byref(v))
    if rc:
        raise BlswError("blsw_microbench failed: %d" % rc)
    return v.value


def hash_to_g2_batch(message, out=None):
    """Batched hash_to_g2_with_cons values (hasher.rs:727-740): message [n, msg_len] uint8 cuda tensor -> [n, 24] int64 affine."""
    torch = _require_cuda()
    n, msg_len = message.shape
    ws = torch.empty(workspace_bytes(n, msg_len), dtype=torch.uint8, device=message.device)
    if out is None:
        out = torch.empty((n, 24), dtype=torch.int64, device=message.device)
    rc = lib().blsw_hash_to_g2_batch(message.data_ptr(), msg_len, n, out.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream(message.device).cuda_stream)
    if rc:
        raise BlswError("blsw_hash_to_g2_batch failed: %d" % rc)
    return out
