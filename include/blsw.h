/* blsw — MI355X-native batched witness generation for the BLS12-381 signature-verify R1CS gadget.
 *
 * C ABI of libblsw.so (HIP, gfx950). The reference (lightec-xyz/bls-verify-gadget) has NO FFI: the path sits
 * behind the arkworks trait surface
 *     impl SigVerifyGadget<BLS<P>, P::Fp> for BlsSignatureVerifyGadget<P>::verify     src/constraints.rs:79-128
 *     AllocVar for ParametersVar / PublicKeyVar / SignatureVar                         src/constraints.rs:194-249
 *     hash_to_g2_with_cons(cs, &[UInt8]) -> G2Var                                      src/hasher.rs:727-740
 * and the data an arkworks prover consumes from it is ConstraintSystem::witness_assignment (Vec<Fq>, Montgomery
 * form, allocation order). These entry points replace exactly that side effect, batched over instances; they are
 * what a Rust `extern "C"` shim binds.
 *
 * Conventions: caller-allocated buffers, integer return codes (0 = ok), no exceptions across the ABI,
 * re-entrant per (device, stream). All `d_*` pointers are DEVICE pointers on the current HIP device.
 * A field element is 6 little-endian u64 limbs in Montgomery form (R = 2^384) == arkworks' in-memory Fq.
 */
#ifndef BLSW_H
#define BLSW_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BLSW_OK 0
#define BLSW_ERR_ARG 1
#define BLSW_ERR_WORKSPACE 2
#define BLSW_ERR_HIP 3
#define BLSW_ERR_NO_DEVICE 4

/* per-instance input status (mirrors src/bls.rs:434-447 and the deserialization fixtures) */
#define BLSW_ST_OK 0
#define BLSW_ST_BAD_ENCODING 1
#define BLSW_ST_NOT_ON_CURVE 2
#define BLSW_ST_NOT_IN_SUBGROUP 3
#define BLSW_ST_IDENTITY 4

/* Segment table of one instance's witness vector for the circuit of src/constraints.rs:335-366
 * (msg witness bytes, params Constant, pk Witness, sig Witness, then verify). Offsets are in field elements. */
typedef struct {
    uint32_t msg_len;
    uint32_t n_instance_vars; /* 1: the constant one (this gadget allocates no public input) */
    uint32_t n_witness;       /* witness_assignment length */
    uint32_t sha_bits;        /* boolean witnesses of the expand_message segment (16 lib_str bits + SHA-256 gadget) */
    uint32_t off_msg, off_pk_alloc, off_sig_alloc, off_pk_not_zero, off_expand, off_map0, off_map1, off_add, off_cofactor, off_prep_h, off_prep_pk,
        off_prep_sig, off_miller, off_final_exp, off_is_one;
    /* aggregate_verify circuits (src/constraints.rs:378-441); all zero for the single-key circuit (then off_pk_alloc is used) */
    uint32_t n_keys, off_keys, off_bitmap, off_count, off_agg;
} blsw_layout_t;

/* layout(circuit shape) — replaces reading cs.num_witness_variables() after synthesis (constraints.rs:369-373). Host only. */
int blsw_layout(uint32_t msg_len, blsw_layout_t* out);

/* layout of the aggregate_verify circuit with n_keys public keys (keys Witness, bitmap booleans Witness, msg, sig, then
 * mapped_aggregate + verify: src/constraints.rs:153-191, 378-441). n_keys == 0 gives blsw_layout. Host only. */
int blsw_layout_aggregate(uint32_t msg_len, uint32_t n_keys, blsw_layout_t* out);
int blsw_aggregate_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t n_keys, uint64_t* bytes);
/* BlsSignatureVerifyGadget::aggregate_verify for n independent instances of n_keys keys each (same message per instance):
 *   d_pks_xy [n][n_keys][12] u64, d_bitmap [n][n_keys] bytes (0/1: Boolean::new_witness), d_sig_xy [n][24], d_msg [n][msg_len]
 *   d_witness [n][witness_stride] (may be NULL), d_result [n] int32 (the Boolean), d_count [n] uint32 (the UInt32 count)
 * Direct mode, asynchronous on `stream`. */
int blsw_aggregate_verify_batch(const uint64_t* d_pks_xy, const uint8_t* d_bitmap, uint32_t n_keys, const uint64_t* d_sig_xy, const uint8_t* d_msg,
                                uint32_t msg_len, uint64_t n, uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, uint32_t* d_count,
                                void* d_workspace, uint64_t workspace_bytes, void* stream);

/* Execution engine. Batches of n instances are SUBMITTED with their input / output pointers and processed in groups
 * of up to max_steps batches by one set of kernel launches (one batch of 1024 instances is only 16 wavefronts per
 * chain; a group of 32 batches fills the 1024 SIMDs of an MI355X). max_steps == 1 is the direct mode: the chains
 * write every witness in place. With max_steps > 1 field witnesses are staged element-major (coalesced stores) and
 * each batch's witness tensor is then written, in submission order, by the streaming placement kernels.
 * n_buffers groups can be in flight at once (each with its own streams and workspace slice). Keep
 * max_steps * n / 64 wavefronts * the largest per-lane stack (11 KB) under the runtime's 140 MB per-dispatch scratch
 * limit (max_steps <= 8 for n = 1024): larger dispatches fall into ROCr's allocate-per-dispatch scratch path.
 * The caller owns the device workspace (blsw_engine_workspace_bytes). One engine per device; not thread-safe. */
typedef struct blsw_engine blsw_engine_t;
int blsw_engine_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, uint64_t* bytes);
int blsw_engine_create(blsw_engine_t** out, uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, void* d_workspace,
                       uint64_t workspace_bytes);
int blsw_engine_destroy(blsw_engine_t* e);

/* Submits one batch of n independent (pk, msg, sig) instances: the witness vectors of the circuit of
 * src/constraints.rs:335-366 are written to d_witness and the gadget's output Boolean (constraints.rs:127) to d_result.
 *   d_pk_xy   [n][12] u64  affine G1 (x, y) Montgomery; (0,0) = point at infinity      (PublicKeyVar, constraints.rs:214-232)
 *   d_sig_xy  [n][24] u64  affine G2 (x.c0, x.c1, y.c0, y.c1); all zero = infinity     (SignatureVar, constraints.rs:234-249)
 *   d_msg     [n][msg_len] bytes                                                        (UInt8::new_witness_vec, constraints.rs:341)
 *   d_witness [n][witness_stride] field elements (48 B each), witness_stride >= layout.n_witness; may be NULL (results only)
 *   d_result  [n] int32, may be NULL
 * Device work is issued when max_steps batches are pending or at blsw_engine_flush; `stream` (hipStream_t, may be NULL)
 * is the stream on which the inputs become valid. Buffers must stay alive until the flush has completed. */
int blsw_engine_submit(blsw_engine_t* e, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, uint64_t* d_witness,
                       uint64_t witness_stride, int32_t* d_result, void* stream);
/* Issues everything pending and makes `stream` wait for all batches submitted so far (asynchronous for the host). */
int blsw_engine_flush(blsw_engine_t* e, void* stream);
/* average duration (ms) of the bit->Fp expansion kernel launches issued since the previous call (HIP events on the stream they
 * ran on, at most 1024 launches); blocks until they have finished and resets the statistics */
int blsw_engine_expand_stats(blsw_engine_t* e, uint32_t* count, float* avg_ms);

/* Input decode (PublicKey::try_from / Signature::try_from -> deserialize_compressed, src/bls.rs:219-242, 316-339):
 *   d_pk48 [n][48], d_sig96 [n][96]  ZCash-format compressed points
 *   d_pk_xy [n][12], d_sig_xy [n][24] affine Montgomery coordinates in the layout blsw_engine_submit takes (identity / failure = zeros)
 *   d_status [n][2] int32: BLSW_ST_* of the key and of the signature (flags, x < p, on curve, prime-order subgroup;
 *   BLSW_ST_IDENTITY = well-formed encoding of the point at infinity)
 * tests/tests.rs:244-263 semantics: an instance verifies iff both statuses are BLSW_ST_OK and the gadget result is 1. */
int blsw_decode_batch(const uint8_t* d_pk48, const uint8_t* d_sig96, uint64_t n, uint64_t* d_pk_xy, uint64_t* d_sig_xy, int32_t* d_status, void* stream);

/* hash_to_g2 only (src/hasher.rs:727-740 / src/bls.rs:477-493): d_out_affine [n][24] u64 (x.c0, x.c1, y.c0, y.c1) Montgomery */
int blsw_hash_to_g2_workspace_bytes(uint64_t n, uint32_t msg_len, uint64_t* bytes);
int blsw_hash_to_g2_batch(const uint8_t* d_msg, uint32_t msg_len, uint64_t n, uint64_t* d_out_affine, void* d_workspace, uint64_t workspace_bytes,
                          void* stream);

/* Native signer for a batch: BLS::sign (src/bls.rs:411-425: sig = sk * H(msg); Err(InvalidSecretKey) for sk = 0) and
 * PublicKey::from(&sk) (src/bls.rs:183-195: pk = sk * g1). One key per instance.
 *   d_sk32_le [n][32]  secret keys as PrivateKey::try_from(&[u8]) takes them (src/bls.rs:97-103): little-endian Fr
 *   outputs, each nullable except d_status: d_sig96 [n][96] / d_pk48 [n][48] compressed points as Signature / PublicKey
 *   serialise (src/bls.rs:244-260, 341-357), d_sig_xy [n][24] / d_pk_xy [n][12] affine Montgomery limbs (engine input)
 *   d_status [n] int32: BLSW_ST_OK, BLSW_ST_BAD_ENCODING (sk >= r) or BLSW_ST_INVALID_SECRET_KEY (sk = 0); on error
 *   the outputs of that instance are the identity encoding / zeros.
 * Workspace: blsw_hash_to_g2_workspace_bytes(n, msg_len). */
#define BLSW_ST_INVALID_SECRET_KEY 5
int blsw_sign_batch(const uint8_t* d_sk32_le, const uint8_t* d_msg, uint32_t msg_len, uint64_t n, uint8_t* d_sig96, uint64_t* d_sig_xy, uint8_t* d_pk48,
                    uint64_t* d_pk_xy, int32_t* d_status, void* d_workspace, uint64_t workspace_bytes, void* stream);

/* Device micro-benchmarks that give the VALU roofline its MEASURED denominator:
 * which = 0: v_mad_u64_u32 rate (32x32+64 multiply-adds per second, all CUs); which = 1: Fp Montgomery products per second. */
int blsw_microbench(int which, uint32_t iters, uint32_t blocks, double* ops_per_s);

int blsw_version(void);

#ifdef __cplusplus
}
#endif
#endif
