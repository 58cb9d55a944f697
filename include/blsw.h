/* blsw — MI355X-native batched witness generation for the BLS12-381 signature-verify R1CS gadget.
 *
 * C ABI of libblsw.so (HIP, gfx950). The reference (lightec-xyz/bls-verify-gadget) has NO FFI: the path sits
 * behind the arkworks trait surface
 *     impl SigVerifyGadget<BLS<P>, P::Fp> for BlsSignatureVerifyGadget<P>::verify     src/constraints.rs:79-128
 *     AllocVar for ParametersVar / PublicKeyVar / SignatureVar                         src/constraints.rs:194-249
 *     hash_to_g2_with_cons(cs, &[UInt8]) -> G2Var                                      src/hasher.rs:727-740
 * and the data an arkworks prover consumes from it is ConstraintSystem::witness_assignment (Vec<Fq>, Montgomery
 * form, allocation order) plus the constraint matrices of cs.to_matrices(). These entry points replace exactly that
 * side effect, batched over instances; they are what a Rust `extern "C"` shim binds.
 * tests/test_abi_contract.py parses THIS file and fails when INTEGRATION.md's Rust block, the ctypes binding or the
 * library's exports drift from it.
 *
 * Conventions: caller-allocated buffers, integer return codes (0 = ok), no exceptions across the ABI,
 * re-entrant per (device, stream). All `d_*` pointers are DEVICE pointers on the engine's / current HIP device.
 * A field element is 6 little-endian u64 limbs in Montgomery form (R = 2^384) == arkworks' in-memory Fq.
 */
#ifndef BLSW_H
#define BLSW_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BLSW_ABI_VERSION 10

#define BLSW_OK 0
#define BLSW_ERR_ARG 1
#define BLSW_ERR_WORKSPACE 2
#define BLSW_ERR_HIP 3
#define BLSW_ERR_NO_DEVICE 4
#define BLSW_ERR_SCRATCH 5 /* the requested mode / n_buffers combination would exhaust the runtime's per-queue scratch */
#define BLSW_ERR_BUSY 6    /* consumer mode: the call has to wait for outputs the caller still holds (drain and release first) */

/* per-instance input status (mirrors src/bls.rs:434-447 and the deserialization fixtures) */
#define BLSW_ST_OK 0
#define BLSW_ST_BAD_ENCODING 1
#define BLSW_ST_NOT_ON_CURVE 2
#define BLSW_ST_NOT_IN_SUBGROUP 3
#define BLSW_ST_IDENTITY 4
#define BLSW_ST_INVALID_SECRET_KEY 5

/* Segment table of one instance's witness vector for the circuit of src/constraints.rs:335-366
 * (msg witness bytes, params Constant, pk Witness, sig Witness, then verify). Offsets are in field elements.
 * 36 uint32_t fields; the Rust mirror in INTEGRATION.md must have the same fields in the same order. */
typedef struct {
    uint32_t msg_len;
    uint32_t n_instance_vars; /* instance_assignment length: 1 (the constant one) + 3 if pk_mode is Input + 6 if sig_mode is Input */
    uint32_t n_witness;       /* witness_assignment length */
    uint32_t sha_bits;        /* boolean witnesses of ONE expand_message segment (16 lib_str bits + SHA-256 gadget) */
    uint32_t off_msg, off_pk_alloc, off_sig_alloc, off_pk_not_zero, off_expand, off_map0, off_map1, off_add, off_cofactor, off_prep_h, off_prep_pk,
        off_prep_sig, off_miller, off_final_exp, off_is_one;
    /* aggregate_verify circuits (src/constraints.rs:378-441); all zero for the single-key circuit (then off_pk_alloc is used) */
    uint32_t n_keys, off_keys, off_bitmap, off_count, off_agg;
    /* N+1-pair product (blsw_layout_multi): the segments msg, pk_alloc, pk_not_zero, {expand, map0, map1, add, cofactor},
     * prep_h and prep_pk exist once per (pk, msg) pair; pair j's copy starts at off_* + j * stride_*. n_pairs = 1 and
     * strides = segment lengths for the single-key and aggregate circuits. */
    uint32_t n_pairs, stride_msg, stride_pk_alloc, stride_pk_not_zero, stride_hash, stride_prep_h, stride_prep_pk;
    /* ParametersVar allocated with AllocationMode::Witness (blsw_layout_params; src/constraints.rs:198-211 takes any mode): the
     * generator's G1Var::new_variable segment between msg and pk_alloc, and prepare_g1(-g1) (to_affine, 7 witnesses) in front of
     * prep_h; the ell of the (-g1, sig) pair then has a variable point (68 x 8 witnesses more, 2 in the first one). All zero for
     * the reference's own circuits (params Constant: no witnesses). */
    uint32_t params_mode, off_params_alloc, off_prep_g1;
    /* PublicKeyVar / SignatureVar allocated with AllocationMode::Input (blsw_layout_io; src/constraints.rs:214-249 take any mode): 0 = Witness (the
     * reference's circuits), 1 = Input: the point's x, y, z are public inputs — instance_assignment = [1, pk.x, pk.y, pk.z, sig.x.c0, sig.x.c1,
     * sig.y.c0, sig.y.c1, sig.z.c0, sig.z.c1] in allocation order — and its allocation segment is empty (ark-r1cs-std 0.4.0 allocates Input points
     * through new_variable_omit_prime_order_check: no in-circuit subgroup check, a verifier checks its public inputs itself). Matrix columns:
     * 0 = one, 1 .. n_instance_vars - 1 = the inputs, n_instance_vars + k = witness k. */
    uint32_t pk_mode, sig_mode;
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

/* N+1-pair product of pairings: ONE signature over n_pairs (pk_j, msg_j) pairs,
 *     e(-g1, sig) * prod_j e(pk_j, H(msg_j)) == 1
 * i.e. BlsSignatureVerifyGadget::verify (src/constraints.rs:90-128) with every per-key statement turned into a loop over the
 * pairs and PairingVar::product_of_pairings called on slices of n_pairs + 1 prepared points (constraints.rs:121-125 passes
 * slices of 2). The reference has no such entry; n_pairs == 1 reproduces the single-key witness vector
 * bit for bit. Circuit allocation order: msg_0..msg_{K-1} witness bytes, params Constant, pk_0..pk_{K-1} Witness, sig Witness.
 *   d_pks_xy [n][n_pairs][12], d_msgs [n][n_pairs][msg_len], d_sig_xy [n][24]
 *   d_witness [n][witness_stride] (may be NULL), d_result [n] int32
 * Direct mode on the device that owns `stream`. The call copies its descriptor to the device and SYNCHRONISES `stream` once
 * before issuing the kernels (asynchronous from there on); side streams are kept per host thread and device. */
int blsw_layout_multi(uint32_t msg_len, uint32_t n_pairs, blsw_layout_t* out);
/* single-key circuit with ParametersVar::new_variable(.., mode) (src/constraints.rs:198-211): params_mode 0 = Constant (blsw_layout),
 * 1 = Witness. AllocationMode::Input would put the generator into instance_assignment, which this engine does not produce:
 * BLSW_ERR_ARG. Host only. */
int blsw_layout_params(uint32_t msg_len, uint32_t params_mode, blsw_layout_t* out);
/* single-key circuit with PublicKeyVar / SignatureVar::new_variable(.., mode) (src/constraints.rs:214-249): pk_mode / sig_mode 0 = Witness, 1 = Input
 * (blsw_layout_t.pk_mode). AllocationMode::Constant for a key or a signature is a different circuit in four segments and is not offered: BLSW_ERR_ARG.
 * Parameters Constant. Host only. */
int blsw_layout_io(uint32_t msg_len, uint32_t pk_mode, uint32_t sig_mode, blsw_layout_t* out);
int blsw_verify_multi_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t n_pairs, uint64_t* bytes);
int blsw_verify_multi_batch(const uint64_t* d_pks_xy, const uint8_t* d_msgs, uint32_t msg_len, uint32_t n_pairs, const uint64_t* d_sig_xy, uint64_t n,
                            uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, void* d_workspace, uint64_t workspace_bytes, void* stream);

/* Execution engine. Batches of n instances are SUBMITTED with their input / output pointers and processed in groups
 * of up to max_steps batches by one set of kernel launches (one batch of 1024 instances is only 16 wavefronts per
 * chain; a group of 16 batches fills a quarter of the 1024 SIMDs of an MI355X per chain kernel and several chains and
 * groups run side by side). max_steps == 1 with n_buffers == 1 is the direct mode: the chains write every witness in
 * place. Otherwise field witnesses are staged (coalesced stores) and each batch's witness tensor is written, in
 * submission order, by two streaming kernels on their own streams: the SHA-256 boolean segment (93 % of the bytes) as
 * soon as the batch's SHA witness bits exist — it does not wait for the curve / pairing chains — and the field segments
 * when the group's chains have finished. n_buffers groups can be in flight at once (each with its own streams and
 * workspace slice). The caller owns the device workspace (blsw_engine_workspace_bytes). Every entry point switches to
 * the engine's device for the duration of the call. One engine per device; not thread-safe.
 * Largest per-lane stack of the default kernels: 4.6 KB (the Fp12 inversion hint of the six-lane pairing kernel). */
typedef struct blsw_engine blsw_engine_t;
typedef struct {
    int32_t device;        /* HIP device ordinal, -1 = the current device */
    uint32_t n_keys;       /* 0 = the single-key circuit (blsw_engine_submit); K > 0 = the aggregate_verify circuit with K keys
                            * (blsw_layout_aggregate; batches go through blsw_engine_submit_aggregate) */
    uint32_t pairing_mode; /* 0 = six lanes per instance (default), 1 = one lane per instance (9.7 KB stack: A/B runs only) */
    uint32_t g2_mode;      /* 0 = one lane per instance (default), 1 = six lanes per instance (needs pairing_mode 0) */
    uint32_t expand_variant; /* store geometry of the SHA expansion kernel, low byte: 0 = 384 threads x 8 pieces, 1 = one 4 KiB-aligned
                              * chunk per 256-thread workgroup, 2 / 3 / 4 = 768 threads x 8 / 4 / 16 pieces in 4 KiB-aligned chunks, 5 = 384 x 16, 6 / 7 = one
                              * 8 / 16 KiB-aligned chunk per 512- / 1024-thread workgroup, 8 / 9 = 384 x 8 / 4 with the bit words in scalar registers
                              * (14 vector registers per lane), 10 / 11 / 12 = 384 x 8 / 16 / 32 with a light instruction stream (scalar bit window, scalar store base: 5 vector
                              * instructions per store instead of ~20), 13 = the blocks of 0 walked by a resident grid of 512 workgroups (a grid that is dispatched at once:
                              * the engine takes it by itself for the expansions it issues beside a latency group's chains); | 0x100 = raised wave priority.
                              * Only 0 is the shipped path. 10 / 11 / 12 (and expand_store 2 / 3) are EXPERIMENTAL, non-default measurement variants: their inline
                              * assembly carries hand-counted hazard wait states for gfx950 (the unit refuses to compile for another target) and only
                              * test_expansion_geometries_bit_exact guards them */
    uint32_t expand_store; /* stores of the SHA expansion kernel: 0 plain (default), 1 nontemporal, 2 sc1, 3 sc0 sc1 (variant 0) */
    uint32_t prio_mode;    /* stream priorities: 0 chains high, 1 placement high (default), 2 equal */
    uint32_t place_lds;    /* optional occupancy limiter of the expansion kernel: bytes of dynamic LDS per workgroup */
    uint32_t consumer_mode; /* 0 (default): free running — a step is written into its output as soon as stream order allows, the
                              caller guarantees that reusing an output is safe; 1: every output (witness tensor or compact buffer)
                              must be released with blsw_engine_output_consumed before its next use, and a step whose output is
                              still held is written later, when it is released (outputs can then be a ring much smaller than
                              the number of steps in flight: the chains run ahead into the staging, the 34 MB vectors exist only
                              between expansion and consumption) */
    uint32_t output_form;  /* witness elements in the output tensors: 0 (default) Montgomery form, 6 little-endian u64 limbs of
                              a * 2^384 mod p — reinterpretable as arkworks' in-memory Fq; 1: canonical integers, 48 bytes
                              little-endian — what CanonicalSerialize writes for an Fq (a prover in another process, a file).
                              The compact wire form is the same for both; blsw_engine_expand_compact follows the expanding
                              engine's option */
    uint32_t chain_variant; /* which compilation of the one-instance-per-lane chain kernels: 0 (default) = out of line in the grouped
                              engine (they leave registers to the streaming kernels that share their SIMDs), inlined in direct mode
                              (max_steps == 1 and n_buffers == 1: shorter under load, the whole register file) and for a small launch
                              group (at most 8 192 lanes) that starts a pipeline — the first two after creation / a flush — whose
                              latency is what a consumer waits for; 1 = out of line; 2 = inlined. Same witnesses either way. */
    uint32_t n_pairs;      /* 0 / 1 = the single-key circuit; K > 1 = the N+1-pair product (blsw_layout_multi): one signature over K (pk, msg)
                              pairs per instance, batches through blsw_engine_submit_multi. Staged engines only (max_steps > 1 or
                              n_buffers > 1), default kernel modes, n * K <= 65535. Compact wire form (blsw_engine_submit_multi_compact,
                              ~180 MB instead of 4.19 GB per instance at K = 128): n * K a multiple of 64 and n a divisor or a multiple of 64 */
    uint32_t cofactor_mode; /* clear_cofactor2 (the longest chain) with the three 255-bit chunks of its scalar on three lanes and a join: half the
                              chain's latency for 38 % more products in it. 0 (default) = for launch groups of at most 8 192 lanes (latency-bound),
                              one chain per lane above; 1 = never; 2 = always. Same witnesses either way. */
    uint32_t params_mode;  /* ParametersVar allocation (src/constraints.rs:198-211): 0 (default) Constant, as in every circuit of the reference;
                              1 Witness (blsw_layout_params): the generator is allocated like a public key, prepare_g1(-g1) and the ell of
                              the (-g1, sig) pair emit witnesses. Single-key circuit with the default kernel modes only (n_keys 0, n_pairs <= 1,
                              pairing_mode 0). */
    uint32_t group_ramp;   /* 0 (default): every launch group is max_steps batches (the last one of a flush: what is pending). 1: the first groups
                              after creation / a flush are 2, 4, 8, ... batches, up to max_steps: the first witness tensors exist after the chain
                              latency of a small group (its cofactor chain on three lanes), which is what a consumer-mode caller waits for
                              before it can consume anything. Same witnesses either way. */
    uint32_t latency_mode; /* the latency kernels for launch groups of at most 8 192 lanes: the hash-to-G2 / prepare / G2-allocation chains of an item on the
                              four lanes of a quad (the independent Fp products of every Fp2 operation on different lanes) and clear_cofactor2 "values
                              first" (the 636 doublings and 304 additions as Jacobian value chains with four inversions in all, every step's witnesses
                              derived by a lane of its own). 0 (default) = for such a group when it finds the engine's chains idle (it starts a pipeline:
                              its latency is what a consumer waits for); 1 = never; 2 = for every such group; 3 / 4 = as 2 with only the values-first
                              cofactor chain / only the quads (A/B runs). Same witnesses either way. */
    uint32_t pk_mode;      /* PublicKeyVar allocation (src/constraints.rs:214-232): 0 (default) Witness, 1 Input (blsw_layout_io): batches go through
                              blsw_engine_submit_io, which also writes instance_assignment. Single-key circuit with Constant parameters only. */
    uint32_t sig_mode;     /* SignatureVar allocation (src/constraints.rs:234-249): 0 (default) Witness, 1 Input; as pk_mode; not with g2_mode 1 */
} blsw_engine_options_t;
/* the defaults (pure: the library reads no environment variable; measurement scripts set the fields they want to vary) */
int blsw_engine_options_default(blsw_engine_options_t* out);
int blsw_engine_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, uint64_t* bytes);
int blsw_engine_workspace_bytes_ex(uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, const blsw_engine_options_t* options, uint64_t* bytes);
/* blsw_engine_create = blsw_engine_create_ex with blsw_engine_options_default. Returns BLSW_ERR_SCRATCH (nothing is
 * allocated) when pairing_mode 1 is combined with so many group buffers that the runtime's per-queue scratch
 * (stack bytes x 64 lanes x wave slots of the device, per queue) would exceed what ROCr can back: that combination
 * used to abort the process with HSA_STATUS_ERROR_OUT_OF_RESOURCES. */
/* BLSW_ERR_ARG also for: n > 65535 (one row of workgroups per instance in the expansion launches; no restriction in practice: a step's
 * output is n witness vectors of 34 MB, so 288 GB of HBM hold steps of at most ~8 000 instances — larger batches are more steps); options->consumer_mode > 1;
 * options->consumer_mode == 1 on a direct-mode engine (max_steps == 1 and n_buffers == 1: it writes witnesses in place while
 * the chains run and cannot hold a step back). */
int blsw_engine_create(blsw_engine_t** out, uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, void* d_workspace,
                       uint64_t workspace_bytes);
int blsw_engine_create_ex(blsw_engine_t** out, uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, const blsw_engine_options_t* options,
                          void* d_workspace, uint64_t workspace_bytes);
int blsw_engine_destroy(blsw_engine_t* e);
/* blsw_engine_submit for an engine with pk_mode / sig_mode Input (it works for every single-key engine): additionally writes
 * d_instance [n][n_instance_vars][6] u64 = each instance's instance_assignment (element 0 = one; Montgomery limbs, or canonical integers with
 * options.output_form 1), i.e. what ConstraintSystem::instance_assignment holds after synthesis. d_instance may be NULL. */
int blsw_engine_submit_io(blsw_engine_t* e, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, uint64_t* d_instance, uint64_t* d_witness,
                          uint64_t witness_stride, int32_t* d_result, void* stream);

/* Submits one batch of n independent (pk, msg, sig) instances: the witness vectors of the circuit of
 * src/constraints.rs:335-366 are written to d_witness and the gadget's output Boolean (constraints.rs:127) to d_result.
 *   d_pk_xy   [n][12] u64  affine G1 (x, y) Montgomery; (0,0) = point at infinity      (PublicKeyVar, constraints.rs:214-232)
 *   d_sig_xy  [n][24] u64  affine G2 (x.c0, x.c1, y.c0, y.c1); all zero = infinity     (SignatureVar, constraints.rs:234-249)
 *   d_msg     [n][msg_len] bytes                                                        (UInt8::new_witness_vec, constraints.rs:341)
 *   d_witness [n][witness_stride] field elements (48 B each), witness_stride >= layout.n_witness; may be NULL (results only)
 *   d_result  [n] int32, may be NULL
 * Device work is issued when max_steps batches are pending or at blsw_engine_flush; `stream` (hipStream_t, may be NULL)
 * is the stream on which THIS batch's inputs become valid (recorded per submit). Buffers must stay alive until the
 * step has completed. Steps are numbered 0, 1, 2, ... in submission order (blsw_engine_submitted).
 * Host blocking: the device work is asynchronous, but a submit that starts a new group in a group buffer whose previous group
 * (n_buffers groups back) has not finished writing its outputs WAITS on the host for that group (its staging is about to be
 * overwritten) — with n_buffers groups in flight this is the engine's back-pressure. In consumer mode it returns BLSW_ERR_BUSY
 * instead when that group still has steps held back by unreleased outputs. */
int blsw_engine_submit(blsw_engine_t* e, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, uint64_t* d_witness,
                       uint64_t witness_stride, int32_t* d_result, void* stream);
/* The same step from COMPRESSED bytes in one call (PublicKey::try_from / Signature::try_from + verify, src/bls.rs:219-242, 316-339 and
 * tests/tests.rs:239-268): d_pk48 [n][48] and d_sig96 [n][96] are decoded on `stream` into d_pk_xy [n][12] / d_sig_xy [n][24]
 * (caller buffers: they are the step's inputs and must stay alive like them; a point that fails to decode, or the identity,
 * becomes all zeros = the default point, as the reference's test substitutes it) with d_status [n][2] as blsw_decode_batch writes
 * it, and the step is submitted. d_result[i] = 1 iff both statuses are BLSW_ST_OK and the gadget's Boolean is true — the
 * verdict of tests/tests.rs:244-263. Witness vectors are written for every instance (for rejected inputs: the defined but
 * unsatisfiable assignment of the default points). BLSW_ERR_BUSY as blsw_engine_submit (nothing is issued then). */
int blsw_engine_submit_bytes(blsw_engine_t* e, const uint8_t* d_pk48, const uint8_t* d_sig96, const uint8_t* d_msg, uint64_t* d_pk_xy, uint64_t* d_sig_xy,
                             int32_t* d_status, uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, void* stream);
/* aggregate_verify (src/constraints.rs:153-191) through the engine, for an engine created with options.n_keys = K: one batch of n
 * instances of K keys each, same grouping / staging / streaming placement as blsw_engine_submit.
 *   d_pks_xy [n][K][12] u64, d_bitmap [n][K] bytes (0/1), d_sig_xy [n][24], d_msg [n][msg_len]
 *   d_witness [n][witness_stride] (stride >= blsw_layout_aggregate().n_witness; may be NULL), d_result [n] int32, d_count [n] uint32 (may be NULL) */
int blsw_engine_submit_aggregate(blsw_engine_t* e, const uint64_t* d_pks_xy, const uint8_t* d_bitmap, const uint64_t* d_sig_xy, const uint8_t* d_msg,
                                 uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, uint32_t* d_count, void* stream);
/* The N+1-pair product (blsw_verify_multi_batch's circuit) through the engine, for an engine created with options.n_pairs = K: one
 * batch of n instances, same grouping / staging / streaming placement / consumer mode as blsw_engine_submit — more instances are in
 * flight than output tensors exist (a vector is 4.19 GB at K = 128), which the direct call cannot do.
 *   d_pks_xy [n][K][12] u64, d_msgs [n][K][msg_len] bytes, d_sig_xy [n][24]
 *   d_witness [n][witness_stride] (stride >= blsw_layout_multi().n_witness; may be NULL), d_result [n] int32 */
int blsw_engine_submit_multi(blsw_engine_t* e, const uint64_t* d_pks_xy, const uint8_t* d_msgs, const uint64_t* d_sig_xy, uint64_t* d_witness,
                             uint64_t witness_stride, int32_t* d_result, void* stream);
/* Issues everything pending and makes `stream` wait for all batches submitted so far (asynchronous for the host). In consumer
 * mode: for all steps written so far (blsw_engine_materialised); the rest follow as their outputs are released. */
int blsw_engine_flush(blsw_engine_t* e, void* stream);
/* Streaming consumers (a prover draining witness tensors through a small ring of output buffers):
 *   blsw_engine_submitted / _launched / _materialised: number of steps submitted / whose chains are issued to the device /
 *   whose output writes are issued (equal to launched unless options.consumer_mode holds steps back);
 *   blsw_engine_wait_step: makes `stream` wait until step `seq` has written its output and results (seq < launched;
 *   BLSW_ERR_BUSY while seq >= materialised);
 *   blsw_engine_output_consumed: records on `stream` that the consumer is done with the output at d_output (witness tensor or
 *   compact buffer); the next step submitted with the same pointer does not overwrite it before that point.
 * consumer_mode 0: that release must have been recorded before the group containing the next user of the output is launched
 * (max_steps <= ring). consumer_mode 1: no such rule — steps are written into their outputs in submission order, each as soon
 * as its output has been released; blsw_engine_submit returns BLSW_ERR_BUSY instead of blocking when the group buffer it needs
 * still has unwritten steps (drain: wait_step + output_consumed of the materialised steps, then submit again).
 * At most 64 distinct output pointers may have a pending release, a held step or an accepted (not yet written) step at one time:
 * a consumer-mode submit reserves its output's slot BEFORE the step is taken and returns BLSW_ERR_BUSY with nothing queued or
 * launched when the table is full, and so does blsw_engine_output_consumed for an output the table has no slot for (releases whose
 * event has completed are recycled: release / drain and call again). */
int blsw_engine_submitted(blsw_engine_t* e, uint64_t* seq);
int blsw_engine_launched(blsw_engine_t* e, uint64_t* seq);
int blsw_engine_materialised(blsw_engine_t* e, uint64_t* seq);
int blsw_engine_wait_step(blsw_engine_t* e, uint64_t seq, void* stream);
int blsw_engine_output_consumed(blsw_engine_t* e, const void* d_output, void* stream);
/* Compact wire form of a step, for the multi-GPU all-gather of witness shards: the full vectors are 34 MB per
 * instance and 94 % of their elements are SHA-256 booleans, so every rank receiving the other ranks' shards over xGMI caps an
 * 8-GPU job far below the generation rate. In compact form a batch is its bit-packed SHA witnesses plus its field witnesses as
 * 48-byte elements (2.6 MB per instance: blsw_engine_compact_bytes per batch); that is what travels, and the receiver turns it
 * into the n witness vectors — bit-exact what blsw_engine_submit writes — with blsw_engine_expand_compact. Engines with
 * n % 64 == 0 and (max_steps > 1 or n_buffers > 1).
 *   blsw_engine_submit_compact: as blsw_engine_submit, the step's output is d_compact (compact_bytes bytes) instead of d_witness;
 *     blsw_engine_wait_step / _output_consumed (with the d_compact pointer) work as for witness tensors;
 *   blsw_engine_expand_compact: enqueues on `stream` the expansion of one compact batch (produced by ANY engine of the same
 *     n, msg_len and options, e.g. another rank's) into d_witness [n][witness_stride]. */
int blsw_engine_compact_bytes(blsw_engine_t* e, uint64_t* bytes);
int blsw_engine_submit_compact(blsw_engine_t* e, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, void* d_compact, int32_t* d_result,
                               void* stream);
/* the same for an aggregate_verify engine (options.n_keys = K): arguments of blsw_engine_submit_aggregate, output d_compact */
int blsw_engine_submit_aggregate_compact(blsw_engine_t* e, const uint64_t* d_pks_xy, const uint8_t* d_bitmap, const uint64_t* d_sig_xy, const uint8_t* d_msg,
                                         void* d_compact, int32_t* d_result, uint32_t* d_count, void* stream);
/* the same for an N+1-pair engine (options.n_pairs = K): arguments of blsw_engine_submit_multi, output d_compact (blsw_engine_compact_bytes bytes);
 * blsw_engine_expand_compact turns it into the n vectors of 4.19 GB (K = 128) on the receiver */
int blsw_engine_submit_multi_compact(blsw_engine_t* e, const uint64_t* d_pks_xy, const uint8_t* d_msgs, const uint64_t* d_sig_xy, void* d_compact, int32_t* d_result,
                                     void* stream);
int blsw_engine_expand_compact(blsw_engine_t* e, const void* d_compact, uint64_t* d_witness, uint64_t witness_stride, void* stream);
/* average duration (ms) of the bit->Fp expansion kernel launches issued since the previous call (HIP events on the stream they
 * ran on, at most 1024 launches); blocks until they have finished and resets the statistics */
int blsw_engine_expand_stats(blsw_engine_t* e, uint32_t* count, float* avg_ms);

/* Position-dependent, order-independent 128-bit digest of each instance's witness vector (the consumer-side check of the
 * sharded runs). With x_j the little-endian u32 words of the instance's n_witness * 12 words, 16-byte
 * piece q = (x_4q, x_4q+1, x_4q+2, x_4q+3), key_q = (q + 1) * 0x9E3779B1 mod 2^32 and A = 0x85EBCA6B (all sums of 32-bit words mod 2^32,
 * products 32 x 32 -> 64 bits):
 *     d[0] = sum_q (x_4q + key_q) * (x_4q+1 + key_q + A) + (x_4q+2 + key_q + 2 A) * (x_4q+3 + key_q + 3 A)   mod 2^64
 *            (the NH family of UMAC with a position-derived key: one multiply per 8 bytes)
 *     d[1] = lo | hi << 32,  lo = sum_q (x_4q ^ key_q) + (x_4q+2 ^ ~key_q),  hi = sum_q (x_4q+1 ^ key_q) + (x_4q+3 ^ ~key_q)   mod 2^32
 *   (ABI 9; ABI <= 8 summed a splitmix64 finalizer per u64 word: four 64-bit multiplies per 16 bytes made the kernel VALU-bound.)
 *   d_witness [n][witness_stride] elements, d_digest [n][2] u64, any n (launched in slices of 65 535 instances). Reads the tensor once at HBM speed. */
int blsw_witness_digest(const uint64_t* d_witness, uint64_t witness_stride, uint64_t n, uint32_t n_witness, uint64_t* d_digest, void* stream);

/* Constraint matrices (host only, no GPU): the R1CS whose witness vectors the entry points above fill, in arkworks'
 * ConstraintMatrices shape — what `cs.to_matrices()` returns after synthesising the circuit (the reference only prints the
 * system's size, src/constraints.rs:369-373). Constraint i is <A_i, z> * <B_i, z> = <C_i, z> with z = [1] ++ witness; column 0
 * is the constant one, column k is witness k - 1 (n_instance_vars = 1: the gadget allocates no public input). Rows are stored
 * CSR: row i of matrix m occupies entries [row_ptr[m][i], row_ptr[m][i + 1]) of col[m] / val[m] (val: 6 u64 Montgomery limbs
 * per entry), columns ascending, no zero coefficients. Circuit shape: (msg_len, n_keys, n_pairs) as blsw_layout (0, 1),
 * blsw_layout_aggregate (n_keys, 1) and blsw_layout_multi (0, n_pairs). Synthesised symbolically from the product's own
 * statement of the arkworks allocation rules (csrc/r1cs.cpp), independent of witness values: emit once per shape.
 * Usage: blsw_matrices_info -> allocate (n_constraints + 1) row pointers and nnz[m] entries per matrix -> blsw_matrices_fill. */
typedef struct {
    uint64_t n_constraints, n_instance_vars, n_witness;
    uint64_t nnz[3]; /* non-zeros of A, B, C */
} blsw_matrices_info_t;
typedef struct {
    uint64_t* row_ptr[3]; /* [n_constraints + 1] each */
    uint32_t* col[3];     /* [nnz[m]] */
    uint64_t* val[3];     /* [nnz[m]][6] */
} blsw_matrices_t;
int blsw_matrices_info(uint32_t msg_len, uint32_t n_keys, uint32_t n_pairs, blsw_matrices_info_t* out);
int blsw_matrices_fill(uint32_t msg_len, uint32_t n_keys, uint32_t n_pairs, const blsw_matrices_info_t* info, blsw_matrices_t* out);
/* the same for the single-key circuit of blsw_layout_params (params_mode 0 = the two calls above with n_keys 0, n_pairs 1) */
int blsw_matrices_info_params(uint32_t msg_len, uint32_t params_mode, blsw_matrices_info_t* out);
int blsw_matrices_fill_params(uint32_t msg_len, uint32_t params_mode, const blsw_matrices_info_t* info, blsw_matrices_t* out);
/* the same for the single-key circuit of blsw_layout_io: info->n_instance_vars = 1 + the number of public-input field elements; column numbering as in
 * blsw_layout_t.pk_mode (ark-relations' ConstraintMatrices: instance variables first, then the witnesses) */
int blsw_matrices_info_io(uint32_t msg_len, uint32_t pk_mode, uint32_t sig_mode, blsw_matrices_info_t* out);
int blsw_matrices_fill_io(uint32_t msg_len, uint32_t pk_mode, uint32_t sig_mode, const blsw_matrices_info_t* info, blsw_matrices_t* out);

/* Input decode (PublicKey::try_from / Signature::try_from -> deserialize_compressed, src/bls.rs:219-242, 316-339):
 *   d_pk48 [n][48], d_sig96 [n][96]  ZCash-format compressed points
 *   d_pk_xy [n][12], d_sig_xy [n][24] affine Montgomery coordinates in the layout blsw_engine_submit takes (identity / failure = zeros)
 *   d_status [n][2] int32: BLSW_ST_* of the key and of the signature (flags, x < p, on curve, prime-order subgroup;
 *   BLSW_ST_IDENTITY = well-formed encoding of the point at infinity)
 * tests/tests.rs:244-263 semantics: an instance verifies iff both statuses are BLSW_ST_OK and the gadget result is 1. */
int blsw_decode_batch(const uint8_t* d_pk48, const uint8_t* d_sig96, uint64_t n, uint64_t* d_pk_xy, uint64_t* d_sig_xy, int32_t* d_status, void* stream);

/* Signature::aggregate / PublicKey::aggregate (src/bls.rs:288-300, 183-195; tests/tests.rs:270-294 and the key sums of :296-334) for n
 * independent lists of k compressed points each: group 2 = signatures (G2, 96 bytes), group 1 = public keys (G1, 48 bytes).
 *   d_in [n][k][96 or 48]   d_out [n][96 or 48] the compressed sum (the infinity encoding for a sum that is the identity)
 *   d_status [n] int32: BLSW_ST_OK, or the status of the first point of the list that does not decode (try_from fails there: the
 *   reference unwraps); a well-formed point at infinity is a valid summand, as in aggregate_infinity_signature.json
 * An empty list is None in the reference: k == 0 is BLSW_ERR_ARG. Every point is decoded with its subgroup check by one lane, then one
 * lane per list adds. Asynchronous on `stream`; the workspace holds the decoded points. */
int blsw_aggregate_points_workspace_bytes(uint32_t group, uint64_t n, uint32_t k, uint64_t* bytes);
int blsw_aggregate_points_batch(uint32_t group, const uint8_t* d_in, uint32_t k, uint64_t n, uint8_t* d_out, int32_t* d_status, void* d_workspace,
                                uint64_t workspace_bytes, void* stream);

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
int blsw_sign_batch(const uint8_t* d_sk32_le, const uint8_t* d_msg, uint32_t msg_len, uint64_t n, uint8_t* d_sig96, uint64_t* d_sig_xy, uint8_t* d_pk48,
                    uint64_t* d_pk_xy, int32_t* d_status, void* d_workspace, uint64_t workspace_bytes, void* stream);

/* BLS::verify (src/bls.rs:427-458; tests/tests.rs:239-268) for a batch, as VALUES — the native algorithm, not the circuit: decode of the compressed key and
 * signature with the endomorphism subgroup checks (as blsw_decode_batch), hash_to_g2 (as blsw_hash_to_g2_batch), a two-pair Miller loop over projective
 * line coefficients (no inversion per step) and the final exponentiation on six lanes per instance, is_one.
 *   d_pk48 [n][48], d_sig96 [n][96], d_msg [n][msg_len];  d_result [n] int32: 1 iff both points decode to non-identity points of the prime-order
 *   subgroups and e(-g1, sig) * e(pk, H(msg)) == 1 (every Err of the reference's verify counts as false, as tests/tests.rs:244-263 does);
 *   d_status [n][2] BLSW_ST_* of key and signature. Asynchronous on `stream` after one synchronising descriptor copy. fast_aggregate_verify
 *   (tests/tests.rs:296-334) = blsw_aggregate_points_batch over the keys, then this. */
int blsw_verify_workspace_bytes(uint64_t n, uint32_t msg_len, uint64_t* bytes);
int blsw_verify_batch(const uint8_t* d_pk48, const uint8_t* d_sig96, const uint8_t* d_msg, uint32_t msg_len, uint64_t n, int32_t* d_result, int32_t* d_status,
                      void* d_workspace, uint64_t workspace_bytes, void* stream);

/* Device micro-benchmarks that give the VALU roofline its MEASURED denominator:
 * which = 0: v_mad_u64_u32 rate (32x32+64 multiply-adds per second, all CUs); 1: Fp Montgomery products per second;
 * 2: Fp inversions (safegcd) per second; 3: Fp products per second inside witness-emitting Fp2 mul + sqr;
 * 4: Fp products per second of the 12 x 32-bit CIOS formulation (cross-check of the shipped 14 x 28-bit one). */
int blsw_microbench(int which, uint32_t iters, uint32_t blocks, double* ops_per_s);
/* Same-box yardstick of the HBM roofline: a plain fill of the caller's device buffer (16-byte aligned; overwritten) in the store geometry of the
 * expansion kernel (384 threads x 8 pieces of 16 bytes), `reps` passes after one warm-up pass on the NULL stream of the current device; bytes per
 * second. Boxes of one pool differ by 20-30 % in what they give a write stream: a reader of a bench line needs this next to the kernel's own rate. */
int blsw_fill_rate(void* d_buf, uint64_t bytes, uint32_t reps, double* bytes_per_s);

int blsw_version(void);

#ifdef __cplusplus
}
#endif
#endif
