// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// C entry points (ctypes) over the CPU restatement. Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library.
// Native-scheme semantics follow /root/reference/src/bls.rs:183-195, 288-300, 411-458, 477-493 and
// tests/tests.rs:203-364; point encodings follow ark-bls12-381 ^0.4.0 (ZCash format).
#include <cstdio>
#include <thread>
#include "circuit.h"

using namespace orc;

namespace {

// ---------------------------------------------------------------- ZCash-format (de)serialisation
// status codes shared with the product ABI (include/blsw.h)
enum { ST_OK = 0, ST_BAD_ENCODING = 1, ST_NOT_ON_CURVE = 2, ST_NOT_IN_SUBGROUP = 3, ST_IDENTITY = 4 };

int g1_decompress(const uint8_t* in, size_t len, G1Aff& out, bool subgroup_check = true) {
    if (len != 48) return ST_BAD_ENCODING;
    bool c = in[0] >> 7, inf = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    if (sort && (!c || inf)) return ST_BAD_ENCODING;
    if (!c) return ST_BAD_ENCODING;
    if (inf) {
        out = {fp_zero(), fp_zero(), true};
        return ST_OK;
    }
    uint8_t b[48];
    memcpy(b, in, 48);
    b[0] &= 0x1f;
    Fp x;
    if (!fp_from_bytes_be(x, b)) return ST_BAD_ENCODING;
    Fp y;
    if (!fp_sqrt(y, fp_add(fp_mul(fp_sqr(x), x), fp_from_u64(4)))) return ST_NOT_ON_CURVE;
    if (fp_is_lexicographically_largest(y) != sort) y = fp_neg(y);
    out = {x, y, false};
    if (subgroup_check && !aff_in_subgroup<FpT>(out)) return ST_NOT_IN_SUBGROUP;
    return ST_OK;
}
int g2_decompress(const uint8_t* in, size_t len, G2Aff& out, bool subgroup_check = true) {
    if (len != 96) return ST_BAD_ENCODING;
    bool c = in[0] >> 7, inf = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    if (sort && (!c || inf)) return ST_BAD_ENCODING;
    if (!c) return ST_BAD_ENCODING;
    if (inf) {
        out = {fp2_zero(), fp2_zero(), true};
        return ST_OK;
    }
    uint8_t b[96];
    memcpy(b, in, 96);
    b[0] &= 0x1f;
    Fp2 x;
    if (!fp_from_bytes_be(x.c1, b)) return ST_BAD_ENCODING;
    if (!fp_from_bytes_be(x.c0, b + 48)) return ST_BAD_ENCODING;
    Fp2 y;
    Fp2 rhs = fp2_add(fp2_mul(fp2_sqr(x), x), Fp2T::coeff_b());
    if (!fp2_sqrt(y, rhs)) return ST_NOT_ON_CURVE;
    if (fp2_is_lexicographically_largest(y) != sort) y = fp2_neg(y);
    out = {x, y, false};
    if (subgroup_check && !aff_in_subgroup<Fp2T>(out)) return ST_NOT_IN_SUBGROUP;
    return ST_OK;
}
void g1_compress(const G1Aff& p, uint8_t* out) {
    memset(out, 0, 48);
    if (p.inf) {
        out[0] = 0xc0;
        return;
    }
    fp_to_bytes_be(out, p.x);
    out[0] |= 0x80;
    if (fp_is_lexicographically_largest(p.y)) out[0] |= 0x20;
}
void g2_compress(const G2Aff& p, uint8_t* out) {
    memset(out, 0, 96);
    if (p.inf) {
        out[0] = 0xc0;
        return;
    }
    fp_to_bytes_be(out, p.x.c1);
    fp_to_bytes_be(out + 48, p.x.c0);
    out[0] |= 0x80;
    if (fp2_is_lexicographically_largest(p.y)) out[0] |= 0x20;
}

struct ValueScope {  // run gadget code in witness-only mode on a private CS
    CS cs;
    CS* prev;
    explicit ValueScope(bool record = false) {
        cs.record = record;
        prev = cur_cs();
        cur_cs() = &cs;
    }
    ~ValueScope() { cur_cs() = prev; }
};

G2Aff hash_to_g2_native(const uint8_t* msg, size_t len, HashTrace* tr = nullptr) {
    ValueScope s;
    std::vector<U8> m = u8const_vec(msg, len);
    G2Var h = hash_to_g2_with_cons(m, tr);
    return h.value_affine();
}
// e(-g1, sig) * e(pk, h) == 1 through the same (affine) pairing restatement
bool pairing_check(const G1Aff& pk, const G2Aff& sig, const G2Aff& h) {
    ValueScope s;
    G1Var g1 = pv_constant<FpT>(g1_generator());
    G1Var g1n = pv_negate<FpT>(g1);
    // witnesses so that the variable code path (not constant folding) is exercised
    G1Var pkv = pv_new_witness_omit_check<FpT>(pk);
    G2Var sigv = pv_new_witness_omit_check<Fp2T>(sig);
    G2Var hv = pv_new_witness_omit_check<Fp2T>(h);
    Fp12Var ml = miller_loop({g1_prepare(g1n), g1_prepare(pkv)}, {g2_prepare(sigv), g2_prepare(hv)});
    Fp12Var fe = final_exponentiation(ml);
    return fp12_eq(fe.val(), fp12_one());
}
// bls.rs:427-458
int native_verify(const G1Aff& pk, const uint8_t* msg, size_t len, const G2Aff& sig) {
    if (pk.inf) return 0;  // Err(InvalidPublicKey)
    if (!aff_on_curve<FpT>(pk) || !aff_in_subgroup<FpT>(pk)) return 0;
    if (!aff_on_curve<Fp2T>(sig) || !aff_in_subgroup<Fp2T>(sig)) return 0;
    if (sig.inf) return 0;  // e(-g1, O) * e(pk, h) != 1 for pk != O; also avoids the affine loop's division by zero
    G2Aff h = hash_to_g2_native(msg, len);
    return pairing_check(pk, sig, h) ? 1 : 0;
}

void aff1_to_limbs(const G1Aff& a, uint64_t* out) {
    memcpy(out, a.x.l, 48);
    memcpy(out + 6, a.y.l, 48);
}
void aff2_to_limbs(const G2Aff& a, uint64_t* out) {
    memcpy(out, a.x.c0.l, 48);
    memcpy(out + 6, a.x.c1.l, 48);
    memcpy(out + 12, a.y.c0.l, 48);
    memcpy(out + 18, a.y.c1.l, 48);
}
G1Aff limbs_to_aff1(const uint64_t* in) {
    G1Aff a;
    memcpy(a.x.l, in, 48);
    memcpy(a.y.l, in + 6, 48);
    a.inf = fp_is_zero(a.x) && fp_is_zero(a.y);
    return a;
}
G2Aff limbs_to_aff2(const uint64_t* in) {
    G2Aff a;
    memcpy(a.x.c0.l, in, 48);
    memcpy(a.x.c1.l, in + 6, 48);
    memcpy(a.y.c0.l, in + 12, 48);
    memcpy(a.y.c1.l, in + 18, 48);
    a.inf = fp2_is_zero(a.x) && fp2_is_zero(a.y);
    return a;
}

}  // namespace

extern "C" {

// ---- constants & field self-checks
int orc_selfcheck() {
    // R = 2^384 mod p, R2 = R^2 mod p, inverse implementations agree, Frobenius tables sane
    Fp one = fp_one();
    Fp two = fp_add(one, one);
    Fp acc = fp_from_u64(1);
    if (!fp_eq(acc, one)) return 1;
    Fp x = fp_from_hex("1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcdef");
    if (!fp_eq(fp_inv(x), fp_inv_fermat(x))) return 2;
    if (!fp_eq(fp_mul(x, fp_inv(x)), one)) return 3;
    Fp2 xi = {one, one};
    // gamma^6 == xi^(p-1) == conj(xi)/xi
    Fp2 g = frob_tables().f12c1[1];
    Fp2 g6 = fp2_mul(fp2_sqr(fp2_mul(fp2_sqr(g), g)), fp2_one());
    g6 = fp2_sqr(fp2_mul(fp2_sqr(g), g));
    if (!fp2_eq(fp2_mul(g6, xi), fp2_conj(xi))) return 4;
    const MapperConsts& K = mapper_consts();
    // hasher.rs:805-808 relations
    Fp2 z3 = fp2_mul(fp2_sqr(K.Z), K.Z);
    if (!fp2_eq(z3, fp2_mul(fp2_mul(fp2_sqr(K.C5), K.C2), K.C3))) return 5;
    if (!fp2_eq(z3, fp2_mul(fp2_sqr(K.C4), K.C3))) return 6;
    if (!fp2_eq(K.C2, fp2_sqr(K.C3))) return 7;
    if (!fp2_eq(fp2_sqr(K.C2), fp2_neg(fp2_one()))) return 8;
    (void)two;
    return 0;
}
void orc_constants(uint64_t* p, uint64_t* r1, uint64_t* r2, uint64_t* inv) {
    memcpy(p, P_LIMBS, 48);
    memcpy(r1, R1_LIMBS, 48);
    memcpy(r2, R2_LIMBS, 48);
    *inv = P_INV;
}

// ---- encodings
int orc_g1_decompress(const uint8_t* in, size_t len, uint64_t* out_xy /*12*/, int* is_inf) {
    G1Aff a;
    int st = g1_decompress(in, len, a);
    if (st == ST_OK) {
        aff1_to_limbs(a, out_xy);
        *is_inf = a.inf;
    }
    return st;
}
int orc_g2_decompress(const uint8_t* in, size_t len, uint64_t* out_xy /*24*/, int* is_inf) {
    G2Aff a;
    int st = g2_decompress(in, len, a);
    if (st == ST_OK) {
        aff2_to_limbs(a, out_xy);
        *is_inf = a.inf;
    }
    return st;
}
void orc_g1_compress(const uint64_t* xy, uint8_t* out) { g1_compress(limbs_to_aff1(xy), out); }
void orc_g2_compress(const uint64_t* xy, uint8_t* out) { g2_compress(limbs_to_aff2(xy), out); }

// ---- native scheme (bls.rs)
// sk: 4 little-endian u64 limbs, already reduced mod r
void orc_sk_to_pk(const uint64_t* sk, uint8_t* out48) {
    G1Aff pk = jac_to_aff<FpT>(jac_mul<FpT>(jac_from_aff<FpT>(g1_generator()), sk, 4));
    g1_compress(pk, out48);
}
// returns 0 on success, 1 if sk == 0 (bls.rs:417-419)
int orc_sign(const uint64_t* sk, const uint8_t* msg, size_t len, uint8_t* out96) {
    if ((sk[0] | sk[1] | sk[2] | sk[3]) == 0) return 1;
    G2Aff h = hash_to_g2_native(msg, len);
    G2Aff s = jac_to_aff<Fp2T>(jac_mul<Fp2T>(jac_from_aff<Fp2T>(h), sk, 4));
    g2_compress(s, out96);
    return 0;
}
void orc_hash_to_g2(const uint8_t* msg, size_t len, uint8_t* out96, uint64_t* out_affine /*24 or null*/) {
    G2Aff h = hash_to_g2_native(msg, len);
    g2_compress(h, out96);
    if (out_affine) aff2_to_limbs(h, out_affine);
}
// hasher.rs:110-173 with an arbitrary DST and output length (test vectors hasher.rs:819-886)
void orc_expand(const uint8_t* msg, size_t len, const uint8_t* dst, size_t dst_len, size_t len_in_bytes, uint8_t* out) {
    ValueScope s;
    std::vector<U8> m = u8witness_vec(msg, len);
    std::vector<U8> d = u8witness_vec(dst, dst_len);
    std::vector<U8> r = hasher_expand(m, d, len_in_bytes);
    for (size_t i = 0; i < len_in_bytes; i++) out[i] = r[i].value();
}
// tests/tests.rs:239-268 semantics: undecodable key/signature fall back to the identity; any Err => false
int orc_verify_bytes(const uint8_t* pk, size_t pk_len, const uint8_t* msg, size_t len, const uint8_t* sig, size_t sig_len) {
    G1Aff p = {fp_zero(), fp_zero(), true};
    G2Aff s = {fp2_zero(), fp2_zero(), true};
    G1Aff pt;
    if (g1_decompress(pk, pk_len, pt) == ST_OK) p = pt;
    G2Aff st;
    if (g2_decompress(sig, sig_len, st) == ST_OK) s = st;
    return native_verify(p, msg, len, s);
}
// bls.rs:183-195 / 288-300: sums of decoded points; returns -1 for an empty list (None), else 0
int orc_aggregate_g1(const uint8_t* pks, size_t n, uint8_t* out48) {
    if (n == 0) return -1;
    Jac<FpT> acc = jac_identity<FpT>();
    for (size_t i = 0; i < n; i++) {
        G1Aff a;
        if (g1_decompress(pks + 48 * i, 48, a) != ST_OK) return -2;
        acc = jac_add<FpT>(acc, jac_from_aff<FpT>(a));
    }
    g1_compress(jac_to_aff<FpT>(acc), out48);
    return 0;
}
int orc_aggregate_g2(const uint8_t* sigs, size_t n, uint8_t* out96) {
    if (n == 0) return -1;
    Jac<Fp2T> acc = jac_identity<Fp2T>();
    for (size_t i = 0; i < n; i++) {
        G2Aff a;
        if (g2_decompress(sigs + 96 * i, 96, a) != ST_OK) return -2;
        acc = jac_add<Fp2T>(acc, jac_from_aff<Fp2T>(a));
    }
    g2_compress(jac_to_aff<Fp2T>(acc), out96);
    return 0;
}

// ---- the circuit: witness vector of constraints.rs:335-366
// pk_xy: 12 limbs (affine, Montgomery), sig_xy: 24 limbs. out_witness may be null (count only).
// Returns the number of witnesses; *result receives the gadget's output boolean.
uint64_t orc_witness(const uint64_t* pk_xy, const uint8_t* msg, size_t msg_len, const uint64_t* sig_xy, uint64_t* out_witness,
                     uint64_t out_capacity_elems, uint64_t* n_constraints, int* result) {
    ValueScope s;
    VerifyTrace tr;
    Bool r = bls_verify_circuit(limbs_to_aff1(pk_xy), msg, msg_len, limbs_to_aff2(sig_xy), &tr);
    if (result) *result = r.val;
    if (n_constraints) *n_constraints = s.cs.ncons;
    uint64_t n = s.cs.wit.size();
    if (out_witness) {
        uint64_t m = std::min(n, out_capacity_elems);
        memcpy(out_witness, s.cs.wit.data(), m * 48);
    }
    return n;
}
// the same circuit with the parameters allocated as witnesses (constraints.rs:198-211 with AllocationMode::Witness)
uint64_t orc_witness_params(const uint64_t* pk_xy, const uint8_t* msg, size_t msg_len, const uint64_t* sig_xy, int params_witness,
                            uint64_t* out_witness, uint64_t out_capacity_elems, uint64_t* n_constraints, int* result) {
    ValueScope s;
    Bool r = bls_verify_circuit(limbs_to_aff1(pk_xy), msg, msg_len, limbs_to_aff2(sig_xy), nullptr, params_witness != 0);
    if (result) *result = r.val;
    if (n_constraints) *n_constraints = s.cs.ncons;
    uint64_t n = s.cs.wit.size();
    if (out_witness) memcpy(out_witness, s.cs.wit.data(), std::min(n, out_capacity_elems) * 48);
    return n;
}
// the same circuit with the key and / or the signature allocated as public inputs (constraints.rs:214-249 with AllocationMode::Input):
// out_instance receives instance_assignment = [1, inputs...] (n_inst = 1 + 3 pk_input + 6 sig_input elements)
uint64_t orc_witness_io(const uint64_t* pk_xy, const uint8_t* msg, size_t msg_len, const uint64_t* sig_xy, int pk_input, int sig_input, uint64_t* out_witness,
                        uint64_t out_capacity_elems, uint64_t* out_instance, uint64_t* n_constraints, int* result) {
    ValueScope s;
    s.cs.n_inst = 1 + (pk_input ? 3 : 0) + (sig_input ? 6 : 0);
    Bool r = bls_verify_circuit(limbs_to_aff1(pk_xy), msg, msg_len, limbs_to_aff2(sig_xy), nullptr, false, pk_input != 0, sig_input != 0);
    if (result) *result = r.val;
    if (n_constraints) *n_constraints = s.cs.ncons;
    uint64_t n = s.cs.wit.size();
    if (out_witness) memcpy(out_witness, s.cs.wit.data(), std::min(n, out_capacity_elems) * 48);
    if (out_instance) {
        Fp one = fp_one();
        memcpy(out_instance, &one, 48);
        if (!s.cs.inst.empty()) memcpy(out_instance + 6, s.cs.inst.data(), s.cs.inst.size() * 48);
    }
    return n;
}
// segment marks of that circuit shape (as orc_layout_params)
uint64_t orc_layout_io(size_t msg_len, int pk_input, int sig_input, uint64_t* starts, uint64_t cap, char* names_buf, size_t names_cap, uint64_t* n_wit, uint64_t* n_cons) {
    std::vector<uint8_t> msg(msg_len, 0);
    G2Aff h = hash_to_g2_native(msg.data(), msg_len);
    ValueScope s;
    s.cs.n_inst = 1 + (pk_input ? 3 : 0) + (sig_input ? 6 : 0);
    bls_verify_circuit(g1_generator(), msg.data(), msg_len, h, nullptr, false, pk_input != 0, sig_input != 0);
    std::string names;
    uint64_t k = 0;
    for (auto& m : s.cs.marks) {
        if (k < cap) starts[k] = m.second;
        names += m.first;
        names += '\n';
        k++;
    }
    if (names_buf && names_cap) {
        size_t c = std::min(names.size(), names_cap - 1);
        memcpy(names_buf, names.data(), c);
        names_buf[c] = 0;
    }
    if (n_wit) *n_wit = s.cs.wit.size();
    if (n_cons) *n_cons = s.cs.ncons;
    return k;
}
uint64_t orc_layout_params(size_t msg_len, int params_witness, uint64_t* starts, uint64_t cap, char* names_buf, size_t names_cap, uint64_t* n_wit,
                           uint64_t* n_cons) {
    std::vector<uint8_t> msg(msg_len, 0);
    G2Aff h = hash_to_g2_native(msg.data(), msg_len);
    ValueScope s;
    bls_verify_circuit(g1_generator(), msg.data(), msg_len, h, nullptr, params_witness != 0);
    std::string names;
    uint64_t k = 0;
    for (auto& m : s.cs.marks) {
        if (k < cap) starts[k] = m.second;
        names += m.first;
        names += '\n';
        k++;
    }
    if (names_buf && names_cap) {
        size_t c = std::min(names.size(), names_cap - 1);
        memcpy(names_buf, names.data(), c);
        names_buf[c] = 0;
    }
    if (n_wit) *n_wit = s.cs.wit.size();
    if (n_cons) *n_cons = s.cs.ncons;
    return k;
}
// batch, multi-threaded (CPU baseline): only results and the per-instance 64-bit FNV digest of the witness bytes
static uint64_t fnv1a(const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n; i++) {
        h ^= b[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}
void orc_witness_batch(const uint64_t* pk_xy, const uint8_t* msgs, size_t msg_len, const uint64_t* sig_xy, uint64_t n, int threads, int* results,
                       uint64_t* digests) {
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            ValueScope s;
            Bool r = bls_verify_circuit(limbs_to_aff1(pk_xy + 12 * i), msgs + msg_len * i, msg_len, limbs_to_aff2(sig_xy + 24 * i));
            if (results) results[i] = r.val;
            if (digests) digests[i] = fnv1a(s.cs.wit.data(), s.cs.wit.size() * 48);
        }
    };
    if (threads <= 1) {
        work(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(work, n * t / threads, n * (t + 1) / threads);
    for (auto& t : th) t.join();
}
// segment table: writes up to cap (name, start) pairs; names joined by '\n' into names_buf
uint64_t orc_layout(size_t msg_len, uint64_t* starts, uint64_t cap, char* names_buf, size_t names_cap, uint64_t* n_wit, uint64_t* n_cons) {
    // a valid dummy instance: pk = g1, sig = sk*H(msg) with sk = 1
    std::vector<uint8_t> msg(msg_len, 0);
    G2Aff h = hash_to_g2_native(msg.data(), msg_len);
    ValueScope s;
    bls_verify_circuit(g1_generator(), msg.data(), msg_len, h);
    std::string names;
    uint64_t k = 0;
    for (auto& m : s.cs.marks) {
        if (k < cap) starts[k] = m.second;
        names += m.first;
        names += '\n';
        k++;
    }
    if (names_buf && names_cap) {
        size_t c = std::min(names.size(), names_cap - 1);
        memcpy(names_buf, names.data(), c);
        names_buf[c] = 0;
    }
    if (n_wit) *n_wit = s.cs.wit.size();
    if (n_cons) *n_cons = s.cs.ncons;
    return k;
}
// aggregate_verify circuit (constraints.rs:378-441): pks_xy [k][12], bitmap [k]
uint64_t orc_witness_aggregate(const uint64_t* pks_xy, const uint8_t* bitmap, uint64_t k, const uint8_t* msg, size_t msg_len, const uint64_t* sig_xy,
                               uint64_t* out_witness, uint64_t out_capacity_elems, uint64_t* n_constraints, int* result, uint32_t* count,
                               uint64_t* mark_starts, uint64_t mark_cap, char* names_buf, size_t names_cap) {
    ValueScope s;
    std::vector<G1Aff> pks;
    for (uint64_t i = 0; i < k; i++) pks.push_back(limbs_to_aff1(pks_xy + 12 * i));
    std::vector<uint8_t> bm(bitmap, bitmap + k);
    uint32_t cv = 0;
    Bool r = bls_aggregate_verify_circuit(pks, bm, msg, msg_len, limbs_to_aff2(sig_xy), &cv);
    if (result) *result = r.val;
    if (count) *count = cv;
    if (n_constraints) *n_constraints = s.cs.ncons;
    uint64_t n = s.cs.wit.size();
    if (out_witness) memcpy(out_witness, s.cs.wit.data(), std::min(n, out_capacity_elems) * 48);
    std::string names;
    uint64_t j = 0;
    for (auto& m : s.cs.marks) {
        if (mark_starts && j < mark_cap) mark_starts[j] = m.second;
        names += m.first;
        names += '\n';
        j++;
    }
    if (names_buf && names_cap) {
        size_t c = std::min(names.size(), names_cap - 1);
        memcpy(names_buf, names.data(), c);
        names_buf[c] = 0;
    }
    return n;
}
// N+1-pair product circuit: pks_xy [k][12], msgs [k][msg_len], one signature. Marks as orc_witness_aggregate.
uint64_t orc_witness_multi(const uint64_t* pks_xy, const uint8_t* msgs, size_t msg_len, uint64_t k, const uint64_t* sig_xy, uint64_t* out_witness,
                           uint64_t out_capacity_elems, uint64_t* n_constraints, int* result, uint64_t* mark_starts, uint64_t mark_cap, char* names_buf,
                           size_t names_cap) {
    ValueScope s;
    std::vector<G1Aff> pks;
    for (uint64_t i = 0; i < k; i++) pks.push_back(limbs_to_aff1(pks_xy + 12 * i));
    Bool r = bls_verify_multi_circuit(pks, msgs, msg_len, limbs_to_aff2(sig_xy));
    if (result) *result = r.val;
    if (n_constraints) *n_constraints = s.cs.ncons;
    uint64_t n = s.cs.wit.size();
    if (out_witness) memcpy(out_witness, s.cs.wit.data(), std::min(n, out_capacity_elems) * 48);
    std::string names;
    uint64_t j = 0;
    for (auto& m : s.cs.marks) {
        if (mark_starts && j < mark_cap) mark_starts[j] = m.second;
        names += m.first;
        names += '\n';
        j++;
    }
    if (names_buf && names_cap) {
        size_t c = std::min(names.size(), names_cap - 1);
        memcpy(names_buf, names.data(), c);
        names_buf[c] = 0;
    }
    return n;
}
// record the full R1CS for the instance and check A z o B z = C z against a witness vector.
// witness == null: use the oracle's own. Returns the index of the first unsatisfied constraint, or -1.
int64_t orc_check_satisfied(const uint64_t* pk_xy, const uint8_t* msg, size_t msg_len, const uint64_t* sig_xy, const uint64_t* witness,
                            uint64_t n_witness, uint64_t* n_constraints, uint64_t* n_nonzero) {
    ValueScope s(true);
    bls_verify_circuit(limbs_to_aff1(pk_xy), msg, msg_len, limbs_to_aff2(sig_xy));
    std::vector<Fp> w;
    if (witness) {
        if (n_witness != s.cs.wit.size()) return -2;
        w.resize(n_witness);
        memcpy(w.data(), witness, n_witness * 48);
    } else
        w = s.cs.wit;
    if (n_constraints) *n_constraints = s.cs.ncons;
    uint64_t nnz = 0;
    int64_t bad = -1;
    for (size_t i = 0; i < s.cs.A.size(); i++) {
        nnz += s.cs.A[i]->size() + s.cs.B[i]->size() + s.cs.C[i]->size();
        Fp a = lc_eval(*s.cs.A[i], w), b = lc_eval(*s.cs.B[i], w), c = lc_eval(*s.cs.C[i], w);
        if (!fp_eq(fp_mul(a, b), c) && bad < 0) bad = (int64_t)i;
    }
    if (n_nonzero) *n_nonzero = nnz;
    return bad;
}
// CSR export of the recorded R1CS (A, B, C) for a circuit SHAPE: (msg_len, n_keys, n_pairs) as the product's blsw_matrices_*.
// The instance is a dummy (generator keys, sig = H(0...0)): the matrices do not depend on values. Two-phase: with null
// arrays only the counts are returned in nnz[3]; returns the number of constraints.
static uint64_t export_csr(ValueScope& s, uint64_t* nnz, uint64_t* n_witness, uint64_t** row_ptr, uint32_t** col, uint64_t** val);
uint64_t orc_matrices_params(size_t msg_len, int params_witness, uint64_t* nnz, uint64_t* n_witness, uint64_t** row_ptr, uint32_t** col, uint64_t** val);
uint64_t orc_matrices(size_t msg_len, uint64_t n_keys, uint64_t n_pairs, uint64_t* nnz, uint64_t* n_witness, uint64_t** row_ptr, uint32_t** col, uint64_t** val) {
    if (!n_keys && n_pairs <= 1) return orc_matrices_params(msg_len, 0, nnz, n_witness, row_ptr, col, val);
    std::vector<uint8_t> msg(msg_len * (n_pairs ? n_pairs : 1) + 1, 0);
    G2Aff h = hash_to_g2_native(msg.data(), msg_len);
    ValueScope s(true);
    if (n_keys) {
        std::vector<G1Aff> pks(n_keys, g1_generator());
        std::vector<uint8_t> bm(n_keys, 1);
        bls_aggregate_verify_circuit(pks, bm, msg.data(), msg_len, h, nullptr);
    } else if (n_pairs > 1) {
        std::vector<G1Aff> pks(n_pairs, g1_generator());
        bls_verify_multi_circuit(pks, msg.data(), msg_len, h);
    }
    return export_csr(s, nnz, n_witness, row_ptr, col, val);
}
// the single-key circuit with the key / the signature as public inputs; columns: 0 = One, 1 .. n_inst - 1 = the inputs, n_inst + k = witness k
uint64_t orc_matrices_io(size_t msg_len, int pk_input, int sig_input, uint64_t* nnz, uint64_t* n_witness, uint64_t** row_ptr, uint32_t** col, uint64_t** val) {
    std::vector<uint8_t> msg(msg_len + 1, 0);
    G2Aff h = hash_to_g2_native(msg.data(), msg_len);
    ValueScope s(true);
    s.cs.n_inst = 1 + (pk_input ? 3 : 0) + (sig_input ? 6 : 0);
    bls_verify_circuit(g1_generator(), msg.data(), msg_len, h, nullptr, false, pk_input != 0, sig_input != 0);
    return export_csr(s, nnz, n_witness, row_ptr, col, val);
}
// the single-key circuit, parameters Constant or allocated as witnesses (constraints.rs:198-211)
uint64_t orc_matrices_params(size_t msg_len, int params_witness, uint64_t* nnz, uint64_t* n_witness, uint64_t** row_ptr, uint32_t** col, uint64_t** val) {
    std::vector<uint8_t> msg(msg_len + 1, 0);
    G2Aff h = hash_to_g2_native(msg.data(), msg_len);
    ValueScope s(true);
    bls_verify_circuit(g1_generator(), msg.data(), msg_len, h, nullptr, params_witness != 0);
    return export_csr(s, nnz, n_witness, row_ptr, col, val);
}
static uint64_t export_csr(ValueScope& s, uint64_t* nnz, uint64_t* n_witness, uint64_t** row_ptr, uint32_t** col, uint64_t** val) {
    const std::vector<LC>* M[3] = {&s.cs.A, &s.cs.B, &s.cs.C};
    for (int m = 0; m < 3; m++) {
        uint64_t k = 0;
        for (size_t i = 0; i < M[m]->size(); i++) {
            const LCv& row = *(*M[m])[i];
            if (row_ptr) row_ptr[m][i] = k;
            if (col)
                for (size_t t = 0; t < row.size(); t++) {
                    col[m][k + t] = row[t].v;
                    memcpy(val[m] + (k + t) * 6, row[t].c.l, 48);
                }
            k += row.size();
        }
        if (row_ptr) row_ptr[m][M[m]->size()] = k;
        nnz[m] = k;
    }
    if (n_witness) *n_witness = s.cs.wit.size();
    return s.cs.ncons;
}
// checkpoints for debugging the device path: u0,u1 (2x Fp2), Q0,Q1,R,H affine (4 x 24 limbs), f_miller, f_final (2 x 72 limbs)
void orc_trace(const uint64_t* pk_xy, const uint8_t* msg, size_t msg_len, const uint64_t* sig_xy, uint64_t* out /* 24 + 96 + 144 limbs */) {
    ValueScope s;
    VerifyTrace tr;
    bls_verify_circuit(limbs_to_aff1(pk_xy), msg, msg_len, limbs_to_aff2(sig_xy), &tr);
    uint64_t* o = out;
    memcpy(o, &tr.hash.u[0], 96);
    o += 12;
    memcpy(o, &tr.hash.u[1], 96);
    o += 12;
    aff2_to_limbs(tr.hash.q[0], o);
    o += 24;
    aff2_to_limbs(tr.hash.q[1], o);
    o += 24;
    aff2_to_limbs(tr.hash.r, o);
    o += 24;
    aff2_to_limbs(tr.hash.h, o);
    o += 24;
    memcpy(o, &tr.f_miller, 576);
    o += 72;
    memcpy(o, &tr.f_final, 576);
}
void orc_opcount(const uint64_t* pk_xy, const uint8_t* msg, size_t msg_len, const uint64_t* sig_xy, uint64_t* out3) {
    opcount() = OpCount();
    ValueScope s;
    bls_verify_circuit(limbs_to_aff1(pk_xy), msg, msg_len, limbs_to_aff2(sig_xy));
    out3[0] = opcount().fp_mul;
    out3[1] = opcount().fp_inv;
    out3[2] = opcount().sha_blocks;
}
// raw field helpers for kernel unit tests (Montgomery limbs in/out)
void orc_fp_mul(const uint64_t* a, const uint64_t* b, uint64_t* r) {
    Fp x, y;
    memcpy(x.l, a, 48);
    memcpy(y.l, b, 48);
    Fp z = fp_mul(x, y);
    memcpy(r, z.l, 48);
}
void orc_fp_inv(const uint64_t* a, uint64_t* r) {
    Fp x;
    memcpy(x.l, a, 48);
    Fp z = fp_inv(x);
    memcpy(r, z.l, 48);
}
void orc_fp_add(const uint64_t* a, const uint64_t* b, uint64_t* r) {
    Fp x, y;
    memcpy(x.l, a, 48);
    memcpy(y.l, b, 48);
    Fp z = fp_add(x, y);
    memcpy(r, z.l, 48);
}
void orc_fp_sub(const uint64_t* a, const uint64_t* b, uint64_t* r) {
    Fp x, y;
    memcpy(x.l, a, 48);
    memcpy(y.l, b, 48);
    Fp z = fp_sub(x, y);
    memcpy(r, z.l, 48);
}
}
