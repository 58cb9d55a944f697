// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Minimal restatement of the pieces of ark-relations ^0.4.0 / ark-r1cs-std ^0.4.0
// (third-party, NOT vendored under /root/reference; Cargo.toml:20,24) that decide
// WHICH values become witnesses and in what order when the reference's gadget
// (src/constraints.rs:90-128, src/hasher.rs) is synthesised:
//   ConstraintSystem (witness_assignment in allocation order, instance = [1]),
//   FpVar {Constant|Var}, Boolean {Constant|Is|Not}, UInt8, UInt32.
// Rules followed: SURVEY.md App. A.1-A.4, A.10 (OptimizationGoal::Constraints:
// linear combinations are inlined, never witnesses).
// Parity vs real arkworks witness order: UNPINNED (no rustc / crates in this image).
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "fp.h"

namespace orc {

typedef uint32_t Var;  // the variable's COLUMN in the matrices (ark-relations: One, then the instance variables, then the witnesses):
                       // 0 = the constant One; 1 .. n_inst - 1 = public inputs in allocation order; n_inst + k = witness k
struct Term {
    Var v;
    Fp c;
};
typedef std::vector<Term> LCv;
typedef std::shared_ptr<const LCv> LC;

struct CS {
    bool record = false;  // keep A,B,C linear combinations (matrix emission / satisfiability check)
    std::vector<Fp> wit;  // witness_assignment
    // instance_assignment = [1, inst...]. n_inst (1 + the number of AllocationMode::Input field elements of the circuit SHAPE) is set before
    // synthesis: a witness's column depends on it, and witnesses are allocated before the last input is (constraints.rs:341 precedes :353)
    uint32_t n_inst = 1;
    std::vector<Fp> inst;
    std::vector<LC> A, B, C;
    uint64_t ncons = 0;
    std::vector<std::pair<std::string, uint64_t>> marks;
    Var new_witness(const Fp& v) {
        wit.push_back(v);
        return (Var)(n_inst - 1 + wit.size());
    }
    Var new_input(const Fp& v) {  // FpVar::new_input
        inst.push_back(v);
        if (inst.size() >= n_inst) abort();  // the shape announced fewer inputs
        return (Var)inst.size();
    }
    void enforce(const LC& a, const LC& b, const LC& c) {
        ncons++;
        if (record) {
            A.push_back(a);
            B.push_back(b);
            C.push_back(c);
        }
    }
    void mark(const std::string& name) { marks.push_back({name, (uint64_t)wit.size()}); }
};

inline CS*& cur_cs() {
    static thread_local CS* p = nullptr;
    return p;
}
#define CSREF (*orc::cur_cs())

// ---- linear combinations (only materialised when recording)
inline bool recording() { return cur_cs() && cur_cs()->record; }
inline LC lc_zero() { return recording() ? std::make_shared<const LCv>() : nullptr; }
inline LC lc_var(Var v, const Fp& c) {
    if (!recording()) return nullptr;
    auto p = std::make_shared<LCv>();
    if (!fp_is_zero(c)) p->push_back({v, c});
    return p;
}
inline LC lc_var(Var v) { return lc_var(v, fp_one()); }
inline LC lc_const(const Fp& c) { return lc_var(0, c); }
inline LC lc_axpy(const LC& a, const LC& b, const Fp& s) {  // a + s*b, both sorted by var
    if (!recording()) return nullptr;
    auto p = std::make_shared<LCv>();
    const LCv &x = *a, &y = *b;
    p->reserve(x.size() + y.size());
    size_t i = 0, j = 0;
    while (i < x.size() || j < y.size()) {
        if (j >= y.size() || (i < x.size() && x[i].v < y[j].v))
            p->push_back(x[i++]);
        else if (i >= x.size() || y[j].v < x[i].v) {
            Fp c = fp_mul(y[j].c, s);
            if (!fp_is_zero(c)) p->push_back({y[j].v, c});
            j++;
        } else {
            Fp c = fp_add(x[i].c, fp_mul(y[j].c, s));
            if (!fp_is_zero(c)) p->push_back({x[i].v, c});
            i++;
            j++;
        }
    }
    return p;
}
inline LC lc_add(const LC& a, const LC& b) { return lc_axpy(a, b, fp_one()); }
inline LC lc_sub(const LC& a, const LC& b) { return lc_axpy(a, b, fp_neg(fp_one())); }
inline LC lc_scale(const LC& a, const Fp& s) { return lc_axpy(lc_zero(), a, s); }
inline Fp lc_eval(const LCv& a, const std::vector<Fp>& wit, const std::vector<Fp>* inst = nullptr) {
    const uint32_t n_inst = inst ? (uint32_t)inst->size() + 1 : 1;
    Fp acc = fp_zero();
    for (auto& t : a) acc = fp_add(acc, fp_mul(t.c, t.v == 0 ? fp_one() : (t.v < n_inst ? (*inst)[t.v - 1] : wit[t.v - n_inst])));
    return acc;
}

// ------------------------------------------------------------------ Boolean  [ark-r1cs-std bits/boolean.rs]
struct Bool {
    uint8_t kind;  // 0 Constant, 1 Is, 2 Not
    bool val;      // the boolean's VALUE (negation already applied)
    Var var;
    bool is_const() const { return kind == 0; }
};
inline Bool bconst(bool v) { return {0, v, 0}; }
inline LC blc(const Bool& b) {
    if (!recording()) return nullptr;
    if (b.kind == 0) return b.val ? lc_const(fp_one()) : lc_zero();
    if (b.kind == 1) return lc_var(b.var);
    return lc_sub(lc_const(fp_one()), lc_var(b.var));
}
inline Fp fp_of_bool(bool b) { return b ? fp_one() : fp_zero(); }
// AllocatedBool::new_witness: witness + booleanity constraint (1-b)*b = 0
inline Bool balloc(bool v) {
    Var x = CSREF.new_witness(fp_of_bool(v));
    CSREF.enforce(lc_sub(lc_const(fp_one()), lc_var(x)), lc_var(x), lc_zero());
    return {1, v, x};
}
// result allocations of and/xor/or carry no booleanity constraint
inline Bool balloc_nocheck(bool v) {
    Var x = CSREF.new_witness(fp_of_bool(v));
    return {1, v, x};
}
inline Bool bnot(const Bool& a) {
    if (a.kind == 0) return bconst(!a.val);
    return {(uint8_t)(a.kind == 1 ? 2 : 1), !a.val, a.var};
}
inline bool bvarval(const Bool& a) { return a.kind == 2 ? !a.val : a.val; }  // value of the underlying variable
inline Bool bxor(const Bool& a, const Bool& b) {
    if (a.kind == 0) return a.val ? bnot(b) : b;
    if (b.kind == 0) return b.val ? bnot(a) : a;
    // both allocated: witness = xor of the underlying variables; (2a)*b = a + b - c
    bool va = bvarval(a), vb = bvarval(b);
    Bool r = balloc_nocheck(va ^ vb);
    CSREF.enforce(recording() ? lc_scale(lc_var(a.var), fp_from_u64(2)) : nullptr, lc_var(b.var),
                  lc_sub(lc_add(lc_var(a.var), lc_var(b.var)), lc_var(r.var)));
    if (a.kind != b.kind) return bnot(r);
    return r;
}
inline Bool band(const Bool& a, const Bool& b) {
    if (a.kind == 0) return a.val ? b : bconst(false);
    if (b.kind == 0) return b.val ? a : bconst(false);
    Bool r = balloc_nocheck(a.val && b.val);
    // Is&Is: a*b=c ; Is&Not: a*(1-b)=c ; Not&Not: (1-a)(1-b)=c
    CSREF.enforce(blc(a), blc(b), lc_var(r.var));
    return r;
}
inline Bool bor(const Bool& a, const Bool& b) {
    if (a.kind == 0) return a.val ? bconst(true) : b;
    if (b.kind == 0) return b.val ? bconst(true) : a;
    if (a.kind == 1 && b.kind == 1) {
        Bool r = balloc_nocheck(a.val || b.val);
        CSREF.enforce(blc(bnot(a)), blc(bnot(b)), blc(bnot(r)));
        return r;
    }
    return bnot(band(bnot(a), bnot(b)));
}
inline Bool bis_eq(const Bool& a, const Bool& b) { return bnot(bxor(a, b)); }
inline Bool kary_and(const std::vector<Bool>& bits) {
    Bool cur = bits[0];
    for (size_t i = 1; i < bits.size(); i++) cur = band(cur, bits[i]);
    return cur;
}
inline void enforce_kary_nand(const std::vector<Bool>& bits) {
    Bool r = bnot(kary_and(bits));
    if (r.kind == 0) return;
    CSREF.enforce(blc(r), lc_const(fp_one()), lc_const(fp_one()));
}
// Boolean::enforce_equal(other const)  ->  constraint only
inline void benforce_equal_const(const Bool& a, bool c) {
    if (a.kind == 0) return;
    // (a - c) * 1 = 0
    CSREF.enforce(lc_sub(blc(a), blc(bconst(c))), lc_const(fp_one()), lc_zero());
}
// Boolean::enforce_not_equal(Constant(true)):  (1 - a) * 1 = 1
inline void benforce_not_equal_const_true(const Bool& a) {
    if (a.kind == 0) return;
    CSREF.enforce(lc_sub(lc_const(fp_one()), blc(a)), lc_const(fp_one()), lc_const(fp_one()));
}

// ------------------------------------------------------------------ FpVar  [ark-r1cs-std fields/fp/mod.rs]
struct FpVar {
    bool konst;
    Fp v;
    LC lc;  // linear combination (recording mode only); for constants: unused
};
inline FpVar fconst(const Fp& v) { return {true, v, nullptr}; }
inline FpVar fwitness(const Fp& v) {
    Var x = CSREF.new_witness(v);
    return {false, v, lc_var(x)};
}
inline FpVar finput(const Fp& v) {
    Var x = CSREF.new_input(v);
    return {false, v, lc_var(x)};
}
inline LC flc(const FpVar& a) { return a.konst ? lc_const(a.v) : a.lc; }
inline FpVar fadd(const FpVar& a, const FpVar& b) {
    if (a.konst && b.konst) return fconst(fp_add(a.v, b.v));
    return {false, fp_add(a.v, b.v), lc_add(flc(a), flc(b))};
}
inline FpVar fsub(const FpVar& a, const FpVar& b) {
    if (a.konst && b.konst) return fconst(fp_sub(a.v, b.v));
    return {false, fp_sub(a.v, b.v), lc_sub(flc(a), flc(b))};
}
inline FpVar fneg(const FpVar& a) {
    if (a.konst) return fconst(fp_neg(a.v));
    return {false, fp_neg(a.v), lc_scale(a.lc, fp_neg(fp_one()))};
}
inline FpVar fdbl(const FpVar& a) { return fadd(a, a); }
inline FpVar fmulc(const FpVar& a, const Fp& c) {
    if (a.konst) return fconst(fp_mul(a.v, c));
    return {false, fp_mul(a.v, c), lc_scale(a.lc, c)};
}
inline FpVar fmul(const FpVar& a, const FpVar& b) {
    if (a.konst && b.konst) return fconst(fp_mul(a.v, b.v));
    if (a.konst) return fmulc(b, a.v);
    if (b.konst) return fmulc(a, b.v);
    Fp pv = fp_mul(a.v, b.v);
    Var x = CSREF.new_witness(pv);
    LC pl = lc_var(x);
    CSREF.enforce(a.lc, b.lc, pl);
    return {false, pv, pl};
}
inline FpVar fsqr(const FpVar& a) { return fmul(a, a); }
inline FpVar finv(const FpVar& a) {
    if (a.konst) return fconst(fp_inv(a.v));
    Fp iv = fp_inv(a.v);
    Var x = CSREF.new_witness(iv);
    LC il = lc_var(x);
    CSREF.enforce(a.lc, il, lc_const(fp_one()));
    return {false, iv, il};
}
inline void fenforce_equal(const FpVar& a, const FpVar& b) {
    if (a.konst && b.konst) return;
    CSREF.enforce(lc_sub(flc(a), flc(b)), lc_const(fp_one()), lc_zero());
}
inline void fmul_equals(const FpVar& a, const FpVar& b, const FpVar& c) {
    if (a.konst && b.konst && c.konst) return;
    if (a.konst || b.konst) {
        fenforce_equal(c, fmul(a, b));
        return;
    }
    CSREF.enforce(a.lc, b.lc, flc(c));
}
inline FpVar ffrom_bool(const Bool& b) {
    if (b.kind == 0) return fconst(fp_of_bool(b.val));
    return {false, fp_of_bool(b.val), blc(b)};
}
// AllocatedFp::is_neq(self, other) — witness order: boolean, then multiplier
inline Bool falloc_is_neq(const FpVar& self, const FpVar& other) {
    bool ne = !fp_eq(self.v, other.v);
    Bool is_ne = balloc(ne);
    Fp diff = fp_sub(self.v, other.v);
    Var m = CSREF.new_witness(ne ? fp_inv(diff) : fp_one());
    LC d = lc_sub(flc(self), flc(other));
    CSREF.enforce(d, lc_var(m), blc(is_ne));
    CSREF.enforce(d, blc(bnot(is_ne)), lc_zero());
    return is_ne;
}
inline Bool fis_eq(const FpVar& a, const FpVar& b) {
    if (a.konst && b.konst) return bconst(fp_eq(a.v, b.v));
    if (a.konst) return bnot(falloc_is_neq(a, b));  // c.is_eq(v)
    if (b.konst) return bnot(falloc_is_neq(b, a));  // (Var v, Constant c) => c.is_eq(v)
    return bnot(falloc_is_neq(a, b));
}
inline FpVar fselect(const Bool& cond, const FpVar& t, const FpVar& f) {
    if (cond.kind == 0) return cond.val ? t : f;
    if (t.konst && f.konst) {
        // is*t + not*f  (no witness)
        FpVar is = ffrom_bool(cond), nt = ffrom_bool(bnot(cond));
        return fadd(fmulc(is, t.v), fmulc(nt, f.v));
    }
    Fp rv = cond.val ? t.v : f.v;
    Var x = CSREF.new_witness(rv);
    LC rl = lc_var(x);
    CSREF.enforce(blc(cond), lc_sub(flc(t), flc(f)), lc_sub(rl, flc(f)));
    return {false, rv, rl};
}

// Boolean::enforce_in_field_le / enforce_smaller_or_equal_than_le with element = p-1
inline void enforce_in_field_le(const std::vector<Bool>& bits) {
    uint64_t b[6];
    memcpy(b, P_LIMBS, 48);
    b[0] -= 1;
    int nb = 381;
    assert((int)bits.size() == nb);
    Bool last_run = bconst(true);
    std::vector<Bool> current_run;
    for (int i = nb - 1; i >= 0; i--) {  // big-endian walk
        bool eb = (b[i / 64] >> (i % 64)) & 1;
        const Bool& a = bits[i];
        if (eb) {
            current_run.push_back(a);
        } else {
            if (!current_run.empty()) {
                current_run.push_back(last_run);
                last_run = kary_and(current_run);
                current_run.clear();
            }
            enforce_kary_nand({last_run, a});
        }
    }
    assert(current_run.empty());
}
// FpVar::to_bits_le
inline std::vector<Bool> fto_bits_le(const FpVar& a) {
    uint64_t raw[6];
    fp_to_raw(raw, a.v);
    std::vector<Bool> bits(381);
    if (a.konst) {
        for (int i = 0; i < 381; i++) bits[i] = bconst((raw[i / 64] >> (i % 64)) & 1);
        return bits;
    }
    for (int i = 0; i < 381; i++) bits[i] = balloc((raw[i / 64] >> (i % 64)) & 1);
    if (recording()) {
        auto p = std::make_shared<LCv>();
        Fp coeff = fp_one();
        for (int i = 0; i < 381; i++) {
            p->push_back({bits[i].var, coeff});
            coeff = fp_dbl(coeff);
        }
        CSREF.enforce(lc_zero(), lc_zero(), lc_sub(p, a.lc));
    } else
        CSREF.enforce(nullptr, nullptr, nullptr);
    enforce_in_field_le(bits);
    return bits;
}

// ------------------------------------------------------------------ UInt8 / UInt32  [bits/uint8.rs, bits/uint.rs]
struct U8 {
    Bool b[8];  // little-endian
    uint8_t value() const {
        uint8_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint8_t)(b[i].val ? 1 : 0) << i;
        return v;
    }
};
inline U8 u8const(uint8_t v) {
    U8 r;
    for (int i = 0; i < 8; i++) r.b[i] = bconst((v >> i) & 1);
    return r;
}
inline U8 u8witness(uint8_t v) {
    U8 r;
    for (int i = 0; i < 8; i++) r.b[i] = balloc((v >> i) & 1);
    return r;
}
inline std::vector<U8> u8const_vec(const uint8_t* p, size_t n) {
    std::vector<U8> r(n);
    for (size_t i = 0; i < n; i++) r[i] = u8const(p[i]);
    return r;
}
inline std::vector<U8> u8witness_vec(const uint8_t* p, size_t n) {
    std::vector<U8> r(n);
    for (size_t i = 0; i < n; i++) r[i] = u8witness(p[i]);
    return r;
}
inline U8 u8xor(const U8& a, const U8& b) {
    U8 r;
    for (int i = 0; i < 8; i++) r.b[i] = bxor(a.b[i], b.b[i]);
    return r;
}
// [UInt8]::to_constraint_field for a chunk of <= 47 bytes (little-endian bytes): LC only, no witness
inline FpVar le_bytes_to_fp_var(const U8* bytes, size_t n) {
    bool all_const = true;
    for (size_t i = 0; i < n; i++)
        for (int j = 0; j < 8; j++) all_const = all_const && bytes[i].b[j].is_const();
    // value = from_le_bytes_mod_order
    std::vector<uint8_t> be(n);
    for (size_t i = 0; i < n; i++) be[n - 1 - i] = bytes[i].value();
    Fp val = fp_from_be_bytes_mod_order(be.data(), n);
    if (all_const) return fconst(val);
    LC lc = nullptr;
    if (recording()) {
        LC acc = lc_zero();
        Fp coeff = fp_one();
        for (size_t i = 0; i < n; i++)
            for (int j = 0; j < 8; j++) {
                acc = lc_axpy(acc, blc(bytes[i].b[j]), coeff);
                coeff = fp_dbl(coeff);
            }
        lc = acc;
    }
    return {false, val, lc};
}

struct U32 {
    Bool b[32];  // little-endian
    uint32_t value() const {
        uint32_t v = 0;
        for (int i = 0; i < 32; i++) v |= (uint32_t)(b[i].val ? 1 : 0) << i;
        return v;
    }
    bool is_const() const {
        for (int i = 0; i < 32; i++)
            if (!b[i].is_const()) return false;
        return true;
    }
};
inline U32 u32const(uint32_t v) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = bconst((v >> i) & 1);
    return r;
}
inline U32 u32rotr(const U32& a, int by) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = a.b[(i + by) % 32];
    return r;
}
inline U32 u32shr(const U32& a, int by) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = (i + by < 32) ? a.b[i + by] : bconst(false);
    return r;
}
inline U32 u32xor(const U32& a, const U32& b) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = bxor(a.b[i], b.b[i]);
    return r;
}
inline U32 u32and(const U32& a, const U32& b) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = band(a.b[i], b.b[i]);
    return r;
}
inline U32 u32not(const U32& a) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = bnot(a.b[i]);
    return r;
}
inline U32 u32from_bytes_be(const U8* bytes) {
    U32 r;
    for (int k = 0; k < 4; k++)  // bytes.iter().rev(): last byte is least significant
        for (int j = 0; j < 8; j++) r.b[k * 8 + j] = bytes[3 - k].b[j];
    return r;
}
inline void u32to_bytes_be(const U32& a, U8* out) {
    for (int k = 0; k < 4; k++)
        for (int j = 0; j < 8; j++) out[3 - k].b[j] = a.b[k * 8 + j];
}
// UInt32::addmany
inline U32 u32addmany(const std::vector<U32>& ops) {
    if (ops.size() == 1) return ops[0];
    uint64_t sum = 0;
    bool all_const = true;
    for (auto& op : ops) {
        sum += op.value();
        all_const = all_const && op.is_const();
    }
    if (all_const) return u32const((uint32_t)sum);
    // number of result bits = bit length of (2^32-1)*k
    uint64_t maxv = 0xffffffffULL * ops.size();
    int nbits = 0;
    while (maxv) {
        nbits++;
        maxv >>= 1;
    }
    LC lc = nullptr;
    if (recording()) {
        LC acc = lc_zero();
        for (auto& op : ops) {
            Fp coeff = fp_one();
            for (int i = 0; i < 32; i++) {
                if (!(op.b[i].kind == 0 && !op.b[i].val)) acc = lc_axpy(acc, blc(op.b[i]), coeff);
                coeff = fp_dbl(coeff);
            }
        }
        lc = acc;
    }
    U32 r;
    Fp coeff = fp_one();
    for (int i = 0; i < nbits; i++) {
        Bool bit = balloc((sum >> i) & 1);
        if (recording()) lc = lc_axpy(lc, lc_var(bit.var), fp_neg(coeff));
        if (i < 32) r.b[i] = bit;
        coeff = fp_dbl(coeff);
    }
    CSREF.enforce(lc_zero(), lc_zero(), lc);
    return r;
}

}  // namespace orc
