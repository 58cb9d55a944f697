// Constraint-matrix emitter of libblsw.so (host only): the R1CS of the circuits whose witness vectors the HIP kernels fill.
//
// The reference's consumer reads the constraint system after synthesis (cs.to_matrices(); src/constraints.rs:369-373 prints
// its size). This file synthesises the same circuits SYMBOLICALLY — no witness values, only which variable is which linear
// combination — and writes A, B, C in arkworks' ConstraintMatrices shape (one sparse row per constraint, column 0 = the
// constant one, column k = witness k - 1), so that an arkworks prover can take (matrices, [1] ++ witness) without running
// synthesis at all. Circuits: single key (src/constraints.rs:335-366 + :90-128), aggregate_verify (:153-191, :378-441),
// N+1-pair product (blsw_verify_multi_batch).
//
// What it encodes is the allocation / constraint discipline of ark-r1cs-std ^0.4.0 and ark-crypto-primitives ^0.4.0 under
// OptimizationGoal::Constraints: constants fold, constant x variable and sums are linear combinations,
// every variable x variable product, inverse, is_eq, select and boolean operation allocates and constrains. It shares no
// code with oracle/ (the test oracle has its own value-carrying restatement; tests compare the two row by row and check
// A z o B z = C z on GPU-produced witnesses). Constants come from constants.hpp / sha.hpp like the kernels'.
#include <string.h>
#include <algorithm>
#include <mutex>
#include <vector>
#include "../../include/blsw.h"
#include "constants.hpp"
#include "layout.h"

namespace blsw {
namespace r1cs {

// ------------------------------------------------------------------------------------------------ linear combinations
struct Term {
    uint32_t v;  // 0 = the constant one, k = witness k - 1
    Fp c;
};
typedef std::vector<Term> Lc;  // sorted by v, no zero coefficients

static Fp f_neg_one() { return fp_neg(fp_one()); }
static Lc lc_axpy(const Lc& a, const Lc& b, const Fp& s) {  // a + s * b
    Lc r;
    r.reserve(a.size() + b.size());
    size_t i = 0, j = 0;
    while (i < a.size() || j < b.size()) {
        if (j >= b.size() || (i < a.size() && a[i].v < b[j].v))
            r.push_back(a[i++]);
        else if (i >= a.size() || b[j].v < a[i].v) {
            Fp c = fp_mul(b[j].c, s);
            if (!fp_is_zero(c)) r.push_back({b[j].v, c});
            j++;
        } else {
            Fp c = fp_add(a[i].c, fp_mul(b[j].c, s));
            if (!fp_is_zero(c)) r.push_back({a[i].v, c});
            i++;
            j++;
        }
    }
    return r;
}
static Lc lc_of(uint32_t v, const Fp& c) {
    Lc r;
    if (!fp_is_zero(c)) r.push_back({v, c});
    return r;
}
static Lc lc_var(uint32_t v) { return lc_of(v, fp_one()); }
static Lc lc_const(const Fp& c) { return lc_of(0, c); }
static Lc lc_add(const Lc& a, const Lc& b) { return lc_axpy(a, b, fp_one()); }
static Lc lc_sub(const Lc& a, const Lc& b) { return lc_axpy(a, b, f_neg_one()); }
static Lc lc_scale(const Lc& a, const Fp& s) { return lc_axpy(Lc(), a, s); }

// ------------------------------------------------------------------------------------------------ the system being written
struct Sys {
    uint64_t n_cons = 0;
    uint32_t n_wit = 0;
    // a variable's number is its COLUMN (ark-relations: one, the instance variables, then the witnesses): 0 = one, 1 .. n_inst - 1 = the public inputs
    // in allocation order, n_inst + k = witness k. n_inst is a property of the circuit shape, set before the synthesis.
    uint32_t n_inst = 1, n_in = 0;
    std::vector<uint64_t> row_ptr[3];
    std::vector<uint32_t> col[3];
    std::vector<Fp> val[3];
    uint32_t alloc() { return n_inst - 1 + ++n_wit; }
    uint32_t alloc_input() { return ++n_in; }  // FpVar::new_input
    void row(int m, const Lc& l) {
        row_ptr[m].push_back(col[m].size());
        for (const Term& t : l) {
            col[m].push_back(t.v);
            val[m].push_back(t.c);
        }
    }
    void enforce(const Lc& a, const Lc& b, const Lc& c) {
        row(0, a);
        row(1, b);
        row(2, c);
        n_cons++;
    }
    void finish() {
        for (int m = 0; m < 3; m++) row_ptr[m].push_back(col[m].size());
    }
};
static thread_local Sys* S = nullptr;

// ------------------------------------------------------------------------------------------------ Boolean
struct B {
    uint8_t kind;  // 0 constant, 1 Is(var), 2 Not(var)
    bool cv;       // value of a constant
    uint32_t var;
    bool konst() const { return kind == 0; }
};
static B b_const(bool v) { return {0, v, 0}; }
static Lc b_lc(const B& b) {
    if (b.kind == 0) return b.cv ? lc_const(fp_one()) : Lc();
    if (b.kind == 1) return lc_var(b.var);
    return lc_sub(lc_const(fp_one()), lc_var(b.var));
}
static B b_alloc() {  // AllocatedBool::new_witness: (1 - b) * b = 0
    uint32_t x = S->alloc();
    S->enforce(lc_sub(lc_const(fp_one()), lc_var(x)), lc_var(x), Lc());
    return {1, false, x};
}
static B b_not(const B& a) {
    if (a.kind == 0) return b_const(!a.cv);
    return {(uint8_t)(a.kind == 1 ? 2 : 1), false, a.var};
}
static B b_xor(const B& a, const B& b) {
    if (a.kind == 0) return a.cv ? b_not(b) : b;
    if (b.kind == 0) return b.cv ? b_not(a) : a;
    uint32_t r = S->alloc();  // xor of the underlying variables: (2a) * b = a + b - c
    S->enforce(lc_of(a.var, fp_dbl(fp_one())), lc_var(b.var), lc_sub(lc_add(lc_var(a.var), lc_var(b.var)), lc_var(r)));
    B res = {1, false, r};
    return a.kind != b.kind ? b_not(res) : res;
}
static B b_and(const B& a, const B& b) {
    if (a.kind == 0) return a.cv ? b : b_const(false);
    if (b.kind == 0) return b.cv ? a : b_const(false);
    uint32_t r = S->alloc();
    S->enforce(b_lc(a), b_lc(b), lc_var(r));
    return {1, false, r};
}
static B b_or(const B& a, const B& b) {
    if (a.kind == 0) return a.cv ? b_const(true) : b;
    if (b.kind == 0) return b.cv ? b_const(true) : a;
    if (a.kind == 1 && b.kind == 1) {
        uint32_t r = S->alloc();
        B res = {1, false, r};
        S->enforce(b_lc(b_not(a)), b_lc(b_not(b)), b_lc(b_not(res)));
        return res;
    }
    return b_not(b_and(b_not(a), b_not(b)));
}
static B b_kary_and(const std::vector<B>& bits) {
    B cur = bits[0];
    for (size_t i = 1; i < bits.size(); i++) cur = b_and(cur, bits[i]);
    return cur;
}
static void b_enforce_nand(const std::vector<B>& bits) {
    B r = b_not(b_kary_and(bits));
    if (r.kind == 0) return;
    S->enforce(b_lc(r), lc_const(fp_one()), lc_const(fp_one()));
}
static void b_enforce_equal_const(const B& a, bool c) {  // (a - c) * 1 = 0
    if (a.kind == 0) return;
    S->enforce(lc_sub(b_lc(a), b_lc(b_const(c))), lc_const(fp_one()), Lc());
}
static void b_enforce_not_true(const B& a) {  // Boolean::enforce_not_equal(TRUE): (1 - a) * 1 = 1
    if (a.kind == 0) return;
    S->enforce(lc_sub(lc_const(fp_one()), b_lc(a)), lc_const(fp_one()), lc_const(fp_one()));
}

// ------------------------------------------------------------------------------------------------ FpVar
struct V {
    bool k;  // constant
    Fp c;    // its value
    Lc l;    // the linear combination of a variable
};
static V v_const(const Fp& c) { return {true, c, Lc()}; }
static V v_zero() { return v_const(fp_zero()); }
static V v_one() { return v_const(fp_one()); }
static V v_alloc() { return {false, fp_zero(), lc_var(S->alloc())}; }
static V v_input() { return {false, fp_zero(), lc_var(S->alloc_input())}; }  // FpVar::new_input
static Lc v_lc(const V& a) { return a.k ? lc_const(a.c) : a.l; }
static V v_add(const V& a, const V& b) {
    if (a.k && b.k) return v_const(fp_add(a.c, b.c));
    return {false, fp_zero(), lc_add(v_lc(a), v_lc(b))};
}
static V v_sub(const V& a, const V& b) {
    if (a.k && b.k) return v_const(fp_sub(a.c, b.c));
    return {false, fp_zero(), lc_sub(v_lc(a), v_lc(b))};
}
static V v_neg(const V& a) {
    if (a.k) return v_const(fp_neg(a.c));
    return {false, fp_zero(), lc_scale(a.l, f_neg_one())};
}
static V v_dbl(const V& a) { return v_add(a, a); }
static V v_scale(const V& a, const Fp& c) {
    if (a.k) return v_const(fp_mul(a.c, c));
    return {false, fp_zero(), lc_scale(a.l, c)};
}
static V v_mul(const V& a, const V& b) {
    if (a.k && b.k) return v_const(fp_mul(a.c, b.c));
    if (a.k) return v_scale(b, a.c);
    if (b.k) return v_scale(a, b.c);
    V r = v_alloc();
    S->enforce(a.l, b.l, r.l);
    return r;
}
static void v_enforce_equal(const V& a, const V& b) {
    if (a.k && b.k) return;
    S->enforce(lc_sub(v_lc(a), v_lc(b)), lc_const(fp_one()), Lc());
}
static void v_mul_equals(const V& a, const V& b, const V& c) {
    if (a.k && b.k && c.k) return;
    if (a.k || b.k) {
        v_enforce_equal(c, v_mul(a, b));
        return;
    }
    S->enforce(a.l, b.l, v_lc(c));
}
static V v_from_bool(const B& b) {
    if (b.kind == 0) return v_const(b.cv ? fp_one() : fp_zero());
    return {false, fp_zero(), b_lc(b)};
}
// AllocatedFp::is_neq: boolean, multiplier; (self - other) * multiplier = ne; (self - other) * (1 - ne) = 0
static B v_alloc_is_neq(const V& self, const V& other) {
    B ne = b_alloc();
    uint32_t m = S->alloc();
    Lc d = lc_sub(v_lc(self), v_lc(other));
    S->enforce(d, lc_var(m), b_lc(ne));
    S->enforce(d, b_lc(b_not(ne)), Lc());
    return ne;
}
static B v_is_eq(const V& a, const V& b) {
    if (a.k && b.k) return b_const(fp_eq(a.c, b.c));
    if (b.k) return b_not(v_alloc_is_neq(b, a));  // (Var, Constant) is evaluated as constant.is_eq(var)
    return b_not(v_alloc_is_neq(a, b));
}
static V v_select(const B& cond, const V& t, const V& f) {
    if (cond.kind == 0) return cond.cv ? t : f;
    if (t.k && f.k) return v_add(v_scale(v_from_bool(cond), t.c), v_scale(v_from_bool(b_not(cond)), f.c));
    V r = v_alloc();
    S->enforce(b_lc(cond), lc_sub(v_lc(t), v_lc(f)), lc_sub(r.l, v_lc(f)));
    return r;
}
// Boolean::enforce_in_field_le against p - 1, bits little-endian
static void enforce_in_field_le(const std::vector<B>& bits) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    B last_run = b_const(true);
    std::vector<B> run;
    for (int i = 380; i >= 0; i--) {
        uint32_t w = P[i >> 5];
        if (i < 32) w -= 1;
        const bool eb = (w >> (i & 31)) & 1;
        if (eb) {
            run.push_back(bits[i]);
        } else {
            if (!run.empty()) {
                run.push_back(last_run);
                last_run = b_kary_and(run);
                run.clear();
            }
            b_enforce_nand({last_run, bits[i]});
        }
    }
}
static std::vector<B> v_to_bits_le(const V& a) {
    std::vector<B> bits(381);
    if (a.k) {
        Fp c = fp_to_canonical(a.c);
        for (int i = 0; i < 381; i++) bits[i] = b_const((c.l[i >> 5] >> (i & 31)) & 1);
        return bits;
    }
    for (int i = 0; i < 381; i++) bits[i] = b_alloc();
    Lc p;
    Fp coeff = fp_one();
    for (int i = 0; i < 381; i++) {
        p.push_back({bits[i].var, coeff});  // freshly allocated: ascending variable numbers
        coeff = fp_dbl(coeff);
    }
    S->enforce(Lc(), Lc(), lc_sub(p, a.l));
    enforce_in_field_le(bits);
    return bits;
}

// ------------------------------------------------------------------------------------------------ Fp2Var
struct V2 {
    V c0, c1;
    bool k() const { return c0.k && c1.k; }
};
static V2 v2_const(const Fp2& c) { return {v_const(c.c0), v_const(c.c1)}; }
static V2 v2_zero() { return v2_const(fp2_zero()); }
static V2 v2_one() { return v2_const(fp2_one()); }
static V2 v2_alloc() {
    V a = v_alloc();
    V b = v_alloc();
    return {a, b};
}
static V2 v2_input() {
    V a = v_input();
    V b = v_input();
    return {a, b};
}
static V2 v2_add(const V2& a, const V2& b) { return {v_add(a.c0, b.c0), v_add(a.c1, b.c1)}; }
static V2 v2_sub(const V2& a, const V2& b) { return {v_sub(a.c0, b.c0), v_sub(a.c1, b.c1)}; }
static V2 v2_neg(const V2& a) { return {v_neg(a.c0), v_neg(a.c1)}; }
static V2 v2_dbl(const V2& a) { return {v_dbl(a.c0), v_dbl(a.c1)}; }
static V2 v2_conj(const V2& a) { return {a.c0, v_neg(a.c1)}; }
static V2 v2_mul_xi(const V2& a) { return {v_sub(a.c0, a.c1), v_add(a.c0, a.c1)}; }
static V2 v2_mul(const V2& a, const V2& b) {  // Karatsuba: a0 b0, a1 b1, (a0 + a1)(b0 + b1)
    V v0 = v_mul(a.c0, b.c0);
    V v1 = v_mul(a.c1, b.c1);
    V s = v_mul(v_add(a.c1, a.c0), v_add(b.c0, b.c1));
    return {v_sub(v0, v1), v_sub(v_sub(s, v0), v1)};
}
static V2 v2_sqr(const V2& a) {  // complex squaring: c0 c1, (c0 - c1)(c0 + c1)
    V d = v_sub(a.c0, a.c1), s = v_add(a.c0, a.c1);
    V v2 = v_mul(a.c0, a.c1);
    V t = v_mul(d, s);
    t = v_add(t, v2);
    return {v_sub(t, v2), v_dbl(v2)};
}
static V2 v2_mulc(const V2& a, const Fp2& c) { return v2_mul(a, v2_const(c)); }
static V2 v2_scale_fp(const V2& a, const Fp& c) { return {v_scale(a.c0, c), v_scale(a.c1, c)}; }
static void v2_mul_equals(const V2& a, const V2& b, const V2& r) {  // QuadExtVar::mul_equals: one product witness, two checks
    V v1 = v_mul(a.c1, b.c1);
    V nr_v1 = v_neg(v1);
    v_mul_equals(a.c0, b.c0, v_sub(r.c0, nr_v1));
    V a01 = v_add(a.c0, a.c1), b01 = v_add(b.c0, b.c1);
    V tmp = v_add(v_add(v_sub(v1, nr_v1), r.c1), r.c0);
    v_mul_equals(a01, b01, tmp);
}
static V2 v2_inv(const V2& a, const Fp2* const_value = nullptr) {
    if (a.k()) return v2_const(const_value ? *const_value : fp2_inv({a.c0.c, a.c1.c}));
    V2 inv = v2_alloc();
    v2_mul_equals(a, inv, v2_one());
    return inv;
}
static V2 v2_div_unchecked(const V2& self, const V2& d) {  // FieldVar::mul_by_inverse_unchecked
    if (self.k() && d.k()) return v2_const(fp2_mul({self.c0.c, self.c1.c}, fp2_inv({d.c0.c, d.c1.c})));
    V2 r = v2_alloc();
    v2_mul_equals(r, d, self);
    return r;
}
static B v2_is_eq(const V2& a, const V2& b) {
    B b0 = v_is_eq(a.c0, b.c0);
    B b1 = v_is_eq(a.c1, b.c1);
    return b_and(b0, b1);
}
static B v2_is_zero(const V2& a) { return v2_is_eq(a, v2_zero()); }
static V2 v2_select(const B& c, const V2& t, const V2& f) {
    V x = v_select(c, t.c0, f.c0);
    V y = v_select(c, t.c1, f.c1);
    return {x, y};
}
static V2 v2_from_bool(const B& b) { return {v_from_bool(b), v_zero()}; }
static V2 v2_frob(const V2& a, int power) { return (power & 1) ? v2_conj(a) : a; }

// ------------------------------------------------------------------------------------------------ Fp6Var, Fp12Var
struct V6 {
    V2 c0, c1, c2;
};
struct V12 {
    V6 c0, c1;
};
static V6 v6_const(const Fp2& a, const Fp2& b, const Fp2& c) { return {v2_const(a), v2_const(b), v2_const(c)}; }
static V6 v6_alloc() {
    V2 a = v2_alloc();
    V2 b = v2_alloc();
    V2 c = v2_alloc();
    return {a, b, c};
}
static V6 v6_add(const V6& a, const V6& b) { return {v2_add(a.c0, b.c0), v2_add(a.c1, b.c1), v2_add(a.c2, b.c2)}; }
static V6 v6_sub(const V6& a, const V6& b) { return {v2_sub(a.c0, b.c0), v2_sub(a.c1, b.c1), v2_sub(a.c2, b.c2)}; }
static V6 v6_neg(const V6& a) { return {v2_neg(a.c0), v2_neg(a.c1), v2_neg(a.c2)}; }
static V6 v6_dbl(const V6& a) { return {v2_dbl(a.c0), v2_dbl(a.c1), v2_dbl(a.c2)}; }
static V6 v6_mul_v(const V6& a) { return {v2_mul_xi(a.c2), a.c0, a.c1}; }
static V6 v6_mul(const V6& a, const V6& b) {
    V2 v0 = v2_mul(a.c0, b.c0);
    V2 v1 = v2_mul(a.c1, b.c1);
    V2 v2 = v2_mul(a.c2, b.c2);
    V2 t0 = v2_mul(v2_add(a.c1, a.c2), v2_add(b.c1, b.c2));
    V2 c0 = v2_add(v2_mul_xi(v2_sub(v2_sub(t0, v1), v2)), v0);
    V2 t1 = v2_mul(v2_add(a.c0, a.c1), v2_add(b.c0, b.c1));
    V2 c1 = v2_add(v2_sub(v2_sub(t1, v0), v1), v2_mul_xi(v2));
    V2 t2 = v2_mul(v2_add(a.c0, a.c2), v2_add(b.c0, b.c2));
    V2 c2 = v2_sub(v2_add(v2_sub(t2, v0), v1), v2);
    return {c0, c1, c2};
}
static void v6_mul_equals(const V6& a, const V6& b, const V6& r) {
    V2 v0 = v2_mul(a.c0, b.c0);
    V2 v1 = v2_mul(a.c1, b.c1);
    V2 v2 = v2_mul(a.c2, b.c2);
    V2 nr_a12 = v2_mul_xi(v2_add(a.c1, a.c2)), b12 = v2_add(b.c1, b.c2);
    V2 nr_v1 = v2_mul_xi(v1), nr_v2 = v2_mul_xi(v2);
    v2_mul_equals(nr_a12, b12, v2_add(v2_add(v2_sub(r.c0, v0), nr_v1), nr_v2));
    v2_mul_equals(v2_add(a.c0, a.c1), v2_add(b.c0, b.c1), v2_add(v2_add(v2_sub(r.c1, nr_v2), v0), v1));
    v2_mul_equals(v2_add(a.c0, a.c2), v2_add(b.c0, b.c2), v2_add(v2_sub(v2_add(r.c2, v0), v1), v2));
}
static V6 v6_mul_by_0_c1_0(const V6& a, const V2& c1) {
    V2 v1 = v2_mul(a.c1, c1);
    V2 a12 = v2_add(a.c1, a.c2), a01 = v2_add(a.c0, a.c1);
    V2 t0 = v2_mul(a12, c1);
    V2 r0 = v2_mul_xi(v2_sub(t0, v1));
    V2 t1 = v2_mul(a01, c1);
    return {r0, v2_sub(t1, v1), v1};
}
static V6 v6_mul_by_c0_c1_0(const V6& a, const V2& c0, const V2& c1) {
    V2 v0 = v2_mul(a.c0, c0);
    V2 v1 = v2_mul(a.c1, c1);
    V2 a12 = v2_add(a.c1, a.c2), a01 = v2_add(a.c0, a.c1), a02 = v2_add(a.c0, a.c2), b01 = v2_add(c0, c1);
    V2 t0 = v2_mul(a12, c1);
    V2 r0 = v2_add(v2_mul_xi(v2_sub(t0, v1)), v0);
    V2 t1 = v2_mul(a01, b01);
    V2 r1 = v2_sub(v2_sub(t1, v0), v1);
    V2 t2 = v2_mul(a02, c0);
    V2 r2 = v2_add(v2_sub(t2, v0), v1);
    return {r0, r1, r2};
}
static B v6_is_eq(const V6& a, const V6& b) {
    B b0 = v2_is_eq(a.c0, b.c0);
    B b1 = v2_is_eq(a.c1, b.c1);
    B b2 = v2_is_eq(a.c2, b.c2);
    B t = b_and(b0, b1);
    return b_and(t, b2);
}
static Fp2 frob6_c1(int p) { return p == 1 ? K_FROB6_C1_1() : (p == 2 ? K_FROB6_C1_2() : K_FROB6_C1_3()); }
static Fp2 frob6_c2(int p) { return p == 1 ? K_FROB6_C2_1() : (p == 2 ? K_FROB6_C2_2() : K_FROB6_C2_3()); }
static Fp2 frob12_c1(int p) { return p == 1 ? K_FROB12_C1_1() : (p == 2 ? K_FROB12_C1_2() : K_FROB12_C1_3()); }
static V6 v6_frob(const V6& a, int p) { return {v2_frob(a.c0, p), v2_mulc(v2_frob(a.c1, p), frob6_c1(p)), v2_mulc(v2_frob(a.c2, p), frob6_c2(p))}; }
static V12 v12_one() { return {v6_const(fp2_one(), fp2_zero(), fp2_zero()), v6_const(fp2_zero(), fp2_zero(), fp2_zero())}; }
static V12 v12_mul(const V12& a, const V12& b) {
    V6 v0 = v6_mul(a.c0, b.c0);
    V6 v1 = v6_mul(a.c1, b.c1);
    V6 s = v6_mul(v6_add(a.c1, a.c0), v6_add(b.c0, b.c1));
    return {v6_add(v0, v6_mul_v(v1)), v6_sub(v6_sub(s, v0), v1)};
}
static V12 v12_sqr(const V12& a) {
    V6 v0 = v6_sub(a.c0, a.c1), v3 = v6_sub(a.c0, v6_mul_v(a.c1));
    V6 v2 = v6_mul(a.c0, a.c1);
    V6 t = v6_mul(v0, v3);
    t = v6_add(t, v2);
    return {v6_add(t, v6_mul_v(v2)), v6_dbl(v2)};
}
static V12 v12_conj(const V12& a) { return {a.c0, v6_neg(a.c1)}; }
static V12 v12_inv(const V12& a) {  // allocate the inverse, QuadExtVar::mul_equals(self, inverse, one)
    V6 i0 = v6_alloc();
    V6 i1 = v6_alloc();
    V12 one = v12_one();
    V6 v1 = v6_mul(a.c1, i1);
    V6 nr_v1 = v6_mul_v(v1);
    v6_mul_equals(a.c0, i0, v6_sub(one.c0, nr_v1));
    v6_mul_equals(v6_add(a.c0, a.c1), v6_add(i0, i1), v6_add(v6_add(v6_sub(v1, nr_v1), one.c1), one.c0));
    return {i0, i1};
}
static V12 v12_frob(const V12& a, int p) {
    V6 c0 = v6_frob(a.c0, p), c1 = v6_frob(a.c1, p);
    const Fp2 k = frob12_c1(p);
    return {c0, {v2_mulc(c1.c0, k), v2_mulc(c1.c1, k), v2_mulc(c1.c2, k)}};
}
static V12 v12_mul_by_014(const V12& f, const V2& c0, const V2& c1, const V2& d1) {
    V6 v0 = v6_mul_by_c0_c1_0(f.c0, c0, c1);
    V6 v1 = v6_mul_by_0_c1_0(f.c1, d1);
    V6 n0 = v6_add(v6_mul_v(v1), v0);
    V6 t = v6_mul_by_c0_c1_0(v6_add(f.c0, f.c1), c0, v2_add(c1, d1));
    return {n0, v6_sub(v6_sub(t, v0), v1)};
}
static void cyc_half(const V2& za, const V2& zb, V2& t_even, V2& t_odd) {
    V2 tmp = v2_mul(za, zb);
    V2 s1 = v2_add(za, zb), s2 = v2_add(v2_mul_xi(zb), za), s4 = v2_add(v2_mul_xi(tmp), tmp);
    V2 prod = v2_mul(s1, s2);
    t_even = v2_sub(prod, s4);
    t_odd = v2_dbl(tmp);
}
static V12 v12_cyclotomic_square(const V12& f) {
    const V2 &z0 = f.c0.c0, &z4 = f.c0.c1, &z3 = f.c0.c2, &z2 = f.c1.c0, &z1 = f.c1.c1, &z5 = f.c1.c2;
    V2 t0, t1, t2, t3, t4, t5;
    cyc_half(z0, z1, t0, t1);
    cyc_half(z2, z3, t2, t3);
    cyc_half(z4, z5, t4, t5);
    V2 xt5 = v2_mul_xi(t5);
    V6 c0 = {v2_add(v2_dbl(v2_sub(t0, z0)), t0), v2_add(v2_dbl(v2_sub(t2, z4)), t2), v2_add(v2_dbl(v2_sub(t4, z3)), t4)};
    V6 c1 = {v2_add(v2_dbl(v2_add(z2, xt5)), xt5), v2_add(v2_dbl(v2_add(t1, z1)), t1), v2_add(v2_dbl(v2_add(t3, z5)), t3)};
    return {c0, c1};
}
static V12 v12_exp_by_x(const V12& f) {  // optimized_cyclotomic_exp over NAF(|x|), then conjugate (x < 0)
    int8_t naf[80];
    int n = 0;
    unsigned __int128 e = BLSW_X_ABS;
    while (e != 0) {
        int8_t z = 0;
        if (e & 1) {
            z = (int8_t)(2 - (int)(e % 4));
            if (z >= 0)
                e -= (unsigned)z;
            else
                e += (unsigned)(-z);
        }
        naf[n++] = z;
        e >>= 1;
    }
    V12 res = v12_one(), f_inv = v12_conj(f);
    bool found = false;
    for (int i = n - 1; i >= 0; i--) {
        if (found) res = v12_cyclotomic_square(res);
        if (naf[i] != 0) {
            found = true;
            res = v12_mul(res, naf[i] > 0 ? f : f_inv);
        }
    }
    return v12_conj(res);
}
static B v12_is_eq(const V12& a, const V12& b) {
    B b0 = v6_is_eq(a.c0, b.c0);
    B b1 = v6_is_eq(a.c1, b.c1);
    return b_and(b0, b1);
}

// ------------------------------------------------------------------------------------------------ curves (ProjectiveVar, a = 0)
// T: field ops of the coordinate field. G1 over V, G2 over V2.
struct T1 {
    typedef V F;
    typedef Fp N;
    static F add(const F& a, const F& b) { return v_add(a, b); }
    static F sub(const F& a, const F& b) { return v_sub(a, b); }
    static F neg(const F& a) { return v_neg(a); }
    static F dbl(const F& a) { return v_dbl(a); }
    static F mul(const F& a, const F& b) { return v_mul(a, b); }
    static F sqr(const F& a) { return v_mul(a, a); }
    static F mul3b(const F& a) { return v_scale(a, K_G1_3B()); }
    static F constant(const N& c) { return v_const(c); }
    static F zero() { return v_zero(); }
    static F one() { return v_one(); }
    static F alloc() { return v_alloc(); }
    static F input() { return v_input(); }
    static bool konst(const F& a) { return a.k; }
    static N value(const F& a) { return a.c; }
    static bool nzero(const N& a) { return fp_is_zero(a); }
    static N ninv(const N& a) { return fp_inv(a); }
    static N nmul(const N& a, const N& b) { return fp_mul(a, b); }
    static B is_eq(const F& a, const F& b) { return v_is_eq(a, b); }
    static F select(const B& c, const F& t, const F& f) { return v_select(c, t, f); }
    static void mul_equals(const F& a, const F& b, const F& r) { v_mul_equals(a, b, r); }
    static F from_bool(const B& b) { return v_from_bool(b); }
    static F div_unchecked(const F& s, const F& d) {
        if (s.k && d.k) return v_const(fp_mul(s.c, fp_inv(d.c)));
        F r = v_alloc();
        v_mul_equals(r, d, s);
        return r;
    }
};
struct T2 {
    typedef V2 F;
    typedef Fp2 N;
    static F add(const F& a, const F& b) { return v2_add(a, b); }
    static F sub(const F& a, const F& b) { return v2_sub(a, b); }
    static F neg(const F& a) { return v2_neg(a); }
    static F dbl(const F& a) { return v2_dbl(a); }
    static F mul(const F& a, const F& b) { return v2_mul(a, b); }
    static F sqr(const F& a) { return v2_sqr(a); }
    static F mul3b(const F& a) { return v2_mulc(a, K_G2_3B()); }
    static F constant(const N& c) { return v2_const(c); }
    static F zero() { return v2_zero(); }
    static F one() { return v2_one(); }
    static F alloc() { return v2_alloc(); }
    static F input() { return v2_input(); }
    static bool konst(const F& a) { return a.k(); }
    static N value(const F& a) { return {a.c0.c, a.c1.c}; }
    static bool nzero(const N& a) { return fp2_is_zero(a); }
    static N ninv(const N& a) { return fp2_inv(a); }
    static N nmul(const N& a, const N& b) { return fp2_mul(a, b); }
    static B is_eq(const F& a, const F& b) { return v2_is_eq(a, b); }
    static F select(const B& c, const F& t, const F& f) { return v2_select(c, t, f); }
    static void mul_equals(const F& a, const F& b, const F& r) { v2_mul_equals(a, b, r); }
    static F from_bool(const B& b) { return v2_from_bool(b); }
    static F div_unchecked(const F& s, const F& d) { return v2_div_unchecked(s, d); }
};
template <class T>
struct Pt {
    typename T::F x, y, z;
    bool konst() const { return T::konst(x) && T::konst(y) && T::konst(z); }
    bool const_is_zero() const { return T::nzero(T::value(z)); }
};
template <class T>
Pt<T> pt_zero() { return {T::zero(), T::one(), T::zero()}; }
template <class T>
Pt<T> pt_neg(const Pt<T>& p) { return {p.x, T::neg(p.y), p.z}; }
// complete doubling (Renes-Costello-Batina 2015, algorithm 9 shape as ark-r1cs-std writes it, a = 0)
template <class T>
Pt<T> pt_double(const Pt<T>& p) {
    typedef typename T::F F;
    F xx = T::sqr(p.x);
    F yy = T::sqr(p.y);
    F zz = T::sqr(p.z);
    F xy2 = T::dbl(T::mul(p.x, p.y));
    F xz2 = T::dbl(T::mul(p.x, p.z));
    F bzz3 = T::mul3b(zz);
    F yy_m = T::sub(yy, bzz3), yy_p = T::add(yy, bzz3);
    F y_frag = T::mul(yy_p, yy_m);
    F x_frag = T::mul(yy_m, xy2);
    F bxz3 = T::mul3b(xz2);
    F xx3 = T::add(T::dbl(xx), xx);
    F m = T::mul(xx3, bxz3);
    F y = T::add(y_frag, m);
    F yz2 = T::dbl(T::mul(p.y, p.z));
    F t = T::mul(bxz3, yz2);
    F x = T::sub(x_frag, t);
    F z = T::dbl(T::dbl(T::mul(yz2, yy)));
    return {x, y, z};
}
template <class T>
Pt<T> pt_add_mixed(const Pt<T>& p, const typename T::F& x2, const typename T::F& y2) {  // other has z = 1 (a constant point)
    typedef typename T::F F;
    F xx = T::mul(p.x, x2);
    F yy = T::mul(p.y, y2);
    F t0 = T::mul(T::add(p.x, p.y), T::add(x2, y2));
    F xy = T::sub(t0, T::add(xx, yy));
    F t1 = T::mul(x2, p.z);
    F xz = T::add(t1, p.x);
    F t2 = T::mul(y2, p.z);
    F yz = T::add(t2, p.y);
    F bz3 = T::mul3b(p.z);
    F yy_m = T::sub(yy, bz3), yy_p = T::add(yy, bz3);
    F xx3 = T::add(T::dbl(xx), xx);
    F bxz3 = T::mul3b(xz);
    F m0 = T::mul(yy_m, xy);
    F m1 = T::mul(yz, bxz3);
    F m2 = T::mul(yy_p, yy_m);
    F m3 = T::mul(xx3, bxz3);
    F m4 = T::mul(yy_p, yz);
    F m5 = T::mul(xy, xx3);
    return {T::sub(m0, m1), T::add(m2, m3), T::add(m4, m5)};
}
template <class T>
Pt<T> pt_add(const Pt<T>& a_, const Pt<T>& b_) {
    typedef typename T::F F;
    const Pt<T>* self = &a_;
    const Pt<T>* other = &b_;
    if (self->konst()) std::swap(self, other);
    if (other->konst()) {
        if (other->const_is_zero()) return *self;
        typename T::N zi = T::ninv(T::value(other->z));
        return pt_add_mixed<T>(*self, T::constant(T::nmul(T::value(other->x), zi)), T::constant(T::nmul(T::value(other->y), zi)));
    }
    const F &x1 = self->x, &y1 = self->y, &z1 = self->z, &x2 = other->x, &y2 = other->y, &z2 = other->z;
    F xx = T::mul(x1, x2);
    F yy = T::mul(y1, y2);
    F zz = T::mul(z1, z2);
    F t0 = T::mul(T::add(x1, y1), T::add(x2, y2));
    F xy = T::sub(t0, T::add(xx, yy));
    F t1 = T::mul(T::add(x1, z1), T::add(x2, z2));
    F xz = T::sub(t1, T::add(xx, zz));
    F t2 = T::mul(T::add(y1, z1), T::add(y2, z2));
    F yz = T::sub(t2, T::add(yy, zz));
    F bzz3 = T::mul3b(zz);
    F yy_m = T::sub(yy, bzz3), yy_p = T::add(yy, bzz3);
    F xx3 = T::add(T::dbl(xx), xx);
    F bxz3 = T::mul3b(xz);
    F m0 = T::mul(yy_m, xy);
    F m1 = T::mul(yz, bxz3);
    F m2 = T::mul(yy_p, yy_m);
    F m3 = T::mul(xx3, bxz3);
    F m4 = T::mul(yy_p, yz);
    F m5 = T::mul(xy, xx3);
    return {T::sub(m0, m1), T::add(m2, m3), T::add(m4, m5)};
}
template <class T>
Pt<T> pt_select(const B& c, const Pt<T>& t, const Pt<T>& f) {
    auto x = T::select(c, t.x, f.x);
    auto y = T::select(c, t.y, f.y);
    auto z = T::select(c, t.z, f.z);
    return {x, y, z};
}
template <class T>
B pt_is_zero(const Pt<T>& p) { return T::is_eq(p.z, T::zero()); }
template <class T>
struct Aff {
    typename T::F x, y;
    B infinity;
};
template <class T>
Aff<T> pt_to_affine(const Pt<T>& p) {  // variable points only (every call site of these circuits)
    typedef typename T::F F;
    B infinity = pt_is_zero<T>(p);
    F z_inv = T::alloc();
    T::mul_equals(z_inv, p.z, T::from_bool(b_not(infinity)));
    F nzx = T::mul(p.x, z_inv);
    F nzy = T::mul(p.y, z_inv);
    F x = T::select(infinity, T::zero(), nzx);
    F y = T::select(infinity, T::zero(), nzy);
    return {x, y, infinity};
}
template <class T>
B pt_is_eq(const Pt<T>& a, const Pt<T>& b) {
    typedef typename T::F F;
    F l0 = T::mul(a.x, b.z);
    F r0 = T::mul(b.x, a.z);
    B x_eq = T::is_eq(l0, r0);
    F l1 = T::mul(a.y, b.z);
    F r1 = T::mul(b.y, a.z);
    B y_eq = T::is_eq(l1, r1);
    B coords = b_and(x_eq, y_eq);
    B za = pt_is_zero<T>(a);
    B zb = pt_is_zero<T>(b);
    return b_or(b_and(za, zb), coords);
}
// NonZeroAffineVar double / add_unchecked on variable points
template <class T>
void nz_double(typename T::F& x, typename T::F& y) {
    typedef typename T::F F;
    F xs = T::sqr(x);
    F num = T::add(T::dbl(xs), xs);
    F lambda = T::div_unchecked(num, T::dbl(y));
    F l2 = T::sqr(lambda);
    F x3 = T::sub(l2, T::dbl(x));
    F t = T::mul(lambda, T::sub(x, x3));
    y = T::sub(t, y);
    x = x3;
}
template <class T>
void nz_add(typename T::F& px, typename T::F& py, const typename T::F& qx, const typename T::F& qy) {  // p <- p + q
    typedef typename T::F F;
    F lambda = T::div_unchecked(T::sub(qy, py), T::sub(qx, px));
    F l2 = T::sqr(lambda);
    F x3 = T::sub(T::sub(l2, px), qx);
    F t = T::mul(lambda, T::sub(px, x3));
    py = T::sub(t, py);
    px = x3;
}
// ProjectiveVar::scalar_mul_le with CONSTANT bits (little-endian, trailing zeros already stripped)
template <class T>
Pt<T> pt_scalar_mul_le_const(const Pt<T>& self, const std::vector<bool>& bits) {
    typedef typename T::F F;
    Aff<T> aff = pt_to_affine<T>(self);
    F mx = aff.x, my = aff.y;
    Pt<T> mul_result = pt_zero<T>();
    for (size_t off = 0; off < bits.size(); off += 255) {
        const size_t n = std::min((size_t)255, bits.size() - off);
        const size_t split = std::min((size_t)253, n);
        F ax = mx, ay = my;
        Pt<T> initial = {mx, my, T::one()};
        nz_double<T>(mx, my);
        for (size_t i = 1; i < split; i++) {
            if (bits[off + i]) nz_add<T>(ax, ay, mx, my);
            nz_double<T>(mx, my);
        }
        Pt<T> result = {ax, ay, T::one()};
        Pt<T> subtrahend = bits[off] ? pt_zero<T>() : initial;  // select on a constant bit
        mul_result = pt_add<T>(mul_result, pt_add<T>(result, pt_neg<T>(subtrahend)));
        for (size_t i = split; i < n; i++) {
            if (bits[off + i]) mul_result = pt_add<T>(mul_result, Pt<T>{mx, my, T::one()});
            nz_double<T>(mx, my);
        }
    }
    return pt_select<T>(aff.infinity, pt_zero<T>(), mul_result);
}
// result = 0; for bits of k, most significant first: double, add ge on set bits (the prime-order checks of new_variable)
template <class T>
Pt<T> pt_mul_bits_be(const Pt<T>& ge, const uint32_t* words, int nbits) {
    Pt<T> result = pt_zero<T>();
    for (int i = nbits - 1; i >= 0; i--) {
        result = result.konst() ? result : pt_double<T>(result);  // doubling the constant zero stays the constant zero
        if ((words[i >> 5] >> (i & 31)) & 1) result = pt_add<T>(result, ge);
    }
    return result;
}
template <class T>
Pt<T> pt_alloc() {
    auto x = T::alloc();
    auto y = T::alloc();
    auto z = T::alloc();
    return {x, y, z};
}
// ProjectiveVar::new_variable(Input) = new_variable_omit_prime_order_check (ark-r1cs-std 0.4.0): x, y, z as public inputs, no in-circuit check
template <class T>
Pt<T> pt_input() {
    auto x = T::input();
    auto y = T::input();
    auto z = T::input();
    return {x, y, z};
}
static Pt<T1> g1_new_witness() {  // allocate g * (h^-1 mod r), multiply by the cofactor in-circuit
    constexpr uint32_t H1[4] = BLSW_H1_WORDS;
    return pt_mul_bits_be<T1>(pt_alloc<T1>(), H1, BLSW_H1_NBITS);
}
static Pt<T2> g2_new_witness() {  // allocate g, multiply by r - 1 in-circuit, then `ge.enforce_equal(&ge)` (ark-r1cs-std 0.4.0)
    constexpr uint32_t RM1[8] = BLSW_RM1_WORDS;
    Pt<T2> ge = pt_alloc<T2>();
    (void)pt_mul_bits_be<T2>(ge, RM1, BLSW_RM1_NBITS);
    b_enforce_equal_const(pt_is_eq<T2>(ge, ge), true);
    return ge;
}

// ------------------------------------------------------------------------------------------------ UInt8 / UInt32 / SHA-256
struct U8 {
    B b[8];
};
struct U32 {
    B b[32];
    bool konst() const {
        for (int i = 0; i < 32; i++)
            if (!b[i].konst()) return false;
        return true;
    }
    uint32_t cvalue() const {
        uint32_t v = 0;
        for (int i = 0; i < 32; i++) v |= (uint32_t)(b[i].cv ? 1 : 0) << i;
        return v;
    }
};
static U8 u8_const(uint8_t v) {
    U8 r;
    for (int i = 0; i < 8; i++) r.b[i] = b_const((v >> i) & 1);
    return r;
}
static U8 u8_alloc() {
    U8 r;
    for (int i = 0; i < 8; i++) r.b[i] = b_alloc();
    return r;
}
static U32 u32_const(uint32_t v) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = b_const((v >> i) & 1);
    return r;
}
static U32 u32_rotr(const U32& a, int by) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = a.b[(i + by) % 32];
    return r;
}
static U32 u32_shr(const U32& a, int by) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = (i + by < 32) ? a.b[i + by] : b_const(false);
    return r;
}
static U32 u32_xor(const U32& a, const U32& b) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = b_xor(a.b[i], b.b[i]);
    return r;
}
static U32 u32_and(const U32& a, const U32& b) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = b_and(a.b[i], b.b[i]);
    return r;
}
static U32 u32_not(const U32& a) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = b_not(a.b[i]);
    return r;
}
static U32 u32_from_bytes_be(const U8* bytes) {
    U32 r;
    for (int k = 0; k < 4; k++)
        for (int j = 0; j < 8; j++) r.b[k * 8 + j] = bytes[3 - k].b[j];
    return r;
}
static U32 u32_addmany(const U32* ops, int n) {
    bool all_const = true;
    uint64_t csum = 0;
    for (int i = 0; i < n; i++) {
        all_const = all_const && ops[i].konst();
        if (ops[i].konst()) csum += ops[i].cvalue();
    }
    if (all_const) return u32_const((uint32_t)csum);
    const int nbits = n == 2 ? 33 : (n <= 4 ? 34 : 35);
    Lc lc;
    for (int k = 0; k < n; k++) {
        Fp coeff = fp_one();
        for (int i = 0; i < 32; i++) {
            const B& bit = ops[k].b[i];
            if (!(bit.kind == 0 && !bit.cv)) lc = lc_axpy(lc, b_lc(bit), coeff);
            coeff = fp_dbl(coeff);
        }
    }
    U32 r;
    Fp coeff = fp_one();
    for (int i = 0; i < nbits; i++) {
        B bit = b_alloc();
        lc.push_back({bit.var, fp_neg(coeff)});  // fresh variable: larger than every index in lc
        if (i < 32) r.b[i] = bit;
        coeff = fp_dbl(coeff);
    }
    S->enforce(Lc(), Lc(), lc);
    return r;
}
static void sha_update_state(U32 st[8], const U8* data) {
    constexpr uint32_t K[64] = BLSW_SHA_K;
    std::vector<U32> w(64);
    for (int i = 0; i < 16; i++) w[i] = u32_from_bytes_be(data + 4 * i);
    for (int i = 16; i < 64; i++) {
        U32 a1 = u32_xor(u32_rotr(w[i - 15], 7), u32_rotr(w[i - 15], 18));
        U32 s0 = u32_xor(a1, u32_shr(w[i - 15], 3));
        U32 b1 = u32_xor(u32_rotr(w[i - 2], 17), u32_rotr(w[i - 2], 19));
        U32 s1 = u32_xor(b1, u32_shr(w[i - 2], 10));
        U32 ops[4] = {w[i - 16], s0, w[i - 7], s1};
        w[i] = u32_addmany(ops, 4);
    }
    U32 h[8];
    for (int i = 0; i < 8; i++) h[i] = st[i];
    for (int i = 0; i < 64; i++) {
        U32 c1 = u32_and(h[4], h[5]);
        U32 c2 = u32_and(u32_not(h[4]), h[6]);
        U32 ch = u32_xor(c1, c2);
        U32 m1 = u32_and(h[0], h[1]);
        U32 m2 = u32_and(h[0], h[2]);
        U32 m3 = u32_and(h[1], h[2]);
        U32 m12 = u32_xor(m1, m2);
        U32 ma = u32_xor(m12, m3);
        U32 p1 = u32_xor(u32_rotr(h[0], 2), u32_rotr(h[0], 13));
        U32 s0 = u32_xor(p1, u32_rotr(h[0], 22));
        U32 q1 = u32_xor(u32_rotr(h[4], 6), u32_rotr(h[4], 11));
        U32 s1 = u32_xor(q1, u32_rotr(h[4], 25));
        U32 o5[5] = {h[7], s1, ch, u32_const(K[i]), w[i]};
        U32 t0 = u32_addmany(o5, 5);
        U32 o2[2] = {s0, ma};
        U32 t1 = u32_addmany(o2, 2);
        h[7] = h[6];
        h[6] = h[5];
        h[5] = h[4];
        U32 o3[2] = {h[3], t0};
        h[4] = u32_addmany(o3, 2);
        h[3] = h[2];
        h[2] = h[1];
        h[1] = h[0];
        U32 o4[2] = {t0, t1};
        h[0] = u32_addmany(o4, 2);
    }
    for (int i = 0; i < 8; i++) {
        U32 o[2] = {st[i], h[i]};
        st[i] = u32_addmany(o, 2);
    }
}
static std::vector<U8> sha256_digest(const std::vector<U8>& data) {
    constexpr uint32_t H0[8] = BLSW_SHA_H0;
    U32 st[8];
    for (int i = 0; i < 8; i++) st[i] = u32_const(H0[i]);
    std::vector<U8> buf = data;
    const uint64_t bitlen = (uint64_t)data.size() * 8;
    buf.push_back(u8_const(0x80));
    while (buf.size() % 64 != 56) buf.push_back(u8_const(0));
    for (int i = 7; i >= 0; i--) buf.push_back(u8_const((uint8_t)(bitlen >> (8 * i))));
    for (size_t o = 0; o < buf.size(); o += 64) sha_update_state(st, buf.data() + o);
    std::vector<U8> out(32);
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 4; k++)
            for (int j = 0; j < 8; j++) out[4 * i + 3 - k].b[j] = st[i].b[k * 8 + j];
    return out;
}
// expand_message_xmd, 256 bytes (hasher.rs:110-173): lib_str is two WITNESS bytes
static std::vector<U8> expand_message(const std::vector<U8>& msg) {
    const char dst[] = BLSW_DST;
    std::vector<U8> dst_prime;
    for (int i = 0; i < BLSW_DST_LEN; i++) dst_prime.push_back(u8_const((uint8_t)dst[i]));
    dst_prime.push_back(u8_const(BLSW_DST_LEN));
    std::vector<U8> lib = {u8_alloc(), u8_alloc()};
    std::vector<U8> mp(64, u8_const(0));
    mp.insert(mp.end(), msg.begin(), msg.end());
    mp.insert(mp.end(), lib.begin(), lib.end());
    mp.push_back(u8_const(0));
    mp.insert(mp.end(), dst_prime.begin(), dst_prime.end());
    std::vector<U8> b0 = sha256_digest(mp);
    std::vector<U8> d = b0;
    d.push_back(u8_const(1));
    d.insert(d.end(), dst_prime.begin(), dst_prime.end());
    std::vector<U8> last = sha256_digest(d), ret = last;
    for (int i = 2; i <= 8; i++) {
        std::vector<U8> bx(32);
        for (int k = 0; k < 32; k++)
            for (int j = 0; j < 8; j++) bx[k].b[j] = b_xor(b0[k].b[j], last[k].b[j]);
        bx.push_back(u8_const((uint8_t)i));
        bx.insert(bx.end(), dst_prime.begin(), dst_prime.end());
        last = sha256_digest(bx);
        ret.insert(ret.end(), last.begin(), last.end());
    }
    return ret;
}
// [UInt8]::to_constraint_field on <= 47 little-endian bytes: a linear combination
static V le_bytes_to_fp(const U8* bytes, size_t n) {
    Lc acc;
    Fp coeff = fp_one();
    bool all_const = true;
    for (size_t i = 0; i < n; i++)
        for (int j = 0; j < 8; j++) {
            all_const = all_const && bytes[i].b[j].konst();
            acc = lc_axpy(acc, b_lc(bytes[i].b[j]), coeff);
            coeff = fp_dbl(coeff);
        }
    if (all_const) return v_const(acc.empty() ? fp_zero() : acc[0].c);
    return {false, fp_zero(), acc};
}
static void hash_to_field(const std::vector<U8>& msg, V2 u[2]) {  // hasher.rs:58-107
    std::vector<U8> uniform = expand_message(msg);
    const Fp c256 = fp_from_u32(256);
    for (int i = 0; i < 2; i++) {
        V e[2];
        for (int j = 0; j < 2; j++) {
            std::vector<U8> le(uniform.begin() + 64 * (j + 2 * i), uniform.begin() + 64 * (j + 2 * i) + 64);
            std::reverse(le.begin(), le.end());
            V f = le_bytes_to_fp(le.data() + 17, 47);
            V tail = le_bytes_to_fp(le.data(), 17);
            for (int l = 0; l < 17; l++) f = v_scale(f, c256);
            e[j] = v_add(f, tail);
        }
        u[i] = {e[0], e[1]};
    }
}

// ------------------------------------------------------------------------------------------------ hasher.rs: SSWU, isogeny, cofactor
static B sgn0(const V2& v) {  // hasher.rs:520-530
    std::vector<B> b0 = v_to_bits_le(v.c0);
    std::vector<B> b1 = v_to_bits_le(v.c1);
    B zero_0 = v_is_eq(v.c0, v_zero());
    return b_or(b0[0], b_and(zero_0, b1[0]));
}
static V2 pow_c1(const V2& v) {  // hasher.rs:532-548: square, then multiply by select(bit, v, 1) on constant bits
    constexpr uint32_t C1[24] = BLSW_SSWU_C1_WORDS;
    V2 r = v2_one();
    for (int i = BLSW_SSWU_C1_NBITS - 1; i >= 0; i--) {
        r = v2_sqr(r);
        if ((C1[i >> 5] >> (i & 31)) & 1) r = v2_mul(r, v);
    }
    return r;
}
static V2 poly_eval(const Fp2* k, int n, const V2& x) {  // hasher.rs:195-206 (one unused power per polynomial)
    V2 result = v2_zero(), cp = v2_one();
    for (int i = 0; i < n; i++) {
        result = v2_add(result, v2_mul(cp, v2_const(k[i])));
        cp = v2_mul(cp, x);
    }
    return result;
}
static Pt<T2> map_to_curve(const V2& u) {
    const Fp2 Z = K_SSWU_Z(), A = K_SSWU_A(), Bc = K_SSWU_B(), C2 = K_SSWU_C2(), C3 = K_SSWU_C3(), C4 = K_SSWU_C4(), C5 = K_SSWU_C5();
    V2 tv1 = v2_sqr(u);
    V2 tv3 = v2_mulc(tv1, Z);
    V2 tv5 = v2_sqr(tv3);
    V2 xd = v2_add(tv5, tv3);
    V2 x1n = v2_mulc(v2_add(xd, v2_one()), Bc);
    xd = v2_mulc(xd, K_SSWU_NEG_A());
    B e1 = v2_is_zero(xd);
    xd = v2_select(e1, v2_const(K_SSWU_ZA()), xd);
    V2 tv2 = v2_sqr(xd);
    V2 gxd = v2_mul(tv2, xd);
    tv2 = v2_mulc(tv2, A);
    V2 gx1 = v2_add(v2_sqr(x1n), tv2);
    gx1 = v2_mul(gx1, x1n);
    tv2 = v2_mulc(gxd, Bc);
    gx1 = v2_add(gx1, tv2);
    V2 tv4 = v2_sqr(gxd);
    tv2 = v2_mul(tv4, gxd);
    tv4 = v2_sqr(tv4);
    tv2 = v2_mul(tv2, tv4);
    tv2 = v2_mul(tv2, gx1);
    tv4 = v2_sqr(tv4);
    tv4 = v2_mul(tv2, tv4);
    V2 y = pow_c1(tv4);
    y = v2_mul(y, tv2);
    tv4 = v2_mulc(y, C2);
    tv2 = v2_mul(v2_sqr(tv4), gxd);
    B e2 = v2_is_eq(tv2, gx1);
    y = v2_select(e2, tv4, y);
    tv4 = v2_mulc(y, C3);
    tv2 = v2_mul(v2_sqr(tv4), gxd);
    B e3 = v2_is_eq(tv2, gx1);
    y = v2_select(e3, tv4, y);
    tv4 = v2_mulc(tv4, C2);
    tv2 = v2_mul(v2_sqr(tv4), gxd);
    B e4 = v2_is_eq(tv2, gx1);
    y = v2_select(e4, tv4, y);
    V2 gx2 = v2_mul(gx1, tv5);
    gx2 = v2_mul(gx2, tv3);
    tv5 = v2_mul(y, tv1);
    tv5 = v2_mul(tv5, u);
    tv1 = v2_mulc(tv5, C4);
    tv4 = v2_mulc(tv1, C2);
    tv2 = v2_mul(v2_sqr(tv4), gxd);
    B e5 = v2_is_eq(tv2, gx2);
    tv1 = v2_select(e5, tv4, tv1);
    tv4 = v2_mulc(tv5, C5);
    tv2 = v2_mul(v2_sqr(tv4), gxd);
    B e6 = v2_is_eq(tv2, gx2);
    tv1 = v2_select(e6, tv4, tv1);
    tv4 = v2_mulc(tv4, C2);
    tv2 = v2_mul(v2_sqr(tv4), gxd);
    B e7 = v2_is_eq(tv2, gx2);
    tv1 = v2_select(e7, tv4, tv1);
    tv2 = v2_mul(v2_sqr(y), gxd);
    B e8 = v2_is_eq(tv2, gx1);
    y = v2_select(e8, y, tv1);
    tv2 = v2_mul(tv3, x1n);
    V2 xn = v2_select(e8, x1n, tv2);
    B su = sgn0(u);
    B sy = sgn0(y);
    B e9 = b_not(b_xor(su, sy));
    y = v2_select(e9, y, v2_neg(y));
    // to_projective_short (hasher.rs:551-559)
    V2 xd2 = v2_sqr(xd);
    V2 xd3 = v2_mul(xd2, xd);
    V2 jx = v2_mul(xn, xd);
    V2 jy = v2_mul(y, xd3);
    // isogeny_map (hasher.rs:294-348); to_affine_unchecked (:569-583)
    B is_inf = v2_is_zero(xd);
    V2 z_inv = v2_inv(xd);
    V2 zi2 = v2_sqr(z_inv);
    V2 zi3 = v2_mul(zi2, z_inv);
    V2 ax = v2_mul(jx, zi2);
    V2 ay = v2_mul(jy, zi3);
    const Fp2 xden[3] = {K_ISO_XDEN0(), K_ISO_XDEN1(), K_ISO_XDEN2()}, yden[4] = {K_ISO_YDEN0(), K_ISO_YDEN1(), K_ISO_YDEN2(), K_ISO_YDEN3()};
    const Fp2 xnum[4] = {K_ISO_XNUM0(), K_ISO_XNUM1(), K_ISO_XNUM2(), K_ISO_XNUM3()}, ynum[4] = {K_ISO_YNUM0(), K_ISO_YNUM1(), K_ISO_YNUM2(), K_ISO_YNUM3()};
    V2 x_den_inv = v2_inv(poly_eval(xden, 3, ax));
    V2 y_den_inv = v2_inv(poly_eval(yden, 4, ax));
    V2 x_num = poly_eval(xnum, 4, ax);
    V2 y_num = poly_eval(ynum, 4, ax);
    V2 img_x = v2_mul(x_num, x_den_inv);
    V2 t = v2_mul(y_num, ay);
    V2 img_y = v2_mul(t, y_den_inv);
    Pt<T2> proj = {img_x, img_y, v2_one()}, zero = {v2_zero(), v2_zero(), v2_zero()};
    return pt_select<T2>(is_inf, zero, proj);
}
static Pt<T2> hash_to_g2(const std::vector<U8>& msg) {  // hasher.rs:641-673, 727-740
    constexpr uint32_t HE[20] = BLSW_H_EFF_WORDS;
    V2 u[2];
    hash_to_field(msg, u);
    Pt<T2> q0 = map_to_curve(u[0]);
    Pt<T2> q1 = map_to_curve(u[1]);
    Pt<T2> r = pt_add<T2>(q0, q1);
    std::vector<bool> bits(BLSW_H_EFF_NBITS);
    for (int i = 0; i < BLSW_H_EFF_NBITS; i++) bits[i] = (HE[i >> 5] >> (i & 31)) & 1;
    return pt_scalar_mul_le_const<T2>(r, bits);
}

// ------------------------------------------------------------------------------------------------ pairing
struct Coeffs {
    std::vector<std::pair<V2, V2>> ell;
};
static Coeffs g2_prepare(const Pt<T2>& q_) {  // G2PreparedVar::from_group_var
    Aff<T2> q = pt_to_affine<T2>(q_);
    b_enforce_not_true(q.infinity);
    const Fp two_inv = K_TWO_INV();
    Coeffs out;
    V2 rx = q.x, ry = q.y;
    for (int i = 62; i >= 0; i--) {
        {
            V2 a = v2_inv(ry);
            V2 b = v2_sqr(rx);
            b = v2_add(v2_scale_fp(b, two_inv), b);
            V2 c = v2_mul(a, b);
            V2 x3 = v2_sub(v2_sqr(c), v2_dbl(rx));
            V2 cx = v2_mul(c, rx);
            V2 e = v2_sub(cx, ry);
            V2 c_x3 = v2_mul(c, x3);
            ry = v2_sub(e, c_x3);
            rx = x3;
            out.ell.push_back({e, v2_neg(c)});
        }
        if ((BLSW_X_ABS >> i) & 1) {
            V2 a = v2_inv(v2_sub(q.x, rx));
            V2 b = v2_sub(q.y, ry);
            V2 c = v2_mul(a, b);
            V2 x3 = v2_sub(v2_sqr(c), v2_add(rx, q.x));
            V2 e = v2_mul(v2_sub(rx, x3), c);
            V2 y3 = v2_sub(e, ry);
            V2 cr = v2_mul(c, rx);
            V2 g = v2_sub(cr, ry);
            rx = x3;
            ry = y3;
            out.ell.push_back({g, v2_neg(c)});
        }
    }
    return out;
}
struct G1Prep {
    V x, y;
};
static G1Prep g1_prepare(const Pt<T1>& p) {
    Aff<T1> a = pt_to_affine<T1>(p);
    return {a.x, a.y};
}
static V12 ell(const V12& f, const std::pair<V2, V2>& co, const G1Prep& p) {
    V k0 = v_mul(co.second.c0, p.x);
    V k1 = v_mul(co.second.c1, p.x);
    return v12_mul_by_014(f, co.first, {k0, k1}, {p.y, v_zero()});
}
static V12 miller_loop(const std::vector<G1Prep>& ps, const std::vector<Coeffs>& qs) {
    V12 f = v12_one();
    size_t idx = 0;
    for (int i = 62; i >= 0; i--) {
        f = v12_sqr(f);
        for (size_t k = 0; k < ps.size(); k++) f = ell(f, qs[k].ell[idx], ps[k]);
        idx++;
        if ((BLSW_X_ABS >> i) & 1) {
            for (size_t k = 0; k < ps.size(); k++) f = ell(f, qs[k].ell[idx], ps[k]);
            idx++;
        }
    }
    return v12_conj(f);
}
static V12 final_exponentiation(const V12& f) {
    V12 f1 = v12_conj(f);
    V12 f2 = v12_inv(f);
    V12 r = v12_mul(f1, f2);
    f2 = r;
    r = v12_mul(v12_frob(r, 2), f2);
    V12 y0 = v12_conj(v12_cyclotomic_square(r));
    V12 y5 = v12_exp_by_x(r);
    V12 y1 = v12_cyclotomic_square(y5);
    V12 y3 = v12_mul(y0, y5);
    y0 = v12_exp_by_x(y3);
    V12 y2 = v12_exp_by_x(y0);
    V12 y4 = v12_exp_by_x(y2);
    y4 = v12_mul(y4, y1);
    y1 = v12_exp_by_x(y4);
    y3 = v12_conj(y3);
    y1 = v12_mul(y1, y3);
    y1 = v12_mul(y1, r);
    y3 = v12_conj(r);
    y0 = v12_frob(v12_mul(y0, r), 3);
    y4 = v12_frob(v12_mul(y4, y3), 1);
    y5 = v12_frob(v12_mul(y5, y2), 2);
    y5 = v12_mul(y5, y0);
    y5 = v12_mul(y5, y4);
    return v12_mul(y5, y1);
}

// ------------------------------------------------------------------------------------------------ constraints.rs
static std::vector<U8> msg_alloc(uint32_t msg_len) {
    std::vector<U8> m(msg_len);
    for (uint32_t i = 0; i < msg_len; i++) m[i] = u8_alloc();
    return m;
}
// verify over K (pk, msg) pairs and one signature (K = 1: constraints.rs:90-128 statement by statement)
// g1: the generator variable of a ParametersVar allocated as witnesses (constraints.rs:198-211), or nullptr for Constant parameters
static B verify_gadget(const std::vector<Pt<T1>>& pks, const std::vector<std::vector<U8>>& msgs, const Pt<T2>& sig, const Pt<T1>* g1 = nullptr) {
    for (auto& pk : pks) b_enforce_equal_const(pt_is_eq<T1>(pk, pt_zero<T1>()), false);  // pk.enforce_not_equal(zero)
    G1Prep g1n = {v_const(K_G1_GEN_X()), v_const(K_G1_GEN_NEG_Y())};  // prepare_g1(-g1): a constant
    std::vector<Pt<T2>> hs;
    for (auto& m : msgs) hs.push_back(hash_to_g2(m));
    if (g1) g1n = g1_prepare(pt_neg<T1>(*g1));  // g1.negate() is linear, prepare_g1 = to_affine (constraints.rs:107-108, 117)
    std::vector<G1Prep> ps = {g1n};
    std::vector<Coeffs> qs(1);
    for (auto& h : hs) qs.push_back(g2_prepare(h));
    for (auto& pk : pks) ps.push_back(g1_prepare(pk));
    qs[0] = g2_prepare(sig);
    V12 fe = final_exponentiation(miller_loop(ps, qs));
    return v12_is_eq(fe, v12_one());
}
static void circuit(uint32_t msg_len, uint32_t n_keys, uint32_t n_pairs, bool params_witness, bool pk_input = false, bool sig_input = false) {
    if (n_keys) {  // constraints.rs:378-441: keys, bitmap booleans, msg, params, sig, aggregate_verify
        std::vector<Pt<T1>> keys;
        for (uint32_t k = 0; k < n_keys; k++) keys.push_back(g1_new_witness());
        std::vector<B> bitmap;
        for (uint32_t k = 0; k < n_keys; k++) bitmap.push_back(b_alloc());
        std::vector<U8> msg = msg_alloc(msg_len);
        Pt<T2> sig = g2_new_witness();
        // mapped_aggregate (constraints.rs:169-191)
        U32 count;
        for (int i = 0; i < 32; i++) count.b[i] = b_alloc();
        Pt<T1> ret = pt_zero<T1>();
        for (uint32_t k = 0; k < n_keys; k++) {
            ret = pt_add<T1>(ret, pt_select<T1>(bitmap[k], keys[k], pt_zero<T1>()));
            U32 inc;  // bit.select(&count_one, &count_zero): (x, FALSE) arms = cond AND x
            for (int b = 0; b < 32; b++) inc.b[b] = b == 0 ? b_and(bitmap[k], b_const(true)) : b_and(bitmap[k], b_const(false));
            U32 ops[2] = {count, inc};
            count = u32_addmany(ops, 2);
        }
        (void)verify_gadget({ret}, {msg}, sig);
        return;
    }
    std::vector<std::vector<U8>> msgs;
    for (uint32_t j = 0; j < n_pairs; j++) msgs.push_back(msg_alloc(msg_len));
    Pt<T1> g1 = pt_zero<T1>();
    if (params_witness) g1 = g1_new_witness();  // ParametersVar::new_variable(Witness): argument order of constraints.rs:346-364
    std::vector<Pt<T1>> pks;
    for (uint32_t j = 0; j < n_pairs; j++) pks.push_back(pk_input ? pt_input<T1>() : g1_new_witness());  // constraints.rs:214-232 with AllocationMode::Input / Witness
    Pt<T2> sig = sig_input ? pt_input<T2>() : g2_new_witness();                                             // constraints.rs:234-249
    (void)verify_gadget(pks, msgs, sig, params_witness ? &g1 : nullptr);
}

// io_modes: bit 0 = pk Input, bit 1 = sig Input (single-key circuit with Constant parameters)
static int run(uint32_t msg_len, uint32_t n_keys, uint32_t n_pairs, Sys& sys, uint32_t params_mode = 0, uint32_t io_modes = 0) {
    if (msg_len > 65535 || n_keys > 65535 || (n_keys && n_pairs > 1) || n_pairs == 0 || n_pairs > 4096) return BLSW_ERR_ARG;
    if (params_mode > 1 || (params_mode && (n_keys || n_pairs != 1))) return BLSW_ERR_ARG;
    if (io_modes > 3 || (io_modes && (n_keys || n_pairs != 1 || params_mode))) return BLSW_ERR_ARG;
    S = &sys;
    sys.n_inst = 1 + ((io_modes & 1) ? 3 : 0) + ((io_modes & 2) ? 6 : 0);
    circuit(msg_len, n_keys, n_pairs, params_mode == 1, (io_modes & 1) != 0, (io_modes & 2) != 0);
    sys.finish();
    S = nullptr;
    return BLSW_OK;
}
// blsw_matrices_info synthesises the system to count it; the result is kept (one shape, ~50 bytes per non-zero) so that the
// blsw_matrices_fill that follows copies instead of synthesising a second time, and released by that fill
struct Cache {
    std::mutex mu;
    bool valid = false;
    uint32_t msg_len = 0, n_keys = 0, n_pairs = 0, params_mode = 0, io_modes = 0;
    Sys sys;
};
static Cache& cache() {
    static Cache c;
    return c;
}

}  // namespace r1cs
}  // namespace blsw

static int matrices_info(uint32_t msg_len, uint32_t n_keys, uint32_t n_pairs, uint32_t params_mode, blsw_matrices_info_t* out, uint32_t io_modes = 0) {
    if (!out) return BLSW_ERR_ARG;
    blsw::r1cs::Cache& c = blsw::r1cs::cache();
    std::lock_guard<std::mutex> lock(c.mu);
    c.valid = false;
    int rc;
    try {  // no exception crosses the ABI: a system that does not fit in memory is BLSW_ERR_WORKSPACE
        c.sys = blsw::r1cs::Sys();
        rc = blsw::r1cs::run(msg_len, n_keys, n_pairs, c.sys, params_mode, io_modes);
    } catch (...) {
        c.sys = blsw::r1cs::Sys();
        return BLSW_ERR_WORKSPACE;
    }
    if (rc) return rc;
    c.valid = true;
    c.io_modes = io_modes;
    c.msg_len = msg_len;
    c.n_keys = n_keys;
    c.n_pairs = n_pairs;
    c.params_mode = params_mode;
    out->n_constraints = c.sys.n_cons;
    out->n_instance_vars = c.sys.n_inst;
    out->n_witness = c.sys.n_wit;
    for (int m = 0; m < 3; m++) out->nnz[m] = c.sys.col[m].size();
    return BLSW_OK;
}

static int matrices_fill(uint32_t msg_len, uint32_t n_keys, uint32_t n_pairs, uint32_t params_mode, const blsw_matrices_info_t* info, blsw_matrices_t* out,
                         uint32_t io_modes = 0) {
    if (!info || !out) return BLSW_ERR_ARG;
    for (int m = 0; m < 3; m++)
        if (!out->row_ptr[m] || (info->nnz[m] && (!out->col[m] || !out->val[m]))) return BLSW_ERR_ARG;
    blsw::r1cs::Cache& c = blsw::r1cs::cache();
    std::lock_guard<std::mutex> lock(c.mu);
    if (!(c.valid && c.msg_len == msg_len && c.n_keys == n_keys && c.n_pairs == n_pairs && c.params_mode == params_mode && c.io_modes == io_modes)) {
        c.valid = false;
        int rc;
        try {
            c.sys = blsw::r1cs::Sys();
            rc = blsw::r1cs::run(msg_len, n_keys, n_pairs, c.sys, params_mode, io_modes);
        } catch (...) {
            c.sys = blsw::r1cs::Sys();
            return BLSW_ERR_WORKSPACE;
        }
        if (rc) return rc;
    }
    const blsw::r1cs::Sys& s = c.sys;
    bool ok = s.n_cons == info->n_constraints && s.n_wit == info->n_witness && s.n_inst == info->n_instance_vars;
    for (int m = 0; m < 3; m++) ok = ok && s.col[m].size() == info->nnz[m];
    if (ok)
        for (int m = 0; m < 3; m++) {
            memcpy(out->row_ptr[m], s.row_ptr[m].data(), (s.n_cons + 1) * sizeof(uint64_t));
            if (!s.col[m].empty()) {
                memcpy(out->col[m], s.col[m].data(), s.col[m].size() * sizeof(uint32_t));
                memcpy(out->val[m], s.val[m].data(), s.val[m].size() * 48);
            }
        }
    c.valid = false;
    c.sys = blsw::r1cs::Sys();  // release
    return ok ? BLSW_OK : BLSW_ERR_ARG;  // BLSW_ERR_ARG: `info` belongs to another circuit shape
}

extern "C" {
int blsw_matrices_info(uint32_t msg_len, uint32_t n_keys, uint32_t n_pairs, blsw_matrices_info_t* out) { return matrices_info(msg_len, n_keys, n_pairs, 0, out); }
int blsw_matrices_fill(uint32_t msg_len, uint32_t n_keys, uint32_t n_pairs, const blsw_matrices_info_t* info, blsw_matrices_t* out) {
    return matrices_fill(msg_len, n_keys, n_pairs, 0, info, out);
}
int blsw_matrices_info_params(uint32_t msg_len, uint32_t params_mode, blsw_matrices_info_t* out) { return matrices_info(msg_len, 0, 1, params_mode, out); }
int blsw_matrices_fill_params(uint32_t msg_len, uint32_t params_mode, const blsw_matrices_info_t* info, blsw_matrices_t* out) {
    return matrices_fill(msg_len, 0, 1, params_mode, info, out);
}
int blsw_matrices_info_io(uint32_t msg_len, uint32_t pk_mode, uint32_t sig_mode, blsw_matrices_info_t* out) {
    if (pk_mode > 1 || sig_mode > 1) return BLSW_ERR_ARG;
    return matrices_info(msg_len, 0, 1, 0, out, pk_mode | sig_mode << 1);
}
int blsw_matrices_fill_io(uint32_t msg_len, uint32_t pk_mode, uint32_t sig_mode, const blsw_matrices_info_t* info, blsw_matrices_t* out) {
    if (pk_mode > 1 || sig_mode > 1) return BLSW_ERR_ARG;
    return matrices_fill(msg_len, 0, 1, 0, info, out, pk_mode | sig_mode << 1);
}
}
