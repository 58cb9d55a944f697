// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Restatement of ark-r1cs-std ^0.4.0 extension-field gadgets (third-party, not vendored):
//   fields/quadratic_extension.rs (Fp2Var, Fp12Var), fields/cubic_extension.rs (Fp6Var),
//   fields/fp6_3over2.rs (mul_by_0_c1_0, mul_by_c0_c1_0), fields/fp12.rs (mul_by_014,
//   cyclotomic_square, optimized_cyclotomic_exp).  Rules: SURVEY.md App. A.2.
// Every Var x Var base-field product allocates ONE witness, in the order written here.
// C++ argument evaluation order is unspecified, so every product is sequenced in a named temporary.
#pragma once
#include "cs.h"

namespace orc {

// ------------------------------------------------------------------ Fp2Var = QuadExtVar<FpVar>
struct Fp2Var {
    FpVar c0, c1;
    bool is_const() const { return c0.konst && c1.konst; }
    Fp2 val() const { return {c0.v, c1.v}; }
};
inline Fp2Var f2const(const Fp2& v) { return {fconst(v.c0), fconst(v.c1)}; }
inline Fp2Var f2witness(const Fp2& v) {
    FpVar a = fwitness(v.c0);
    FpVar b = fwitness(v.c1);
    return {a, b};
}
inline Fp2Var f2input(const Fp2& v) {
    FpVar a = finput(v.c0);
    FpVar b = finput(v.c1);
    return {a, b};
}
inline Fp2Var f2zero() { return f2const(fp2_zero()); }
inline Fp2Var f2one() { return f2const(fp2_one()); }
inline Fp2Var f2add(const Fp2Var& a, const Fp2Var& b) { return {fadd(a.c0, b.c0), fadd(a.c1, b.c1)}; }
inline Fp2Var f2sub(const Fp2Var& a, const Fp2Var& b) { return {fsub(a.c0, b.c0), fsub(a.c1, b.c1)}; }
inline Fp2Var f2neg(const Fp2Var& a) { return {fneg(a.c0), fneg(a.c1)}; }
inline Fp2Var f2dbl(const Fp2Var& a) { return {fdbl(a.c0), fdbl(a.c1)}; }
inline Fp2Var f2conj(const Fp2Var& a) { return {a.c0, fneg(a.c1)}; }
// Karatsuba: v0 = a0*b0, v1 = a1*b1, (a0+a1)*(b0+b1)
inline Fp2Var f2mul(const Fp2Var& a, const Fp2Var& b) {
    FpVar v0 = fmul(a.c0, b.c0);
    FpVar v1 = fmul(a.c1, b.c1);
    FpVar s = fmul(fadd(a.c1, a.c0), fadd(b.c0, b.c1));
    FpVar c1 = fsub(fsub(s, v0), v1);
    FpVar c0 = fsub(v0, v1);  // v0 + beta*v1, beta = -1
    return {c0, c1};
}
// complex squaring: v2 = c0*c1, then (c0-c1)*(c0-beta*c1)
inline Fp2Var f2sqr(const Fp2Var& a) {
    FpVar v0 = fsub(a.c0, a.c1);
    FpVar v3 = fadd(a.c0, a.c1);  // c0 - beta*c1
    FpVar v2 = fmul(a.c0, a.c1);
    FpVar t = fmul(v0, v3);
    t = fadd(t, v2);
    FpVar c0 = fsub(t, v2);  // t + beta*v2
    FpVar c1 = fdbl(v2);
    return {c0, c1};
}
inline Fp2Var f2mulc(const Fp2Var& a, const Fp2& c) { return f2mul(a, f2const(c)); }
inline Fp2Var f2mul_fp_const(const Fp2Var& a, const Fp& c) { return {fmulc(a.c0, c), fmulc(a.c1, c)}; }
inline Fp2Var f2mul_xi(const Fp2Var& a) { return {fsub(a.c0, a.c1), fadd(a.c0, a.c1)}; }
// QuadExtVar::mul_equals: 1 witness (v1) + 3 constraints
inline void f2mul_equals(const Fp2Var& a, const Fp2Var& b, const Fp2Var& r) {
    FpVar v1 = fmul(a.c1, b.c1);
    FpVar nr_v1 = fneg(v1);
    FpVar rhs = fsub(r.c0, nr_v1);
    fmul_equals(a.c0, b.c0, rhs);
    FpVar a01 = fadd(a.c0, a.c1);
    FpVar b01 = fadd(b.c0, b.c1);
    FpVar one_minus_nr_v1 = fsub(v1, nr_v1);
    FpVar tmp = fadd(fadd(one_minus_nr_v1, r.c1), r.c0);
    fmul_equals(a01, b01, tmp);
}
inline Fp2Var f2inv(const Fp2Var& a) {
    Fp2 iv = fp2_is_zero(a.val()) ? fp2_zero() : fp2_inv(a.val());
    if (a.is_const()) return f2const(iv);
    Fp2Var inv = f2witness(iv);
    f2mul_equals(a, inv, f2one());
    return inv;
}
inline Bool f2is_eq(const Fp2Var& a, const Fp2Var& b) {
    Bool b0 = fis_eq(a.c0, b.c0);
    Bool b1 = fis_eq(a.c1, b.c1);
    return band(b0, b1);
}
inline Bool f2is_zero(const Fp2Var& a) { return f2is_eq(a, f2zero()); }
inline Fp2Var f2select(const Bool& c, const Fp2Var& t, const Fp2Var& f) {
    FpVar c0 = fselect(c, t.c0, f.c0);
    FpVar c1 = fselect(c, t.c1, f.c1);
    return {c0, c1};
}
inline Fp2Var f2from_bool(const Bool& b) { return {ffrom_bool(b), fconst(fp_zero())}; }
// FieldVar::mul_by_inverse_unchecked: witness result = self/d, then result*d == self
inline Fp2Var f2mul_by_inverse_unchecked(const Fp2Var& self, const Fp2Var& d) {
    Fp2 dv = fp2_is_zero(d.val()) ? fp2_zero() : fp2_inv(d.val());
    Fp2 rv = fp2_mul(self.val(), dv);
    if (self.is_const() && d.is_const()) return f2const(rv);
    Fp2Var r = f2witness(rv);
    f2mul_equals(r, d, self);
    return r;
}
inline Fp2Var f2frobenius(const Fp2Var& a, int power) { return (power & 1) ? f2conj(a) : a; }

// ------------------------------------------------------------------ Fp6Var = CubicExtVar<Fp2Var>
struct Fp6Var {
    Fp2Var c0, c1, c2;
    Fp6 val() const { return {c0.val(), c1.val(), c2.val()}; }
};
inline Fp6Var f6const(const Fp6& v) { return {f2const(v.c0), f2const(v.c1), f2const(v.c2)}; }
inline Fp6Var f6witness(const Fp6& v) {
    Fp2Var a = f2witness(v.c0);
    Fp2Var b = f2witness(v.c1);
    Fp2Var c = f2witness(v.c2);
    return {a, b, c};
}
inline Fp6Var f6add(const Fp6Var& a, const Fp6Var& b) { return {f2add(a.c0, b.c0), f2add(a.c1, b.c1), f2add(a.c2, b.c2)}; }
inline Fp6Var f6sub(const Fp6Var& a, const Fp6Var& b) { return {f2sub(a.c0, b.c0), f2sub(a.c1, b.c1), f2sub(a.c2, b.c2)}; }
inline Fp6Var f6neg(const Fp6Var& a) { return {f2neg(a.c0), f2neg(a.c1), f2neg(a.c2)}; }
inline Fp6Var f6dbl(const Fp6Var& a) { return {f2dbl(a.c0), f2dbl(a.c1), f2dbl(a.c2)}; }
// multiply by v (Fp12's non-residue): (c0,c1,c2) -> (xi*c2, c0, c1)
inline Fp6Var f6mul_v(const Fp6Var& a) { return {f2mul_xi(a.c2), a.c0, a.c1}; }
inline Fp6Var f6mul(const Fp6Var& a, const Fp6Var& b) {
    Fp2Var v0 = f2mul(a.c0, b.c0);
    Fp2Var v1 = f2mul(a.c1, b.c1);
    Fp2Var v2 = f2mul(a.c2, b.c2);
    Fp2Var t0 = f2mul(f2add(a.c1, a.c2), f2add(b.c1, b.c2));
    Fp2Var c0 = f2add(f2mul_xi(f2sub(f2sub(t0, v1), v2)), v0);
    Fp2Var t1 = f2mul(f2add(a.c0, a.c1), f2add(b.c0, b.c1));
    Fp2Var c1 = f2add(f2sub(f2sub(t1, v0), v1), f2mul_xi(v2));
    Fp2Var t2 = f2mul(f2add(a.c0, a.c2), f2add(b.c0, b.c2));
    Fp2Var c2 = f2sub(f2add(f2sub(t2, v0), v1), v2);
    return {c0, c1, c2};
}
inline void f6mul_equals(const Fp6Var& a, const Fp6Var& b, const Fp6Var& r) {
    Fp2Var v0 = f2mul(a.c0, b.c0);
    Fp2Var v1 = f2mul(a.c1, b.c1);
    Fp2Var v2 = f2mul(a.c2, b.c2);
    Fp2Var nr_a12 = f2mul_xi(f2add(a.c1, a.c2));
    Fp2Var b12 = f2add(b.c1, b.c2);
    Fp2Var nr_v1 = f2mul_xi(v1), nr_v2 = f2mul_xi(v2);
    Fp2Var chk0 = f2add(f2add(f2sub(r.c0, v0), nr_v1), nr_v2);
    f2mul_equals(nr_a12, b12, chk0);
    Fp2Var a01 = f2add(a.c0, a.c1), b01 = f2add(b.c0, b.c1);
    Fp2Var chk1 = f2add(f2add(f2sub(r.c1, nr_v2), v0), v1);
    f2mul_equals(a01, b01, chk1);
    Fp2Var a02 = f2add(a.c0, a.c2), b02 = f2add(b.c0, b.c2);
    Fp2Var chk2 = f2add(f2sub(f2add(r.c2, v0), v1), v2);
    f2mul_equals(a02, b02, chk2);
}
// sparse: other = (0, c1, 0)
inline Fp6Var f6mul_by_0_c1_0(const Fp6Var& a, const Fp2Var& c1) {
    Fp2Var v1 = f2mul(a.c1, c1);
    Fp2Var a12 = f2add(a.c1, a.c2);
    Fp2Var a01 = f2add(a.c0, a.c1);
    Fp2Var t0 = f2mul(a12, c1);
    Fp2Var r0 = f2mul_xi(f2sub(t0, v1));
    Fp2Var t1 = f2mul(a01, c1);
    Fp2Var r1 = f2sub(t1, v1);
    return {r0, r1, v1};
}
// sparse: other = (c0, c1, 0)
inline Fp6Var f6mul_by_c0_c1_0(const Fp6Var& a, const Fp2Var& c0, const Fp2Var& c1) {
    Fp2Var v0 = f2mul(a.c0, c0);
    Fp2Var v1 = f2mul(a.c1, c1);
    Fp2Var a12 = f2add(a.c1, a.c2);
    Fp2Var a01 = f2add(a.c0, a.c1);
    Fp2Var a02 = f2add(a.c0, a.c2);
    Fp2Var b01 = f2add(c0, c1);
    Fp2Var t0 = f2mul(a12, c1);
    Fp2Var r0 = f2add(f2mul_xi(f2sub(t0, v1)), v0);
    Fp2Var t1 = f2mul(a01, b01);
    Fp2Var r1 = f2sub(f2sub(t1, v0), v1);
    Fp2Var t2 = f2mul(a02, c0);
    Fp2Var r2 = f2add(f2sub(t2, v0), v1);
    return {r0, r1, r2};
}
inline Bool f6is_eq(const Fp6Var& a, const Fp6Var& b) {
    Bool b0 = f2is_eq(a.c0, b.c0);
    Bool b1 = f2is_eq(a.c1, b.c1);
    Bool b2 = f2is_eq(a.c2, b.c2);
    Bool t = band(b0, b1);
    return band(t, b2);
}
inline Fp6Var f6frobenius(const Fp6Var& a, int power) {
    const FrobTables& T = frob_tables();
    Fp2Var c0 = f2frobenius(a.c0, power);
    Fp2Var c1 = f2mulc(f2frobenius(a.c1, power), T.f6c1[power % 6]);
    Fp2Var c2 = f2mulc(f2frobenius(a.c2, power), T.f6c2[power % 6]);
    return {c0, c1, c2};
}

// ------------------------------------------------------------------ Fp12Var = QuadExtVar<Fp6Var>
struct Fp12Var {
    Fp6Var c0, c1;
    Fp12 val() const { return {c0.val(), c1.val()}; }
};
inline Fp12Var f12const(const Fp12& v) { return {f6const(v.c0), f6const(v.c1)}; }
inline Fp12Var f12one() { return f12const(fp12_one()); }
inline Fp12Var f12mul(const Fp12Var& a, const Fp12Var& b) {
    Fp6Var v0 = f6mul(a.c0, b.c0);
    Fp6Var v1 = f6mul(a.c1, b.c1);
    Fp6Var s = f6mul(f6add(a.c1, a.c0), f6add(b.c0, b.c1));
    Fp6Var c1 = f6sub(f6sub(s, v0), v1);
    Fp6Var c0 = f6add(v0, f6mul_v(v1));
    return {c0, c1};
}
inline Fp12Var f12sqr(const Fp12Var& a) {
    Fp6Var v0 = f6sub(a.c0, a.c1);
    Fp6Var v3 = f6sub(a.c0, f6mul_v(a.c1));
    Fp6Var v2 = f6mul(a.c0, a.c1);
    Fp6Var t = f6mul(v0, v3);
    t = f6add(t, v2);
    Fp6Var c0 = f6add(t, f6mul_v(v2));
    Fp6Var c1 = f6dbl(v2);
    return {c0, c1};
}
inline Fp12Var f12conj(const Fp12Var& a) { return {a.c0, f6neg(a.c1)}; }  // unitary_inverse
inline Fp12Var f12inv(const Fp12Var& a) {
    Fp12 iv = fp12_inv(a.val());
    Fp6Var i0 = f6witness(iv.c0);
    Fp6Var i1 = f6witness(iv.c1);
    Fp12Var inv = {i0, i1};
    // QuadExtVar::mul_equals(self, inverse, one)
    Fp12Var one = f12one();
    Fp6Var v1 = f6mul(a.c1, inv.c1);
    Fp6Var nr_v1 = f6mul_v(v1);
    Fp6Var rhs = f6sub(one.c0, nr_v1);
    f6mul_equals(a.c0, inv.c0, rhs);
    Fp6Var a01 = f6add(a.c0, a.c1), b01 = f6add(inv.c0, inv.c1);
    Fp6Var tmp = f6add(f6add(f6sub(v1, nr_v1), one.c1), one.c0);
    f6mul_equals(a01, b01, tmp);
    return inv;
}
inline Fp12Var f12frobenius(const Fp12Var& a, int power) {
    const FrobTables& T = frob_tables();
    Fp6Var c0 = f6frobenius(a.c0, power);
    Fp6Var c1 = f6frobenius(a.c1, power);
    const Fp2& k = T.f12c1[power % 12];
    c1 = {f2mulc(c1.c0, k), f2mulc(c1.c1, k), f2mulc(c1.c2, k)};
    return {c0, c1};
}
// multiply by the sparse element (c0 = (c0, c1, 0), c1 = (0, d1, 0))
inline Fp12Var f12mul_by_014(const Fp12Var& f, const Fp2Var& c0, const Fp2Var& c1, const Fp2Var& d1) {
    Fp6Var v0 = f6mul_by_c0_c1_0(f.c0, c0, c1);
    Fp6Var v1 = f6mul_by_0_c1_0(f.c1, d1);
    Fp6Var new_c0 = f6add(f6mul_v(v1), v0);
    Fp6Var t = f6mul_by_c0_c1_0(f6add(f.c0, f.c1), c0, f2add(c1, d1));
    Fp6Var new_c1 = f6sub(f6sub(t, v0), v1);
    return {new_c0, new_c1};
}
inline Fp12Var f12cyclotomic_square(const Fp12Var& f) {
    const Fp2Var &z0 = f.c0.c0, &z4 = f.c0.c1, &z3 = f.c0.c2, &z2 = f.c1.c0, &z1 = f.c1.c1, &z5 = f.c1.c2;
    auto half = [](const Fp2Var& za, const Fp2Var& zb, Fp2Var& t_even, Fp2Var& t_odd) {
        Fp2Var tmp = f2mul(za, zb);
        Fp2Var tmp1 = f2add(za, zb);
        Fp2Var tmp2 = f2add(f2mul_xi(zb), za);
        Fp2Var tmp4 = f2add(f2mul_xi(tmp), tmp);
        Fp2Var prod = f2mul(tmp1, tmp2);
        t_even = f2sub(prod, tmp4);
        t_odd = f2dbl(tmp);
    };
    Fp2Var t0, t1, t2, t3, t4, t5;
    half(z0, z1, t0, t1);
    half(z2, z3, t2, t3);
    half(z4, z5, t4, t5);
    Fp2Var c0_c0 = f2add(f2dbl(f2sub(t0, z0)), t0);
    Fp2Var c1_c1 = f2add(f2dbl(f2add(t1, z1)), t1);
    Fp2Var xt5 = f2mul_xi(t5);
    Fp2Var c1_c0 = f2add(f2dbl(f2add(z2, xt5)), xt5);
    Fp2Var c0_c2 = f2add(f2dbl(f2sub(t4, z3)), t4);
    Fp2Var c0_c1 = f2add(f2dbl(f2sub(t2, z4)), t2);
    Fp2Var c1_c2 = f2add(f2dbl(f2add(t3, z5)), t3);
    return {{c0_c0, c0_c1, c0_c2}, {c1_c0, c1_c1, c1_c2}};
}
// NAF of |X| = 0xd201000000010000, least-significant digit first (ark-ff find_naf)
inline std::vector<int8_t> naf_of_u64(uint64_t x) {
    std::vector<int8_t> res;
    u128 e = x;
    while (e != 0) {
        int8_t z = 0;
        if (e & 1) {
            z = 2 - (int8_t)(e % 4);
            if (z >= 0)
                e -= (u128)z;
            else
                e += (u128)(-z);
        }
        res.push_back(z);
        e >>= 1;
    }
    return res;
}
static const uint64_t BLS_X = 0xd201000000010000ULL;  // |x|, x is negative
inline Fp12Var f12optimized_cyclotomic_exp(const Fp12Var& f, uint64_t exponent) {
    Fp12Var res = f12one();
    Fp12Var f_inv = f12conj(f);
    bool found_nonzero = false;
    std::vector<int8_t> naf = naf_of_u64(exponent);
    for (int i = (int)naf.size() - 1; i >= 0; i--) {
        if (found_nonzero) res = f12cyclotomic_square(res);
        if (naf[i] != 0) {
            found_nonzero = true;
            res = f12mul(res, naf[i] > 0 ? f : f_inv);
        }
    }
    return res;
}
inline Fp12Var f12exp_by_x(const Fp12Var& f) { return f12conj(f12optimized_cyclotomic_exp(f, BLS_X)); }
inline Bool f12is_eq(const Fp12Var& a, const Fp12Var& b) {
    Bool b0 = f6is_eq(a.c0, b.c0);
    Bool b1 = f6is_eq(a.c1, b.c1);
    return band(b0, b1);
}

}  // namespace orc
