// Fp6 = Fp2[v]/(v^3 - (1+u)), Fp12 = Fp6[w]/(w^2 - v) witness programs (one instance per lane), in the order
// ark-r1cs-std ^0.4.0 allocates them: fields/cubic_extension.rs, quadratic_extension.rs, fp6_3over2.rs, fp12.rs
// (SURVEY.md App. A.2). Every Var x Var Fp2 product is 3 Fp witnesses (Karatsuba), every Fp2 square is 2.
#pragma once
#include "constants.hpp"
#include "gadgets.hpp"

namespace blsw {

struct Fp6 {
    Fp2 c0, c1, c2;
};
struct Fp12 {
    Fp6 c0, c1;
};
BLSW_HD Fp6 fp6_add(const Fp6& a, const Fp6& b) { return {fp2_add(a.c0, b.c0), fp2_add(a.c1, b.c1), fp2_add(a.c2, b.c2)}; }
BLSW_HD Fp6 fp6_sub(const Fp6& a, const Fp6& b) { return {fp2_sub(a.c0, b.c0), fp2_sub(a.c1, b.c1), fp2_sub(a.c2, b.c2)}; }
BLSW_HD Fp6 fp6_neg(const Fp6& a) { return {fp2_neg(a.c0), fp2_neg(a.c1), fp2_neg(a.c2)}; }
BLSW_HD Fp6 fp6_dbl(const Fp6& a) { return {fp2_dbl(a.c0), fp2_dbl(a.c1), fp2_dbl(a.c2)}; }
BLSW_HD Fp6 fp6_mul_v(const Fp6& a) { return {fp2_mul_xi(a.c2), a.c0, a.c1}; }
BLSW_HD Fp6 fp6_zero() { return {fp2_zero(), fp2_zero(), fp2_zero()}; }
BLSW_HD Fp6 fp6_one() { return {fp2_one(), fp2_zero(), fp2_zero()}; }
BLSW_HD Fp12 fp12_one() { return {fp6_one(), fp6_zero()}; }
BLSW_HD Fp12 fp12_conj(const Fp12& a) { return {a.c0, fp6_neg(a.c1)}; }

// CubicExtVar mul: v0, v1, v2, (a1+a2)(b1+b2), (a0+a1)(b0+b1), (a0+a2)(b0+b2)   -> 18 Fp witnesses
BLSW_FN Fp6 fp6_mul_w(Emitter& e, const Fp6& a, const Fp6& b) {
    Fp2 v0 = fp2_mul_w(e, a.c0, b.c0);
    Fp2 v1 = fp2_mul_w(e, a.c1, b.c1);
    Fp2 v2 = fp2_mul_w(e, a.c2, b.c2);
    Fp2 t0 = fp2_mul_w(e, fp2_add(a.c1, a.c2), fp2_add(b.c1, b.c2));
    Fp2 c0 = fp2_add(fp2_mul_xi(fp2_sub(fp2_sub(t0, v1), v2)), v0);
    Fp2 t1 = fp2_mul_w(e, fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1));
    Fp2 c1 = fp2_add(fp2_sub(fp2_sub(t1, v0), v1), fp2_mul_xi(v2));
    Fp2 t2 = fp2_mul_w(e, fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2));
    Fp2 c2 = fp2_sub(fp2_add(fp2_sub(t2, v0), v1), v2);
    return {c0, c1, c2};
}
// value-only Fp6 product (constant operand in the circuit)
BLSW_FN Fp6 fp6_mul(const Fp6& a, const Fp6& b) {
    Fp2 v0 = fp2_mul(a.c0, b.c0), v1 = fp2_mul(a.c1, b.c1), v2 = fp2_mul(a.c2, b.c2);
    Fp2 t0 = fp2_mul(fp2_add(a.c1, a.c2), fp2_add(b.c1, b.c2));
    Fp2 t1 = fp2_mul(fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1));
    Fp2 t2 = fp2_mul(fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2));
    return {fp2_add(fp2_mul_xi(fp2_sub(fp2_sub(t0, v1), v2)), v0), fp2_add(fp2_sub(fp2_sub(t1, v0), v1), fp2_mul_xi(v2)),
            fp2_sub(fp2_add(fp2_sub(t2, v0), v1), v2)};
}
// QuadExtVar::mul_equals over Fp: only v1 = a.c1*b.c1 is a witness
BLSW_HD void fp2_mul_equals_w(Emitter& e, const Fp2& a, const Fp2& b) { fp_mul_w(e, a.c1, b.c1); }
// CubicExtVar::mul_equals: v0, v1, v2 (9 witnesses) then three Fp2 mul_equals (1 witness each)
BLSW_FN void fp6_mul_equals_w(Emitter& e, const Fp6& a, const Fp6& b) {
    fp2_mul_w(e, a.c0, b.c0);
    fp2_mul_w(e, a.c1, b.c1);
    fp2_mul_w(e, a.c2, b.c2);
    fp2_mul_equals_w(e, fp2_mul_xi(fp2_add(a.c1, a.c2)), fp2_add(b.c1, b.c2));
    fp2_mul_equals_w(e, fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1));
    fp2_mul_equals_w(e, fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2));
}
// Fp6Var::mul_by_c0_c1_0: 5 Fp2 products
BLSW_FN Fp6 fp6_mul_by_c0_c1_0_w(Emitter& e, const Fp6& a, const Fp2& c0, const Fp2& c1) {
    Fp2 v0 = fp2_mul_w(e, a.c0, c0);
    Fp2 v1 = fp2_mul_w(e, a.c1, c1);
    Fp2 t0 = fp2_mul_w(e, fp2_add(a.c1, a.c2), c1);
    Fp2 r0 = fp2_add(fp2_mul_xi(fp2_sub(t0, v1)), v0);
    Fp2 t1 = fp2_mul_w(e, fp2_add(a.c0, a.c1), fp2_add(c0, c1));
    Fp2 r1 = fp2_sub(fp2_sub(t1, v0), v1);
    Fp2 t2 = fp2_mul_w(e, fp2_add(a.c0, a.c2), c0);
    Fp2 r2 = fp2_add(fp2_sub(t2, v0), v1);
    return {r0, r1, r2};
}
BLSW_FN Fp6 fp6_mul_by_c0_c1_0(const Fp6& a, const Fp2& c0, const Fp2& c1) {  // value only (f constant)
    Fp2 v0 = fp2_mul(a.c0, c0), v1 = fp2_mul(a.c1, c1);
    Fp2 t0 = fp2_mul(fp2_add(a.c1, a.c2), c1);
    Fp2 t1 = fp2_mul(fp2_add(a.c0, a.c1), fp2_add(c0, c1));
    Fp2 t2 = fp2_mul(fp2_add(a.c0, a.c2), c0);
    return {fp2_add(fp2_mul_xi(fp2_sub(t0, v1)), v0), fp2_sub(fp2_sub(t1, v0), v1), fp2_add(fp2_sub(t2, v0), v1)};
}
// Fp6Var::mul_by_0_c1_0 with c1 = (y, 0), y in Fp.
//  WITNESS = false: y is a constant (pair (-g1, sig)): linear combinations only.
//  WITNESS = true : y is a variable (pair (pk, H)): each Fp2 product a*(y,0) is 2 witnesses: a.c0*y, (a.c0+a.c1)*y
template <bool WITNESS>
BLSW_FN Fp2 fp2_mul_by_fp_maybe_w(Emitter& e, const Fp2& a, const Fp& y) {
    if (WITNESS) {
        Fp v0 = fp_mul_w(e, a.c0, y);
        Fp s = fp_mul_w(e, fp_add(a.c0, a.c1), y);
        return {v0, fp_sub(s, v0)};
    }
    return fp2_mul_fp(a, y);
}
template <bool WITNESS>
BLSW_FN Fp6 fp6_mul_by_0_y_0(Emitter& e, const Fp6& a, const Fp& y) {
    Fp2 v1 = fp2_mul_by_fp_maybe_w<WITNESS>(e, a.c1, y);
    Fp2 t0 = fp2_mul_by_fp_maybe_w<WITNESS>(e, fp2_add(a.c1, a.c2), y);
    Fp2 r0 = fp2_mul_xi(fp2_sub(t0, v1));
    Fp2 t1 = fp2_mul_by_fp_maybe_w<WITNESS>(e, fp2_add(a.c0, a.c1), y);
    Fp2 r1 = fp2_sub(t1, v1);
    return {r0, r1, v1};
}
// Fp12Var::mul_by_014(c0, c1, d1 = (y,0))
template <bool YVAR>
BLSW_FN Fp12 fp12_mul_by_014_w(Emitter& e, const Fp12& f, const Fp2& c0, const Fp2& c1, const Fp& y) {
    Fp6 v0 = fp6_mul_by_c0_c1_0_w(e, f.c0, c0, c1);
    Fp6 v1 = fp6_mul_by_0_y_0<YVAR>(e, f.c1, y);
    Fp6 new_c0 = fp6_add(fp6_mul_v(v1), v0);
    Fp2 c1d1 = {fp_add(c1.c0, y), c1.c1};
    Fp6 t = fp6_mul_by_c0_c1_0_w(e, fp6_add(f.c0, f.c1), c0, c1d1);
    Fp6 new_c1 = fp6_sub(fp6_sub(t, v0), v1);
    return {new_c0, new_c1};
}
// same on a CONSTANT f (first Miller iteration, f = 1): no witnesses at all
BLSW_FN Fp12 fp12_mul_by_014_const_f(const Fp12& f, const Fp2& c0, const Fp2& c1, const Fp& y) {
    Emitter dummy = {nullptr, 0};
    Fp6 v0 = fp6_mul_by_c0_c1_0(f.c0, c0, c1);
    Fp6 v1 = fp6_mul_by_0_y_0<false>(dummy, f.c1, y);
    Fp6 new_c0 = fp6_add(fp6_mul_v(v1), v0);
    Fp2 c1d1 = {fp_add(c1.c0, y), c1.c1};
    Fp6 t = fp6_mul_by_c0_c1_0(fp6_add(f.c0, f.c1), c0, c1d1);
    Fp6 new_c1 = fp6_sub(fp6_sub(t, v0), v1);
    return {new_c0, new_c1};
}
// QuadExtVar::square over Fp6: v2 = c0*c1, then (c0-c1)*(c0 - v*c1)   -> 36 witnesses
BLSW_FN Fp12 fp12_sqr_w(Emitter& e, const Fp12& a) {
    Fp6 v0 = fp6_sub(a.c0, a.c1);
    Fp6 v3 = fp6_sub(a.c0, fp6_mul_v(a.c1));
    Fp6 v2 = fp6_mul_w(e, a.c0, a.c1);
    Fp6 t = fp6_mul_w(e, v0, v3);
    t = fp6_add(t, v2);
    return {fp6_add(t, fp6_mul_v(v2)), fp6_dbl(v2)};
}
// QuadExtVar mul over Fp6: v0, v1, (a0+a1)(b0+b1)   -> 54 witnesses
BLSW_FN Fp12 fp12_mul_w(Emitter& e, const Fp12& a, const Fp12& b) {
    Fp6 v0 = fp6_mul_w(e, a.c0, b.c0);
    Fp6 v1 = fp6_mul_w(e, a.c1, b.c1);
    Fp6 s = fp6_mul_w(e, fp6_add(a.c1, a.c0), fp6_add(b.c0, b.c1));
    return {fp6_add(v0, fp6_mul_v(v1)), fp6_sub(fp6_sub(s, v0), v1)};
}
// Granger-Scott cyclotomic square: 6 Fp2 products -> 18 witnesses
BLSW_FN void cyc_half_w(Emitter& e, const Fp2& za, const Fp2& zb, Fp2& t_even, Fp2& t_odd) {
    Fp2 tmp = fp2_mul_w(e, za, zb);
    Fp2 tmp1 = fp2_add(za, zb);
    Fp2 tmp2 = fp2_add(fp2_mul_xi(zb), za);
    Fp2 tmp4 = fp2_add(fp2_mul_xi(tmp), tmp);
    Fp2 prod = fp2_mul_w(e, tmp1, tmp2);
    t_even = fp2_sub(prod, tmp4);
    t_odd = fp2_dbl(tmp);
}
BLSW_FN Fp12 fp12_cyclotomic_square_w(Emitter& e, const Fp12& f) {
    const Fp2 &z0 = f.c0.c0, &z4 = f.c0.c1, &z3 = f.c0.c2, &z2 = f.c1.c0, &z1 = f.c1.c1, &z5 = f.c1.c2;
    Fp2 t0, t1, t2, t3, t4, t5;
    cyc_half_w(e, z0, z1, t0, t1);
    cyc_half_w(e, z2, z3, t2, t3);
    cyc_half_w(e, z4, z5, t4, t5);
    Fp2 c0_c0 = fp2_add(fp2_dbl(fp2_sub(t0, z0)), t0);
    Fp2 c1_c1 = fp2_add(fp2_dbl(fp2_add(t1, z1)), t1);
    Fp2 xt5 = fp2_mul_xi(t5);
    Fp2 c1_c0 = fp2_add(fp2_dbl(fp2_add(z2, xt5)), xt5);
    Fp2 c0_c2 = fp2_add(fp2_dbl(fp2_sub(t4, z3)), t4);
    Fp2 c0_c1 = fp2_add(fp2_dbl(fp2_sub(t2, z4)), t2);
    Fp2 c1_c2 = fp2_add(fp2_dbl(fp2_add(t3, z5)), t3);
    return {{c0_c0, c0_c1, c0_c2}, {c1_c0, c1_c1, c1_c2}};
}
// value-only inverses for the Fp12 inverse hint
BLSW_FN Fp6 fp6_inv(const Fp6& a) {
    Fp2 t0 = fp2_sub(fp2_sqr(a.c0), fp2_mul_xi(fp2_mul(a.c1, a.c2)));
    Fp2 t1 = fp2_sub(fp2_mul_xi(fp2_sqr(a.c2)), fp2_mul(a.c0, a.c1));
    Fp2 t2 = fp2_sub(fp2_sqr(a.c1), fp2_mul(a.c0, a.c2));
    Fp2 n = fp2_add(fp2_mul(a.c0, t0), fp2_mul_xi(fp2_add(fp2_mul(a.c2, t1), fp2_mul(a.c1, t2))));
    Fp2 ni = fp2_inv(n);
    return {fp2_mul(t0, ni), fp2_mul(t1, ni), fp2_mul(t2, ni)};
}
BLSW_FN Fp12 fp12_inv(const Fp12& a) {
    Fp6 n = fp6_sub(fp6_mul(a.c0, a.c0), fp6_mul_v(fp6_mul(a.c1, a.c1)));
    Fp6 ni = fp6_inv(n);
    return {fp6_mul(a.c0, ni), fp6_neg(fp6_mul(a.c1, ni))};
}
// QuadExtVar::inverse over Fp6: 12 witnesses (the inverse), then mul_equals(self, inverse, one):
// v1 = self.c1*inv.c1 (18), self.c0.mul_equals(inv.c0, .) (12), (a0+a1).mul_equals(b0+b1, .) (12)
BLSW_FN Fp12 fp12_inv_w(Emitter& e, const Fp12& a) {
    Fp12 inv = fp12_inv(a);
    const Fp2* parts[6] = {&inv.c0.c0, &inv.c0.c1, &inv.c0.c2, &inv.c1.c0, &inv.c1.c1, &inv.c1.c2};
    for (int i = 0; i < 6; i++) {
        e.put(parts[i]->c0);
        e.put(parts[i]->c1);
    }
    fp6_mul_w(e, a.c1, inv.c1);
    fp6_mul_equals_w(e, a.c0, inv.c0);
    fp6_mul_equals_w(e, fp6_add(a.c0, a.c1), fp6_add(inv.c0, inv.c1));
    return inv;
}
// Frobenius maps (constants only: linear combinations, no witnesses)
template <int POWER>
BLSW_HD Fp2 k_frob12_c1();
template <>
BLSW_HD Fp2 k_frob12_c1<1>() { return K_FROB12_C1_1(); }
template <>
BLSW_HD Fp2 k_frob12_c1<2>() { return K_FROB12_C1_2(); }
template <>
BLSW_HD Fp2 k_frob12_c1<3>() { return K_FROB12_C1_3(); }
template <int POWER>
BLSW_HD Fp2 k_frob6_c1();
template <>
BLSW_HD Fp2 k_frob6_c1<1>() { return K_FROB6_C1_1(); }
template <>
BLSW_HD Fp2 k_frob6_c1<2>() { return K_FROB6_C1_2(); }
template <>
BLSW_HD Fp2 k_frob6_c1<3>() { return K_FROB6_C1_3(); }
template <int POWER>
BLSW_HD Fp2 k_frob6_c2();
template <>
BLSW_HD Fp2 k_frob6_c2<1>() { return K_FROB6_C2_1(); }
template <>
BLSW_HD Fp2 k_frob6_c2<2>() { return K_FROB6_C2_2(); }
template <>
BLSW_HD Fp2 k_frob6_c2<3>() { return K_FROB6_C2_3(); }
template <int POWER>
BLSW_FN Fp6 fp6_frobenius(const Fp6& a) {
    Fp2 c0 = (POWER & 1) ? fp2_conj(a.c0) : a.c0;
    Fp2 c1 = (POWER & 1) ? fp2_conj(a.c1) : a.c1;
    Fp2 c2 = (POWER & 1) ? fp2_conj(a.c2) : a.c2;
    return {c0, fp2_mul(c1, k_frob6_c1<POWER>()), fp2_mul(c2, k_frob6_c2<POWER>())};
}
template <int POWER>
BLSW_FN Fp12 fp12_frobenius(const Fp12& a) {
    Fp6 c0 = fp6_frobenius<POWER>(a.c0);
    Fp6 c1 = fp6_frobenius<POWER>(a.c1);
    Fp2 k = k_frob12_c1<POWER>();
    return {c0, {fp2_mul(c1.c0, k), fp2_mul(c1.c1, k), fp2_mul(c1.c2, k)}};
}
// optimized_cyclotomic_exp(|X|) followed by the conjugation of exp_by_x.
// NAF(|X|) (LSB first) has length 65 with +1 at 16, 48, 57, 60, 64 and -1 at 62.
BLSW_FN Fp12 fp12_exp_by_x_w(Emitter& e, const Fp12& f) {
    const uint64_t plus = (1ull << 16) | (1ull << 48) | (1ull << 57) | (1ull << 60);
    const uint64_t minus = (1ull << 62);
    Fp12 f_inv = fp12_conj(f);
    Fp12 res = f;  // digit 64 (+1): one * f, a linear combination
#pragma unroll 1
    for (int i = 63; i >= 0; i--) {
        res = fp12_cyclotomic_square_w(e, res);
        if ((plus >> i) & 1)
            res = fp12_mul_w(e, res, f);
        else if ((minus >> i) & 1)
            res = fp12_mul_w(e, res, f_inv);
    }
    return fp12_conj(res);
}

}  // namespace blsw
