// Group-law witness programs, one instance per lane.
//  * ProjectiveVar complete formulas (Renes-Costello-Batina 2015, a = 0) as ark-r1cs-std ^0.4.0
//    groups/curves/short_weierstrass/mod.rs synthesises them: used by the Witness-mode
//    prime-order checks of G1Var/G2Var::new_variable (constraints.rs:206,226,245), by `Q0 + Q1`
//    (hasher.rs:656) and by the projective tail of scalar_mul_le (hasher.rs:672).
//  * NonZeroAffineVar incomplete affine double/add (non_zero_affine.rs) used by scalar_mul_le.
//  * value-only Jacobian arithmetic for the native `mul_by_cofactor_inv` that precedes G1 allocation.
#pragma once
#include "constants.hpp"
#include "gadgets.hpp"

namespace blsw {

struct OpsFp {
    typedef Fp F;
    static BLSW_HD F add(const F& a, const F& b) { return fp_add(a, b); }
    static BLSW_HD F sub(const F& a, const F& b) { return fp_sub(a, b); }
    static BLSW_HD F dbl(const F& a) { return fp_dbl(a); }
    static BLSW_HD F neg(const F& a) { return fp_neg(a); }
    static BLSW_HD F mul_w(Emitter& e, const F& a, const F& b) { return fp_mul_w(e, a, b); }
    static BLSW_HD F sqr_w(Emitter& e, const F& a) { return fp_mul_w(e, a, a); }
    static BLSW_HD F mul3b(const F& a) {  // * 12
        F a4 = fp_dbl(fp_dbl(a));
        return fp_add(fp_dbl(a4), a4);
    }
    static BLSW_HD F one() { return fp_one(); }
    static BLSW_HD F zero() { return fp_zero(); }
    static BLSW_HD F k3b() { return K_G1_3B(); }
};
struct OpsFp2 {
    typedef Fp2 F;
    static BLSW_HD F add(const F& a, const F& b) { return fp2_add(a, b); }
    static BLSW_HD F sub(const F& a, const F& b) { return fp2_sub(a, b); }
    static BLSW_HD F dbl(const F& a) { return fp2_dbl(a); }
    static BLSW_HD F neg(const F& a) { return fp2_neg(a); }
    static BLSW_HD F mul_w(Emitter& e, const F& a, const F& b) { return fp2_mul_w(e, a, b); }
    static BLSW_HD F sqr_w(Emitter& e, const F& a) { return fp2_sqr_w(e, a); }
    static BLSW_HD F mul3b(const F& a) {  // * 12(1+u)
        F x = fp2_mul_xi(a);
        F a4 = fp2_dbl(fp2_dbl(x));
        return fp2_add(fp2_dbl(a4), a4);
    }
    static BLSW_HD F one() { return fp2_one(); }
    static BLSW_HD F zero() { return fp2_zero(); }
    static BLSW_HD F k3b() { return K_G2_3B(); }
};

template <class O>
struct Proj {
    typename O::F x, y, z;
};

// ProjectiveVar::double_in_place on variables: 3 squarings + 8 products, in this order
template <class O>
BLSW_HD Proj<O> proj_double_inl(Emitter& e, const Proj<O>& p) {
    typedef typename O::F F;
    F xx = O::sqr_w(e, p.x);
    F yy = O::sqr_w(e, p.y);
    F zz = O::sqr_w(e, p.z);
    F xy2 = O::dbl(O::mul_w(e, p.x, p.y));
    F xz2 = O::dbl(O::mul_w(e, p.x, p.z));
    F bzz3 = O::mul3b(zz);
    F yy_m = O::sub(yy, bzz3);
    F yy_p = O::add(yy, bzz3);
    F y_frag = O::mul_w(e, yy_p, yy_m);
    F x_frag = O::mul_w(e, yy_m, xy2);
    F bxz3 = O::mul3b(xz2);
    F xx3 = O::add(O::dbl(xx), xx);
    F t = O::mul_w(e, xx3, bxz3);
    F y = O::add(y_frag, t);
    F yz2 = O::dbl(O::mul_w(e, p.y, p.z));
    F t2 = O::mul_w(e, bxz3, yz2);
    F x = O::sub(x_frag, t2);
    F z = O::dbl(O::dbl(O::mul_w(e, yz2, yy)));
    return {x, y, z};
}
// ProjectiveVar + ProjectiveVar, both variable. ZMODE: 0 = both z variable (12 products);
// 1 = z2 is the constant one (zz = z1 is a linear combination: 11 products);
// 2 = both z are the constant one (zz constant: 11 products).
template <class O, int ZMODE>
BLSW_HD Proj<O> proj_add_inl(Emitter& e, const Proj<O>& a, const Proj<O>& b) {
    typedef typename O::F F;
    F xx = O::mul_w(e, a.x, b.x);
    F yy = O::mul_w(e, a.y, b.y);
    F zz;
    if (ZMODE == 0)
        zz = O::mul_w(e, a.z, b.z);
    else if (ZMODE == 1)
        zz = a.z;
    else
        zz = O::one();
    F t0 = O::mul_w(e, O::add(a.x, a.y), O::add(b.x, b.y));
    F xy_pairs = O::sub(t0, O::add(xx, yy));
    F t1 = O::mul_w(e, O::add(a.x, a.z), O::add(b.x, b.z));
    F xz_pairs = O::sub(t1, O::add(xx, zz));
    F t2 = O::mul_w(e, O::add(a.y, a.z), O::add(b.y, b.z));
    F yz_pairs = O::sub(t2, O::add(yy, zz));
    F bzz3 = (ZMODE == 2) ? O::k3b() : O::mul3b(zz);
    F yy_m = O::sub(yy, bzz3);
    F yy_p = O::add(yy, bzz3);
    F xx3 = O::add(O::dbl(xx), xx);
    F bxz3 = O::mul3b(xz_pairs);
    F m0 = O::mul_w(e, yy_m, xy_pairs);
    F m1 = O::mul_w(e, yz_pairs, bxz3);
    F x = O::sub(m0, m1);
    F m2 = O::mul_w(e, yy_p, yy_m);
    F m3 = O::mul_w(e, xx3, bxz3);
    F y = O::add(m2, m3);
    F m4 = O::mul_w(e, yy_p, yz_pairs);
    F m5 = O::mul_w(e, xy_pairs, xx3);
    F z = O::add(m4, m5);
    return {x, y, z};
}
// out-of-line entry points (single uses); the scalar-multiplication loop below inlines both steps: a call per step would
// save and restore the callee-saved registers that hold the running point (scratch traffic that is written back to HBM)
template <class O>
BLSW_FN Proj<O> proj_double_w(Emitter& e, const Proj<O>& p) {
    return proj_double_inl<O>(e, p);
}
template <class O, int ZMODE>
BLSW_FN Proj<O> proj_add_w(Emitter& e, const Proj<O>& a, const Proj<O>& b) {
    return proj_add_inl<O, ZMODE>(e, a, b);
}
// result = [k] ge with `result = zero; for b in BE bits: double; if b: += ge` (the first set bit costs nothing:
// zero is a constant, so the first add returns ge itself)
template <class O>
BLSW_FN Proj<O> proj_mul_bits_be_w(Emitter& e, const Proj<O>& ge, const uint32_t* words, int nbits) {
    Proj<O> result = ge;
#pragma unroll 1
    for (int i = nbits - 2; i >= 0; i--) {
        result = proj_double_inl<O>(e, result);
        if ((words[i >> 5] >> (i & 31)) & 1) result = proj_add_inl<O, 0>(e, result, ge);
    }
    return result;
}

// ---- NonZeroAffineVar over Fp2
struct Aff2 {
    Fp2 x, y;
};
BLSW_FN Aff2 nz_double_w(Emitter& e, const Aff2& p) {
    Fp2 x1_sqr = fp2_sqr_w(e, p.x);
    Fp2 num = fp2_add(fp2_dbl(x1_sqr), x1_sqr);
    Fp2 den = fp2_dbl(p.y);
    Fp2 lambda = fp2_div_w(e, num, den);
    Fp2 l2 = fp2_sqr_w(e, lambda);
    Fp2 x3 = fp2_sub(l2, fp2_dbl(p.x));
    Fp2 t = fp2_mul_w(e, lambda, fp2_sub(p.x, x3));
    Fp2 y3 = fp2_sub(t, p.y);
    return {x3, y3};
}
BLSW_FN Aff2 nz_add_unchecked_w(Emitter& e, const Aff2& p, const Aff2& q) {
    Fp2 num = fp2_sub(q.y, p.y);
    Fp2 den = fp2_sub(q.x, p.x);
    Fp2 lambda = fp2_div_w(e, num, den);
    Fp2 l2 = fp2_sqr_w(e, lambda);
    Fp2 x3 = fp2_sub(fp2_sub(l2, p.x), q.x);
    Fp2 t = fp2_mul_w(e, lambda, fp2_sub(p.x, x3));
    Fp2 y3 = fp2_sub(t, p.y);
    return {x3, y3};
}

// variants with the slope denominator's inverse precomputed (see fp2_inv2): same witnesses as the two functions above
BLSW_FN Aff2 nz_double_pre_w(Emitter& e, const Aff2& p, const Fp2& den_inv) {
    Fp2 x1_sqr = fp2_sqr_w(e, p.x);
    Fp2 num = fp2_add(fp2_dbl(x1_sqr), x1_sqr);
    Fp2 den = fp2_dbl(p.y);
    Fp2 lambda = fp2_div_pre_w(e, num, den, den_inv);
    Fp2 l2 = fp2_sqr_w(e, lambda);
    Fp2 x3 = fp2_sub(l2, fp2_dbl(p.x));
    Fp2 t = fp2_mul_w(e, lambda, fp2_sub(p.x, x3));
    Fp2 y3 = fp2_sub(t, p.y);
    return {x3, y3};
}
BLSW_FN Aff2 nz_add_unchecked_pre_w(Emitter& e, const Aff2& p, const Aff2& q, const Fp2& den_inv) {
    Fp2 num = fp2_sub(q.y, p.y);
    Fp2 den = fp2_sub(q.x, p.x);
    Fp2 lambda = fp2_div_pre_w(e, num, den, den_inv);
    Fp2 l2 = fp2_sqr_w(e, lambda);
    Fp2 x3 = fp2_sub(fp2_sub(l2, p.x), q.x);
    Fp2 t = fp2_mul_w(e, lambda, fp2_sub(p.x, x3));
    Fp2 y3 = fp2_sub(t, p.y);
    return {x3, y3};
}

// the same two steps inlined into their caller's loop (clear_cofactor: 636 doublings + 304 additions per instance). As
// functions, each call saves and restores the callee-saved registers that hold the running points: measured as ~10 GB of
// scratch write-back per 16 384 instances and a third of the kernel's time.
BLSW_HD Fp2 fp2_div_pre_inl(Emitter& e, const Fp2& num, const Fp2& den, const Fp2& den_inv) {
    Fp2 r = fp2_mul_inl(num, den_inv);
    e.put(r.c0);
    e.put(r.c1);
    fp_mul_w(e, r.c1, den.c1);
    return r;
}
BLSW_HD Aff2 nz_double_pre_inl(Emitter& e, const Aff2& p, const Fp2& den_inv) {
    Fp2 x1_sqr = fp2_sqr_w(e, p.x);
    Fp2 num = fp2_add(fp2_dbl(x1_sqr), x1_sqr);
    Fp2 lambda = fp2_div_pre_inl(e, num, fp2_dbl(p.y), den_inv);
    Fp2 l2 = fp2_sqr_w(e, lambda);
    Fp2 x3 = fp2_sub(l2, fp2_dbl(p.x));
    Fp2 t = fp2_mul_w(e, lambda, fp2_sub(p.x, x3));
    return {x3, fp2_sub(t, p.y)};
}
BLSW_HD Aff2 nz_add_unchecked_pre_inl(Emitter& e, const Aff2& p, const Aff2& q, const Fp2& den_inv) {
    Fp2 lambda = fp2_div_pre_inl(e, fp2_sub(q.y, p.y), fp2_sub(q.x, p.x), den_inv);
    Fp2 l2 = fp2_sqr_w(e, lambda);
    Fp2 x3 = fp2_sub(fp2_sub(l2, p.x), q.x);
    Fp2 t = fp2_mul_w(e, lambda, fp2_sub(p.x, x3));
    return {x3, fp2_sub(t, p.y)};
}

// ---- value-only Jacobian arithmetic over Fp (a = 0), for g * (h^-1 mod r) before G1 allocation
struct Jac1 {
    Fp x, y, z;
};
BLSW_FN Jac1 jac1_dbl(const Jac1& p) {
    Fp A = fp_sqr(p.x), B = fp_sqr(p.y), C = fp_sqr(B);
    Fp t = fp_add(p.x, B);
    Fp D = fp_dbl(fp_sub(fp_sub(fp_sqr(t), A), C));
    Fp E = fp_add(fp_dbl(A), A);
    Fp F = fp_sqr(E);
    Fp x3 = fp_sub(F, fp_dbl(D));
    Fp c8 = fp_dbl(fp_dbl(fp_dbl(C)));
    Fp y3 = fp_sub(fp_mul(E, fp_sub(D, x3)), c8);
    Fp z3 = fp_dbl(fp_mul(p.y, p.z));
    return {x3, y3, z3};
}
// mixed addition p + (qx, qy); p must not be the identity and must differ from +-q (true for 1 < k < r)
BLSW_FN Jac1 jac1_add_mixed(const Jac1& p, const Fp& qx, const Fp& qy) {
    Fp z1z1 = fp_sqr(p.z);
    Fp u2 = fp_mul(qx, z1z1);
    Fp s2 = fp_mul(fp_mul(qy, p.z), z1z1);
    Fp h = fp_sub(u2, p.x);
    Fp hh = fp_sqr(h);
    Fp i = fp_dbl(fp_dbl(hh));
    Fp j = fp_mul(h, i);
    Fp r = fp_dbl(fp_sub(s2, p.y));
    Fp v = fp_mul(p.x, i);
    Fp x3 = fp_sub(fp_sub(fp_sqr(r), j), fp_dbl(v));
    Fp y3 = fp_sub(fp_mul(r, fp_sub(v, x3)), fp_dbl(fp_mul(p.y, j)));
    Fp z3 = fp_sub(fp_sub(fp_sqr(fp_add(p.z, h)), z1z1), hh);
    return {x3, y3, z3};
}

}  // namespace blsw
