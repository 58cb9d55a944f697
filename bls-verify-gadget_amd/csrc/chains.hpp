// The witness programs ("chains") of the BLS-verify circuit, one instance per lane. Each chain fills one or
// more contiguous segments of the instance's witness vector, in arkworks allocation order.
//   chain_g1_alloc      constraints.rs:226 (G1Var::new_variable Witness) + :97-99 (pk != 0) + :119 (prepare_g1(pk))
//   chain_g2_alloc      constraints.rs:245 (G2Var::new_variable Witness)
//   chain_map_to_curve  hasher.rs:273-276 = map_to_curve_9mod16 (:352-502) + isogeny_map (:294-348)
//   chain_cofactor      hasher.rs:656 (Q0 + Q1) + clear_cofactor2 (:664-673)
//   chain_prepare_g2    constraints.rs:118,120 (G2PreparedVar::from_group_var)
//   chain_pairing       constraints.rs:121-127 (miller_loop, final_exponentiation, is_one)
// Segment sizes are fixed by the circuit shape (layout.h); tests pin them against the CPU oracle.
#pragma once
#include "curve.hpp"
#include "tower.hpp"

namespace blsw {

BLSW_HD bool bit_of(const uint32_t* words, int i) { return (words[i >> 5] >> (i & 31)) & 1; }

// [is_not_equal, multiplier] for diff = self - other whose inverse (or 0) the caller already has
BLSW_FN bool fp_is_eq_pre_w(Emitter& e, const Fp& diff, const Fp& diff_inv) {
    bool ne = !fp_is_zero(diff);
    e.put_bool(ne);
    e.put(ne ? diff_inv : fp_one());
    return !ne;
}

// ------------------------------------------------------------------------------------------------ G1
struct G1ChainOut {
    Fp ax, ay;  // prepare_g1(pk): affine coordinates (after the infinity select)
};
// G1Var::new_variable(Witness): native g*(h^-1 mod r), allocation, in-circuit multiplication by the cofactor
BLSW_FN Proj<OpsFp> chain_g1_alloc_only(Emitter e_alloc, const Fp& pkx, const Fp& pky) {
    constexpr uint32_t H1[4] = BLSW_H1_WORDS;
    constexpr uint32_t H1INV[8] = BLSW_H1INV_WORDS;
    bool inf = fp_is_zero(pkx) && fp_is_zero(pky);
    // native: pre = pk * (h^-1 mod r)   [G1Affine::mul_by_cofactor_inv]
    Jac1 acc = {pkx, pky, fp_one()};
#pragma unroll 1
    for (int i = BLSW_H1INV_NBITS - 2; i >= 0; i--) {
        acc = jac1_dbl(acc);
        if (bit_of(H1INV, i)) acc = jac1_add_mixed(acc, pkx, pky);
    }
    Fp zi = fp_inv(acc.z);
    Fp zi2 = fp_sqr(zi);
    Fp px = fp_mul(acc.x, zi2), py = fp_mul(acc.y, fp_mul(zi2, zi));
    Proj<OpsFp> ge;
    ge.x = inf ? fp_zero() : px;
    ge.y = inf ? fp_one() : py;
    ge.z = inf ? fp_zero() : fp_one();
    e_alloc.put(ge.x);
    e_alloc.put(ge.y);
    e_alloc.put(ge.z);
    return proj_mul_bits_be_w<OpsFp>(e_alloc, ge, H1, BLSW_H1_NBITS);
}
// pk.enforce_not_equal(G1Var::zero()) (constraints.rs:97-99) and prepare_g1(pk) (constraints.rs:119) on a variable pk
BLSW_FN G1ChainOut chain_g1_post(Emitter e_notzero, Emitter e_prep, const Proj<OpsFp>& pk) {
    Fp nz = fp_neg(pk.z);
    Fp nzi = fp_inv(nz);
    bool x_eq = fp_is_eq_pre_w(e_notzero, fp_zero(), fp_zero());  // (x*0) vs (0*z)
    bool y_eq = fp_is_eq_pre_w(e_notzero, nz, nzi);                // (y*0) vs (1*z): diff = 0 - z
    bool coords_eq = x_eq && y_eq;
    e_notzero.put_bool(coords_eq);
    bool z_is_zero = fp_is_eq_pre_w(e_notzero, nz, nzi);  // pk.is_zero(): zero.is_eq(z)
    e_notzero.put_bool(!z_is_zero && !coords_eq);          // or(both_zero, coords_eq) via and(ne_z, !coords_eq)
    // prepare_g1(pk) = to_affine
    bool infinity = fp_is_eq_pre_w(e_prep, nz, nzi);
    Fp z_inv = fp_neg(nzi);  // 1/z, or 0 when z = 0
    e_prep.put(z_inv);
    Fp nzx = fp_mul_w(e_prep, pk.x, z_inv);
    Fp nzy = fp_mul_w(e_prep, pk.y, z_inv);
    G1ChainOut o;
    o.ax = fp_select_w(e_prep, infinity, fp_zero(), nzx);
    o.ay = fp_select_w(e_prep, infinity, fp_zero(), nzy);
    return o;
}
BLSW_FN G1ChainOut chain_g1_alloc(Emitter e_alloc, Emitter e_notzero, Emitter e_prep, const Fp& pkx, const Fp& pky) {
    Proj<OpsFp> pk = chain_g1_alloc_only(e_alloc, pkx, pky);
    return chain_g1_post(e_notzero, e_prep, pk);
}
// mapped_aggregate (constraints.rs:169-191): count = UInt32 witness 0; per key: ret += bit.select(key, zero) and
// count = addmany(count, bit.select(1, 0)). K::ld(k) returns key k of this instance as a projective point.
template <class K>
BLSW_FN Proj<OpsFp> chain_mapped_aggregate(Emitter e_count, Emitter e_agg, const K& keys, const uint8_t* bitmap, uint32_t n_keys, uint32_t* count_out) {
    for (int i = 0; i < 32; i++) e_count.put_bool(false);  // UInt32::new_variable(|| Ok(0), Witness)
    Proj<OpsFp> ret = {fp_zero(), fp_one(), fp_zero()};
    uint32_t count = 0;
#pragma unroll 1
    for (uint32_t k = 0; k < n_keys; k++) {
        bool bit = bitmap[k] != 0;
        Proj<OpsFp> key = keys.ld(k);
        Proj<OpsFp> sel;
        sel.x = fp_select_w(e_agg, bit, key.x, fp_zero());
        sel.y = fp_select_w(e_agg, bit, key.y, fp_one());
        sel.z = fp_select_w(e_agg, bit, key.z, fp_zero());
        ret = (k == 0) ? sel : proj_add_w<OpsFp, 0>(e_agg, ret, sel);  // zero + sel returns sel (constant special case)
        uint64_t sum = (uint64_t)count + (bit ? 1u : 0u);
        for (int i = 0; i < 33; i++) e_agg.put_bool((sum >> i) & 1);  // addmany of two operands: 33 result bits
        count = (uint32_t)sum;
    }
    if (count_out) *count_out = count;
    return ret;
}

// ------------------------------------------------------------------------------------------------ G2 allocation
// the rest of the G2 allocation after the subgroup scalar multiplication: ge.enforce_equal(&ge) and the zero tests
BLSW_FN void chain_g2_alloc_tail(Emitter e, const Proj<OpsFp2>& ge) {
    // ge.enforce_equal(&ge)   (sic: ark-r1cs-std 0.4.0)
    Fp2 l0 = fp2_mul_w(e, ge.x, ge.z);
    Fp2 r0 = fp2_mul_w(e, ge.x, ge.z);
    bool x_eq = fp2_is_eq_w(e, l0, r0);
    Fp2 l1 = fp2_mul_w(e, ge.y, ge.z);
    Fp2 r1 = fp2_mul_w(e, ge.y, ge.z);
    bool y_eq = fp2_is_eq_w(e, l1, r1);
    bool coords_eq = x_eq && y_eq;
    e.put_bool(coords_eq);
    bool za = fp2_is_zero_w(e, ge.z);
    bool zb = fp2_is_zero_w(e, ge.z);
    bool both_zero = za && zb;
    e.put_bool(both_zero);
    e.put_bool(both_zero || coords_eq);
}

BLSW_FN void chain_g2_alloc(Emitter e, const Fp2& sx, const Fp2& sy) {
    constexpr uint32_t RM1[8] = BLSW_RM1_WORDS;
    bool inf = fp2_is_zero(sx) && fp2_is_zero(sy);
    Proj<OpsFp2> ge;
    ge.x = inf ? fp2_zero() : sx;
    ge.y = inf ? fp2_one() : sy;
    ge.z = inf ? fp2_zero() : fp2_one();
    e.put(ge.x.c0);
    e.put(ge.x.c1);
    e.put(ge.y.c0);
    e.put(ge.y.c1);
    e.put(ge.z.c0);
    e.put(ge.z.c1);
    (void)proj_mul_bits_be_w<OpsFp2>(e, ge, RM1, BLSW_RM1_NBITS);
    chain_g2_alloc_tail(e, ge);
}

// ------------------------------------------------------------------------------------------------ map_to_curve
BLSW_FN Fp2 sswu_pow_c1_w(Emitter& e, const Fp2& v) {
    constexpr uint32_t C1[24] = BLSW_SSWU_C1_WORDS;
    // bits 759, 758 are 0 and bit 757 is the first 1: r stays the constant one, then becomes 1*v (no witness)
    Fp2 r = v;
#pragma unroll 1
    for (int i = BLSW_SSWU_C1_NBITS - 4; i >= 0; i--) {
        r = fp2_sqr_w(e, r);
        if (bit_of(C1, i)) r = fp2_mul_w(e, r, v);
    }
    return r;
}
// sgn0 (hasher.rs:520-530)
BLSW_FN bool sswu_sgn0_w(Emitter& e, const Fp2& v) {
    bool sign_0 = fp_to_bits_le_w(e, v.c0);
    bool sign_1 = fp_to_bits_le_w(e, v.c1);
    bool zero_0 = fp_is_eq_w(e, fp_zero(), v.c0);
    bool r = zero_0 && sign_1;
    e.put_bool(r);
    bool s = sign_0 || r;
    e.put_bool(s);
    return s;
}
BLSW_FN Fp2 poly_step_w(Emitter& e, Fp2& result, Fp2& curr_pow, const Fp2& coeff, const Fp2& point, bool pow_is_const_one) {
    // term = curr_pow * coeff (constant coefficient: no witness); curr_pow *= point (witness unless curr_pow == const 1)
    result = fp2_add(result, pow_is_const_one ? coeff : fp2_mul(curr_pow, coeff));
    curr_pow = pow_is_const_one ? point : fp2_mul_w(e, curr_pow, point);
    return result;
}
BLSW_FN Fp2 poly_eval4_w(Emitter& e, const Fp2& k0, const Fp2& k1, const Fp2& k2, const Fp2& k3, const Fp2& x) {
    Fp2 result = fp2_zero(), cp = fp2_one();
    poly_step_w(e, result, cp, k0, x, true);
    poly_step_w(e, result, cp, k1, x, false);
    poly_step_w(e, result, cp, k2, x, false);
    poly_step_w(e, result, cp, k3, x, false);
    return result;
}
BLSW_FN Fp2 poly_eval3_w(Emitter& e, const Fp2& k0, const Fp2& k1, const Fp2& k2, const Fp2& x) {
    Fp2 result = fp2_zero(), cp = fp2_one();
    poly_step_w(e, result, cp, k0, x, true);
    poly_step_w(e, result, cp, k1, x, false);
    poly_step_w(e, result, cp, k2, x, false);
    return result;
}
BLSW_FN Proj<OpsFp2> chain_map_to_curve(Emitter e, const Fp2& u) {
    const Fp2 Z = K_SSWU_Z(), A = K_SSWU_A(), B = K_SSWU_B(), C2 = K_SSWU_C2(), C3 = K_SSWU_C3(), C4 = K_SSWU_C4(), C5 = K_SSWU_C5();
    Fp2 tv1 = fp2_sqr_w(e, u);                       // 1
    Fp2 tv3 = fp2_mul(Z, tv1);                       // 2
    Fp2 tv5 = fp2_sqr_w(e, tv3);                     // 3
    Fp2 xd = fp2_add(tv5, tv3);                      // 4
    Fp2 x1n = fp2_add(xd, fp2_one());                // 5
    x1n = fp2_mul(x1n, B);                           // 6
    xd = fp2_mul(K_SSWU_NEG_A(), xd);                // 7
    bool e1 = fp2_is_zero_w(e, xd);                  // 8
    xd = fp2_select_w(e, e1, K_SSWU_ZA(), xd);       // 9
    Fp2 tv2 = fp2_sqr_w(e, xd);                      // 10
    Fp2 gxd = fp2_mul_w(e, tv2, xd);                 // 11
    tv2 = fp2_mul(A, tv2);                           // 12
    Fp2 gx1 = fp2_add(fp2_sqr_w(e, x1n), tv2);       // 13, 14
    gx1 = fp2_mul_w(e, gx1, x1n);                    // 15
    tv2 = fp2_mul(B, gxd);                           // 16
    gx1 = fp2_add(gx1, tv2);                         // 17
    Fp2 tv4 = fp2_sqr_w(e, gxd);                     // 18
    tv2 = fp2_mul_w(e, tv4, gxd);                    // 19
    tv4 = fp2_sqr_w(e, tv4);                         // 20
    tv2 = fp2_mul_w(e, tv2, tv4);                    // 21
    tv2 = fp2_mul_w(e, tv2, gx1);                    // 22
    tv4 = fp2_sqr_w(e, tv4);                         // 23
    tv4 = fp2_mul_w(e, tv2, tv4);                    // 24
    Fp2 y = sswu_pow_c1_w(e, tv4);                   // 25
    y = fp2_mul_w(e, y, tv2);                        // 26
    tv4 = fp2_mul(y, C2);                            // 27
    tv2 = fp2_sqr_w(e, tv4);                         // 28
    tv2 = fp2_mul_w(e, tv2, gxd);                    // 29
    bool e2 = fp2_is_eq_w(e, tv2, gx1);              // 30
    y = fp2_select_w(e, e2, tv4, y);                 // 31
    tv4 = fp2_mul(y, C3);                            // 32
    tv2 = fp2_sqr_w(e, tv4);                         // 33
    tv2 = fp2_mul_w(e, tv2, gxd);                    // 34
    bool e3 = fp2_is_eq_w(e, tv2, gx1);              // 35
    y = fp2_select_w(e, e3, tv4, y);                 // 36
    tv4 = fp2_mul(tv4, C2);                          // 37
    tv2 = fp2_sqr_w(e, tv4);                         // 38
    tv2 = fp2_mul_w(e, tv2, gxd);                    // 39
    bool e4 = fp2_is_eq_w(e, tv2, gx1);              // 40
    y = fp2_select_w(e, e4, tv4, y);                 // 41
    Fp2 gx2 = fp2_mul_w(e, gx1, tv5);                // 42
    gx2 = fp2_mul_w(e, gx2, tv3);                    // 43
    tv5 = fp2_mul_w(e, y, tv1);                      // 44
    tv5 = fp2_mul_w(e, tv5, u);                      // 45
    tv1 = fp2_mul(tv5, C4);                          // 46
    tv4 = fp2_mul(tv1, C2);                          // 47
    tv2 = fp2_sqr_w(e, tv4);                         // 48
    tv2 = fp2_mul_w(e, tv2, gxd);                    // 49
    bool e5 = fp2_is_eq_w(e, tv2, gx2);              // 50
    tv1 = fp2_select_w(e, e5, tv4, tv1);             // 51
    tv4 = fp2_mul(tv5, C5);                          // 52
    tv2 = fp2_sqr_w(e, tv4);                         // 53
    tv2 = fp2_mul_w(e, tv2, gxd);                    // 54
    bool e6 = fp2_is_eq_w(e, tv2, gx2);              // 55
    tv1 = fp2_select_w(e, e6, tv4, tv1);             // 56
    tv4 = fp2_mul(tv4, C2);                          // 57
    tv2 = fp2_sqr_w(e, tv4);                         // 58
    tv2 = fp2_mul_w(e, tv2, gxd);                    // 59
    bool e7 = fp2_is_eq_w(e, tv2, gx2);              // 60
    tv1 = fp2_select_w(e, e7, tv4, tv1);             // 61
    tv2 = fp2_sqr_w(e, y);                           // 62
    tv2 = fp2_mul_w(e, tv2, gxd);                    // 63
    bool e8 = fp2_is_eq_w(e, tv2, gx1);              // 64
    y = fp2_select_w(e, e8, y, tv1);                 // 65  CMOV(tv1, y, e8)
    tv2 = fp2_mul_w(e, tv3, x1n);                    // 66
    Fp2 xn = fp2_select_w(e, e8, x1n, tv2);          // 67  CMOV(tv2, x1n, e8)
    bool sgn0_u = sswu_sgn0_w(e, u);                 // 68
    bool sgn0_y = sswu_sgn0_w(e, y);
    e.put_bool(sgn0_u ^ sgn0_y);                     //     e9 = !(xor witness)
    bool e9 = !(sgn0_u ^ sgn0_y);
    y = fp2_select_w(e, e9, y, fp2_neg(y));          // 69  CMOV(-y, y, e9)
    // to_projective_short(xd, xn, y)  (hasher.rs:551-559)
    Fp2 xd2 = fp2_sqr_w(e, xd);
    Fp2 xd3 = fp2_mul_w(e, xd2, xd);
    Fp2 jx = fp2_mul_w(e, xn, xd);
    Fp2 jy = fp2_mul_w(e, y, xd3);
    Fp2 jz = xd;
    // isogeny_map (hasher.rs:294-348)
    bool is_infinity = fp2_is_zero_w(e, jz);
    Fp2 z_inv = fp2_inv_w(e, jz);  // to_affine_unchecked (hasher.rs:569-583)
    Fp2 z_inv_2 = fp2_sqr_w(e, z_inv);
    Fp2 z_inv_3 = fp2_mul_w(e, z_inv_2, z_inv);
    Fp2 ax = fp2_mul_w(e, jx, z_inv_2);
    Fp2 ay = fp2_mul_w(e, jy, z_inv_3);
    Fp2 x_den = poly_eval3_w(e, K_ISO_XDEN0(), K_ISO_XDEN1(), K_ISO_XDEN2(), ax);
    Fp2 x_den_inv = fp2_inv_w(e, x_den);
    Fp2 y_den = poly_eval4_w(e, K_ISO_YDEN0(), K_ISO_YDEN1(), K_ISO_YDEN2(), K_ISO_YDEN3(), ax);
    Fp2 y_den_inv = fp2_inv_w(e, y_den);
    Fp2 x_num = poly_eval4_w(e, K_ISO_XNUM0(), K_ISO_XNUM1(), K_ISO_XNUM2(), K_ISO_XNUM3(), ax);
    Fp2 y_num = poly_eval4_w(e, K_ISO_YNUM0(), K_ISO_YNUM1(), K_ISO_YNUM2(), K_ISO_YNUM3(), ax);
    Fp2 img_x = fp2_mul_w(e, x_num, x_den_inv);
    Fp2 t = fp2_mul_w(e, y_num, ay);
    Fp2 img_y = fp2_mul_w(e, t, y_den_inv);
    Proj<OpsFp2> q;
    q.x = fp2_select_w(e, is_infinity, fp2_zero(), img_x);
    q.y = fp2_select_w(e, is_infinity, fp2_zero(), img_y);
    q.z = is_infinity ? fp2_zero() : fp2_one();  // select between two constants: a linear combination
    return q;
}

// ------------------------------------------------------------------------------------------------ to_affine (G2)
struct Aff2Inf {
    Fp2 x, y;
    bool infinity;
};
BLSW_FN Aff2Inf g2_to_affine_w(Emitter& e, const Proj<OpsFp2>& p) {
    bool infinity = fp2_is_zero_w(e, p.z);
    Fp2 z_inv = fp2_inv(p.z);
    e.put(z_inv.c0);
    e.put(z_inv.c1);
    fp_mul_w(e, z_inv.c1, p.z.c1);  // z_inv.mul_equals(z, from(!infinity)): v1 = z_inv.c1 * z.c1
    Fp2 nzx = fp2_mul_w(e, p.x, z_inv);
    Fp2 nzy = fp2_mul_w(e, p.y, z_inv);
    Aff2Inf a;
    a.x = fp2_select_w(e, infinity, fp2_zero(), nzx);
    a.y = fp2_select_w(e, infinity, fp2_zero(), nzy);
    a.infinity = infinity;
    return a;
}

// ------------------------------------------------------------------------------------------------ Q0 + Q1, clear_cofactor2
// ZSTATE of a projective value: 0 = z is a variable, 2 = z is the constant one
BLSW_FN Proj<OpsFp2> proj_add_zstate_w(Emitter& e, const Proj<OpsFp2>& a, int za, const Proj<OpsFp2>& b, int zb) {
    if (za == 2 && zb == 2) return proj_add_w<OpsFp2, 2>(e, a, b);
    if (zb == 2) return proj_add_w<OpsFp2, 1>(e, a, b);
    if (za == 2) return proj_add_w<OpsFp2, 1>(e, b, a);
    return proj_add_w<OpsFp2, 0>(e, a, b);
}
BLSW_FN Proj<OpsFp2> chain_cofactor(Emitter e_add, Emitter e, const Proj<OpsFp2>& q0, const Proj<OpsFp2>& q1) {
    constexpr uint32_t HE[20] = BLSW_H_EFF_WORDS;
    Proj<OpsFp2> r = proj_add_w<OpsFp2, 0>(e_add, q0, q1);
    // scalar_mul_le(h_eff bits): to_affine, then 255-bit chunks of affine double-and-add
    Aff2Inf ra = g2_to_affine_w(e, r);
    Aff2 mopt = {ra.x, ra.y};
    Proj<OpsFp2> mul_result = {fp2_zero(), fp2_one(), fp2_zero()};
    int mr_state = -1;  // -1: the constant zero; 0: z variable; 2: z constant one
    const int nbits = BLSW_H_EFF_NBITS;
#pragma unroll 1
    for (int off = 0; off < nbits; off += 255) {
        int n = nbits - off < 255 ? nbits - off : 255;
        int split = n < 253 ? n : 253;
        Aff2 acc = mopt;
        Aff2 init = mopt;
        mopt = nz_double_w(e, mopt);
#pragma unroll 1
        for (int i = 1; i < split; i++) {
            const bool add = bit_of(HE, off + i);  // a constant of the circuit: uniform over the wave
            Fp2 inv_add, inv_dbl;
            if (add) {
                // the addition's and the doubling's slope denominators are both known here: one shared inversion
                fp2_inv2_inl(fp2_sub(mopt.x, acc.x), fp2_dbl(mopt.y), inv_add, inv_dbl);
                acc = nz_add_unchecked_pre_inl(e, acc, mopt, inv_add);
            } else {
                inv_dbl = fp2_inv_inl(fp2_dbl(mopt.y));
            }
            mopt = nz_double_pre_inl(e, mopt, inv_dbl);
        }
        Proj<OpsFp2> diff = {acc.x, acc.y, fp2_one()};
        int diff_state = 2;
        if (!bit_of(HE, off)) {  // subtract the initial accumulator value
            Proj<OpsFp2> neg_init = {init.x, fp2_neg(init.y), fp2_one()};
            diff = proj_add_w<OpsFp2, 2>(e, diff, neg_init);
            diff_state = 0;
        }
        if (mr_state < 0) {
            mul_result = diff;
            mr_state = diff_state;
        } else {
            mul_result = proj_add_zstate_w(e, mul_result, mr_state, diff, diff_state);
            mr_state = 0;
        }
#pragma unroll 1
        for (int i = split; i < n; i++) {
            if (bit_of(HE, off + i)) {
                Proj<OpsFp2> m = {mopt.x, mopt.y, fp2_one()};
                mul_result = proj_add_zstate_w(e, mul_result, mr_state, m, 2);
                mr_state = 0;
            }
            mopt = nz_double_w(e, mopt);
        }
    }
    // infinity.select(zero = (0,1,0), mul_result)
    Proj<OpsFp2> h;
    h.x = fp2_select_w(e, ra.infinity, fp2_zero(), mul_result.x);
    h.y = fp2_select_w(e, ra.infinity, fp2_one(), mul_result.y);
    if (mr_state == 2)
        h.z = ra.infinity ? fp2_zero() : fp2_one();
    else
        h.z = fp2_select_w(e, ra.infinity, fp2_zero(), mul_result.z);
    return h;
}

// ------------------------------------------------------------------------------------------------ prepare_g2
// coeffs: 68 pairs (c0, c1) handed to `out.st(4k + j, .)` in the order c0.c0, c0.c1, c1.c0, c1.c1
struct CoeffLinear {  // plain array (host harness)
    Fp* p;
    BLSW_FN void st(uint32_t idx, const Fp& v) const { p[idx] = v; }
    BLSW_FN Fp ld(uint32_t idx) const { return p[idx]; }
};
// QuadExtVar::inverse with the inverse supplied (gadgets.hpp: fp2_inv_w): witnesses inv.c0, inv.c1, a.c1 * inv.c1
BLSW_HD Fp2 fp2_inv_pre_w(Emitter& e, const Fp2& a, const Fp2& inv) {
    e.put(inv.c0);
    e.put(inv.c1);
    fp_mul_w(e, a.c1, inv.c1);
    return inv;
}
// one doubling / addition step of G2PreparedVar::from_group_var on the running affine point r, the inverse of the slope's
// denominator supplied (`ry_inv` = 1 / r.y; `dx_inv` = 1 / (q.x - r.x)): 16 / 14 witnesses, the step's line coefficients to out[4 k ..]
template <class C>
BLSW_HD void prepare_dbl_step(Emitter& e, Fp2& rx, Fp2& ry, const Fp2& ry_inv, const C& out, uint32_t k) {
    const Fp two_inv = K_TWO_INV();
    Fp2 a = fp2_inv_pre_w(e, ry, ry_inv);
    Fp2 b = fp2_sqr_w(e, rx);
    b = fp2_add(fp2_mul_fp(b, two_inv), b);
    Fp2 c = fp2_mul_w(e, a, b);
    Fp2 x3 = fp2_sub(fp2_sqr_w(e, c), fp2_dbl(rx));
    Fp2 cx = fp2_mul_w(e, c, rx);
    Fp2 ee = fp2_sub(cx, ry);
    Fp2 c_x3 = fp2_mul_w(e, c, x3);
    Fp2 y3 = fp2_sub(ee, c_x3);
    Fp2 f = fp2_neg(c);
    rx = x3;
    ry = y3;
    out.st(4 * k + 0, ee.c0);
    out.st(4 * k + 1, ee.c1);
    out.st(4 * k + 2, f.c0);
    out.st(4 * k + 3, f.c1);
}
template <class C>
BLSW_HD void prepare_add_step(Emitter& e, const Fp2& qx, const Fp2& qy, Fp2& rx, Fp2& ry, const Fp2& dx_inv, const C& out, uint32_t k) {
    Fp2 a = fp2_inv_pre_w(e, fp2_sub(qx, rx), dx_inv);
    Fp2 b = fp2_sub(qy, ry);
    Fp2 c = fp2_mul_w(e, a, b);
    Fp2 x3 = fp2_sub(fp2_sqr_w(e, c), fp2_add(rx, qx));
    Fp2 ee = fp2_mul_w(e, fp2_sub(rx, x3), c);
    Fp2 y3 = fp2_sub(ee, ry);
    Fp2 cr = fp2_mul_w(e, c, rx);
    Fp2 g = fp2_sub(cr, ry);
    Fp2 f = fp2_neg(c);
    rx = x3;
    ry = y3;
    out.st(4 * k + 0, g.c0);
    out.st(4 * k + 1, g.c1);
    out.st(4 * k + 2, f.c0);
    out.st(4 * k + 3, f.c1);
}
template <class C>
BLSW_FN void chain_prepare_g2(Emitter e, const Proj<OpsFp2>& q_, const C& out) {
    Aff2Inf q = g2_to_affine_w(e, q_);
    Fp2 rx = q.x, ry = q.y;
    uint32_t k = 0;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        prepare_dbl_step(e, rx, ry, fp2_inv(ry), out, k++);
        if ((BLSW_X_ABS >> i) & 1) prepare_add_step(e, q.x, q.y, rx, ry, fp2_inv(fp2_sub(q.x, rx)), out, k++);
    }
}

// ------------------------------------------------------------------------------------------------ pairing
// ell for the pair (-g1 constant, sig): c1 * p.x is a linear combination, d1 = (p.y, 0) constant
template <class C>
BLSW_FN Fp12 ell_const_p_w(Emitter& e, const Fp12& f, const C& coeff, uint32_t k, bool f_is_const) {
    Fp2 c0 = {coeff.ld(4 * k + 0), coeff.ld(4 * k + 1)}, c1 = {coeff.ld(4 * k + 2), coeff.ld(4 * k + 3)};
    const Fp px = K_G1_GEN_X(), py = K_G1_GEN_NEG_Y();
    c1 = fp2_mul_fp(c1, px);
    if (f_is_const) return fp12_mul_by_014_const_f(f, c0, c1, py);
    return fp12_mul_by_014_w<false>(e, f, c0, c1, py);
}
// ell for the pair (pk variable, H(m)): c1.c0*p.x and c1.c1*p.x are witnesses, d1 = (p.y, 0) with p.y variable
template <class C>
BLSW_FN Fp12 ell_var_p_w(Emitter& e, const Fp12& f, const C& coeff, uint32_t k, const Fp& px, const Fp& py) {
    Fp2 c0 = {coeff.ld(4 * k + 0), coeff.ld(4 * k + 1)}, c1 = {coeff.ld(4 * k + 2), coeff.ld(4 * k + 3)};
    Fp k0 = fp_mul_w(e, c1.c0, px);
    Fp k1 = fp_mul_w(e, c1.c1, px);
    c1 = {k0, k1};
    return fp12_mul_by_014_w<true>(e, f, c0, c1, py);
}
BLSW_FN bool fp6_is_eq_w(Emitter& e, const Fp6& self, const Fp6& other) {
    bool b0 = fp2_is_eq_w(e, self.c0, other.c0);
    bool b1 = fp2_is_eq_w(e, self.c1, other.c1);
    bool b2 = fp2_is_eq_w(e, self.c2, other.c2);
    bool t = b0 && b1;
    e.put_bool(t);
    bool r = t && b2;
    e.put_bool(r);
    return r;
}
// miller_loop([-g1, pk], [sig, H])
template <class C>
BLSW_FN Fp12 chain_miller(Emitter e, const Fp& pkx, const Fp& pky, const C& coeff_sig, const C& coeff_h) {
    Fp12 f = fp12_one();
    uint32_t k = 0;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        bool first = (i == 62);
        if (!first) f = fp12_sqr_w(e, f);
        f = ell_const_p_w(e, f, coeff_sig, k, first);
        f = ell_var_p_w(e, f, coeff_h, k, pkx, pky);
        k++;
        if ((BLSW_X_ABS >> i) & 1) {
            f = ell_const_p_w(e, f, coeff_sig, k, false);
            f = ell_var_p_w(e, f, coeff_h, k, pkx, pky);
            k++;
        }
    }
    return fp12_conj(f);
}
// miller_loop over K + 1 pairs, single-lane statement of team_miller_multi: P::pk(j, x, y) = prepare_g1(pk_j),
// P::coeff_h(j) = line coefficients of prepare_g2(H(m_j))
template <class C, class P>
BLSW_FN Fp12 chain_miller_multi(Emitter e, uint32_t K, const P& pairs, const C& coeff_sig) {
    Fp12 f = fp12_one();
    uint32_t k = 0;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        const int reps = ((BLSW_X_ABS >> i) & 1) ? 2 : 1;
        for (int rep = 0; rep < reps; rep++) {
            bool first = (i == 62 && rep == 0);
            if (rep == 0 && i != 62) f = fp12_sqr_w(e, f);
            f = ell_const_p_w(e, f, coeff_sig, k, first);
            for (uint32_t j = 0; j < K; j++) {
                Fp px, py;
                pairs.pk(j, px, py);
                f = ell_var_p_w(e, f, pairs.coeff_h(j), k, px, py);
            }
            k++;
        }
    }
    return fp12_conj(f);
}
// final_exponentiation . is_one
BLSW_FN bool chain_final_exp_is_one(Emitter e_fe, Emitter e_one, const Fp12& f) {
    // final exponentiation
    Emitter& g = e_fe;
    Fp12 f1 = fp12_conj(f);
    Fp12 f2 = fp12_inv_w(g, f);
    Fp12 r = fp12_mul_w(g, f1, f2);
    f2 = r;
    r = fp12_frobenius<2>(r);
    r = fp12_mul_w(g, r, f2);
    Fp12 y0 = fp12_conj(fp12_cyclotomic_square_w(g, r));
    Fp12 y5 = fp12_exp_by_x_w(g, r);
    Fp12 y1 = fp12_cyclotomic_square_w(g, y5);
    Fp12 y3 = fp12_mul_w(g, y0, y5);
    y0 = fp12_exp_by_x_w(g, y3);
    Fp12 y2 = fp12_exp_by_x_w(g, y0);
    Fp12 y4 = fp12_exp_by_x_w(g, y2);
    y4 = fp12_mul_w(g, y4, y1);
    y1 = fp12_exp_by_x_w(g, y4);
    y3 = fp12_conj(y3);
    y1 = fp12_mul_w(g, y1, y3);
    y1 = fp12_mul_w(g, y1, r);
    y3 = fp12_conj(r);
    y0 = fp12_mul_w(g, y0, r);
    y0 = fp12_frobenius<3>(y0);
    y4 = fp12_mul_w(g, y4, y3);
    y4 = fp12_frobenius<1>(y4);
    y5 = fp12_mul_w(g, y5, y2);
    y5 = fp12_frobenius<2>(y5);
    y5 = fp12_mul_w(g, y5, y0);
    y5 = fp12_mul_w(g, y5, y4);
    y5 = fp12_mul_w(g, y5, y1);
    // is_one: one.is_eq(result) componentwise
    Fp12 one = fp12_one();
    bool b0 = fp6_is_eq_w(e_one, one.c0, y5.c0);
    bool b1 = fp6_is_eq_w(e_one, one.c1, y5.c1);
    bool res = b0 && b1;
    e_one.put_bool(res);
    return res;
}

}  // namespace blsw
