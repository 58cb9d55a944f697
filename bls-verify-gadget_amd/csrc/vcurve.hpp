// Value-only curve arithmetic with inlined Fp2 operations (only the Fp product and the inversion are calls), for the entries that
// want group elements and no witnesses (blsw_hash_to_g2_batch, blsw_sign_batch). Compiles for the host as well: tests/hostsim runs
// the same functions against the CPU oracle.
#pragma once
#include "chains.hpp"
#include "decode.hpp"

namespace blsw {

// Value-only entry points (hash_to_g2 batch, signer): the same group element without the circuit's witness structure — the
// in-circuit clear_cofactor2 is an AFFINE double-and-add with one slope inversion per step (939 Fp2 inversions, App. A.5);
// here it is a Jacobian ladder over the 636 bits of h_eff with two inversions in all. Output as k_cofactor's: homogeneous (x, y, z).
// The ladder is written out with inlined Fp2 operations (only the Fp product and the inversion are calls), so that the kernel
// fits two waves per SIMD: the shared jac2_dbl / jac2_add_mixed are separate functions that take 248 VGPRs + 32 AGPRs each.
#ifdef BLSW_QUAD_DEV
BLSW_HD Fp2 v_sqr(const Fp2& a) {
    Fp prod;
    return fp2_sqr_quad(a, prod);
}
#else
BLSW_HD Fp2 v_sqr(const Fp2& a) {
    Fp v = fp_mul(a.c0, a.c1);
    Fp t = fp_mul(fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1));
    return {t, fp_dbl(v)};
}
#endif
BLSW_HD Jac2 v_dbl(const Jac2& p) {  // dbl-2009-l, a = 0
    Fp2 A = v_sqr(p.x), B = v_sqr(p.y), C = v_sqr(B);
    Fp2 D = fp2_dbl(fp2_sub(fp2_sub(v_sqr(fp2_add(p.x, B)), A), C));
    Fp2 E = fp2_add(fp2_dbl(A), A);
    Fp2 x3 = fp2_sub(v_sqr(E), fp2_dbl(D));
    Fp2 y3 = fp2_sub(fp2_mul_inl(E, fp2_sub(D, x3)), fp2_dbl(fp2_dbl(fp2_dbl(C))));
    Fp2 z3 = fp2_dbl(fp2_mul_inl(p.y, p.z));
    return {x3, y3, z3};
}
// (X, Y, Z) <- 2 (X, Y, Z), the formulas of v_dbl, for the value chains of cofactor_vf.hpp / prepare_vf.hpp. On a quad the 16 Fp products of the
// step (five Fp2 squares, two Fp2 products) are scheduled over the four lanes in FOUR rounds of one product per lane instead of one round per
// Fp2 operation (seven): X^2 | Y^2;  B^2 | (X + B)^2;  E^2 | Y0 Z0, Y1 Z1;  the three Karatsuba products of E (D - X3) | (Y0 + Y1)(Z0 + Z1).
BLSW_HD void v_dbl_inplace(Fp2& X, Fp2& Y, Fp2& Z) {
#ifdef BLSW_QUAD_DEV
    const uint32_t role = quad_role();
    const bool hi = role >= 2u, odd = (role & 1u) != 0;
    auto pick = [&](bool c, const Fp& t, const Fp& f) {  // on values (see quad_sel2)
        Fp r;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const uint32_t x = t.l[i], y = f.l[i];
            r.l[i] = c ? x : y;
        }
        return r;
    };
    // a pair of lanes squares v: the even lane v0 v1, the odd lane (v0 - v1)(v0 + v1)
    auto sq_operands = [&](const Fp2& lo, const Fp2& hi_v, Fp& a, Fp& b) {
        const Fp v0 = pick(hi, hi_v.c0, lo.c0), v1 = pick(hi, hi_v.c1, lo.c1);
        a = pick(odd, fp_sub(v0, v1), v0);
        b = pick(odd, fp_add(v0, v1), v1);
    };
    Fp a, b;
    sq_operands(X, Y, a, b);  // round 1: A = X^2 on lanes 0, 1; B = Y^2 on lanes 2, 3
    Fp p = fp_mul(a, b);
    const Fp2 A = {quad_bcast<1>(p), fp_dbl(quad_bcast<0>(p))}, B = {quad_bcast<3>(p), fp_dbl(quad_bcast<2>(p))};
    sq_operands(B, fp2_add(X, B), a, b);  // round 2: C = B^2; T = (X + B)^2
    p = fp_mul(a, b);
    const Fp2 C = {quad_bcast<1>(p), fp_dbl(quad_bcast<0>(p))}, T = {quad_bcast<3>(p), fp_dbl(quad_bcast<2>(p))};
    const Fp2 D = fp2_dbl(fp2_sub(fp2_sub(T, A), C));
    const Fp2 E = fp2_add(fp2_dbl(A), A);
    // round 3: F = E^2 on lanes 0, 1; Y0 Z0 on lane 2, Y1 Z1 on lane 3
    const Fp e_sum = fp_add(E.c0, E.c1);
    a = pick(hi, pick(odd, Y.c1, Y.c0), pick(odd, fp_sub(E.c0, E.c1), E.c0));
    b = pick(hi, pick(odd, Z.c1, Z.c0), pick(odd, e_sum, E.c1));
    p = fp_mul(a, b);
    const Fp2 F = {quad_bcast<1>(p), fp_dbl(quad_bcast<0>(p))};
    const Fp yz0 = quad_bcast<2>(p), yz1 = quad_bcast<3>(p);
    const Fp2 x3 = fp2_sub(F, fp2_dbl(D));
    const Fp2 W = fp2_sub(D, x3);
    // round 4: E W (Karatsuba) on lanes 0, 1, 2; (Y0 + Y1)(Z0 + Z1) on lane 3
    a = pick(hi, pick(odd, fp_add(Y.c0, Y.c1), e_sum), pick(odd, E.c1, E.c0));
    b = pick(hi, pick(odd, fp_add(Z.c0, Z.c1), fp_add(W.c0, W.c1)), pick(odd, W.c1, W.c0));
    p = fp_mul(a, b);
    const Fp m0 = quad_bcast<0>(p), m1 = quad_bcast<1>(p), ms = quad_bcast<2>(p), yzs = quad_bcast<3>(p);
    const Fp2 EW = {fp_sub(m0, m1), fp_sub(fp_sub(ms, m0), m1)};
    const Fp2 YZ = {fp_sub(yz0, yz1), fp_sub(fp_sub(yzs, yz0), yz1)};
    X = x3;
    Y = fp2_sub(EW, fp2_dbl(fp2_dbl(fp2_dbl(C))));
    Z = fp2_dbl(YZ);
#else
    const Fp2 A = v_sqr(X), B = v_sqr(Y), C = v_sqr(B);
    const Fp2 D = fp2_dbl(fp2_sub(fp2_sub(v_sqr(fp2_add(X, B)), A), C));
    const Fp2 E = fp2_add(fp2_dbl(A), A);
    const Fp2 x3 = fp2_sub(v_sqr(E), fp2_dbl(D));
    const Fp2 y3 = fp2_sub(fp2_mul_inl(E, fp2_sub(D, x3)), fp2_dbl(fp2_dbl(fp2_dbl(C))));
    Z = fp2_dbl(fp2_mul_inl(Y, Z));
    X = x3;
    Y = y3;
#endif
}
BLSW_HD Jac1v v1_dbl(const Jac1v& p) {
    Fp A = fp_sqr(p.x), B = fp_sqr(p.y), C = fp_sqr(B);
    Fp D = fp_dbl(fp_sub(fp_sub(fp_sqr(fp_add(p.x, B)), A), C));
    Fp E = fp_add(fp_dbl(A), A);
    Fp x3 = fp_sub(fp_sqr(E), fp_dbl(D));
    Fp y3 = fp_sub(fp_mul(E, fp_sub(D, x3)), fp_dbl(fp_dbl(fp_dbl(C))));
    Fp z3 = fp_dbl(fp_mul(p.y, p.z));
    return {x3, y3, z3};
}
BLSW_HD Jac1v v1_add_mixed(const Jac1v& p, const Fp& qx, const Fp& qy) {
    if (fp_is_zero(p.z)) return {qx, qy, fp_one()};
    Fp z1z1 = fp_sqr(p.z);
    Fp u2 = fp_mul(qx, z1z1);
    Fp s2 = fp_mul(fp_mul(qy, p.z), z1z1);
    Fp h = fp_sub(u2, p.x);
    Fp rr = fp_dbl(fp_sub(s2, p.y));
    if (fp_is_zero(h)) {
        if (fp_is_zero(rr)) return v1_dbl(p);
        return {fp_one(), fp_one(), fp_zero()};
    }
    Fp hh = fp_sqr(h);
    Fp i = fp_dbl(fp_dbl(hh));
    Fp j = fp_mul(h, i);
    Fp v = fp_mul(p.x, i);
    Fp x3 = fp_sub(fp_sub(fp_sqr(rr), j), fp_dbl(v));
    Fp y3 = fp_sub(fp_mul(rr, fp_sub(v, x3)), fp_dbl(fp_mul(p.y, j)));
    Fp z3 = fp_sub(fp_sub(fp_sqr(fp_add(p.z, h)), z1z1), hh);
    return {x3, y3, z3};
}
BLSW_HD Jac2 v_add_mixed(const Jac2& p, const Fp2& qx, const Fp2& qy) {  // madd-2007-bl; p = 0, p = +-q handled
    if (fp2_is_zero(p.z)) return {qx, qy, fp2_one()};
    Fp2 z1z1 = v_sqr(p.z);
    Fp2 u2 = fp2_mul_inl(qx, z1z1);
    Fp2 s2 = fp2_mul_inl(fp2_mul_inl(qy, p.z), z1z1);
    Fp2 h = fp2_sub(u2, p.x);
    Fp2 rr = fp2_dbl(fp2_sub(s2, p.y));
    if (fp2_is_zero(h)) {
        if (fp2_is_zero(rr)) return v_dbl(p);
        return {fp2_one(), fp2_one(), fp2_zero()};
    }
    Fp2 hh = v_sqr(h);
    Fp2 i = fp2_dbl(fp2_dbl(hh));
    Fp2 j = fp2_mul_inl(h, i);
    Fp2 v = fp2_mul_inl(p.x, i);
    Fp2 x3 = fp2_sub(fp2_sub(v_sqr(rr), j), fp2_dbl(v));
    Fp2 y3 = fp2_sub(fp2_mul_inl(rr, fp2_sub(v, x3)), fp2_dbl(fp2_mul_inl(p.y, j)));
    Fp2 z3 = fp2_sub(fp2_sub(v_sqr(fp2_add(p.z, h)), z1z1), hh);
    return {x3, y3, z3};
}
// map_to_curve_9mod16 + isogeny_map (hasher.rs:352-502, 294-348) for the value-only entries: the statements of
// chain_map_to_curve without the witness cursor, on the inlined Fp2 operations above (two waves per SIMD). Same field
// operations in the same order, so the result is the same element bit for bit.
BLSW_HD bool v_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
BLSW_HD Fp2 v_sel(bool c, const Fp2& a, const Fp2& b) { return c ? a : b; }
BLSW_HD bool v_sgn0(const Fp2& v) {  // hasher.rs:520-530
    const Fp c0 = fp_to_canonical(v.c0), c1 = fp_to_canonical(v.c1);
    return (c0.l[0] & 1) || (fp_is_zero(v.c0) && (c1.l[0] & 1));
}
BLSW_HD Fp2 v_poly(const Fp2* k, int n, const Fp2& x) {  // sum k[i] x^i, powers as DensePolynomialVar::evaluate builds them
    Fp2 result = k[0], cp = x;
    for (int i = 1; i < n; i++) {
        result = fp2_add(result, fp2_mul_inl(cp, k[i]));
        if (i + 1 < n) cp = fp2_mul_inl(cp, x);
    }
    return result;
}
// a^c1, c1 = (p^2 - 9) / 16 (hasher.rs:242): the circuit squares and multiplies over the 758 bits of c1 (every step is a witness);
// as a value, a^p = conj(a) in Fp2, so with c1 = e1 p + e0 the power is conj(a)^e1 * a^e0 — one joint ladder over 381 bits with
// the table {a, conj(a), a conj(a)} (the norm lies in Fp: two products instead of three). Same field element, canonical limbs.
BLSW_HD Fp2 v_pow_c1(const Fp2& a) {
    constexpr uint32_t E0[12] = BLSW_SSWU_E0_WORDS;
    constexpr uint32_t E1[12] = BLSW_SSWU_E1_WORDS;
    const Fp norm = fp_add(fp_sqr(a.c0), fp_sqr(a.c1));
    Fp2 r = fp2_one();
    bool started = false;
#pragma unroll 1
    for (int i = BLSW_SSWU_E_NBITS - 1; i >= 0; i--) {
        if (started) r = v_sqr(r);
        const uint32_t idx = ((E0[i >> 5] >> (i & 31)) & 1u) | (((E1[i >> 5] >> (i & 31)) & 1u) << 1);
        if (idx == 0) continue;
        if (!started) {
            r = idx == 1 ? a : (idx == 2 ? fp2_conj(a) : Fp2{norm, fp_zero()});
            started = true;
        } else if (idx == 3) {
            r = fp2_mul_fp(r, norm);
        } else {
            r = fp2_mul_inl(r, idx == 1 ? a : fp2_conj(a));
        }
    }
    return r;
}
BLSW_HD Proj<OpsFp2> v_map_to_curve(const Fp2& u) {
    const Fp2 Z = K_SSWU_Z(), A = K_SSWU_A(), B = K_SSWU_B(), C2 = K_SSWU_C2(), C3 = K_SSWU_C3(), C4 = K_SSWU_C4(), C5 = K_SSWU_C5();
    Fp2 tv1 = v_sqr(u);
    Fp2 tv3 = fp2_mul_inl(Z, tv1);
    Fp2 tv5 = v_sqr(tv3);
    Fp2 xd = fp2_add(tv5, tv3);
    Fp2 x1n = fp2_mul_inl(fp2_add(xd, fp2_one()), B);
    xd = fp2_mul_inl(K_SSWU_NEG_A(), xd);
    xd = v_sel(fp2_is_zero(xd), K_SSWU_ZA(), xd);
    Fp2 tv2 = v_sqr(xd);
    Fp2 gxd = fp2_mul_inl(tv2, xd);
    tv2 = fp2_mul_inl(A, tv2);
    Fp2 gx1 = fp2_add(v_sqr(x1n), tv2);
    gx1 = fp2_mul_inl(gx1, x1n);
    gx1 = fp2_add(gx1, fp2_mul_inl(B, gxd));
    Fp2 tv4 = v_sqr(gxd);
    tv2 = fp2_mul_inl(tv4, gxd);
    tv4 = v_sqr(tv4);
    tv2 = fp2_mul_inl(tv2, tv4);
    tv2 = fp2_mul_inl(tv2, gx1);
    tv4 = v_sqr(tv4);
    tv4 = fp2_mul_inl(tv2, tv4);
    Fp2 y = v_pow_c1(tv4);
    y = fp2_mul_inl(y, tv2);
    tv4 = fp2_mul_inl(y, C2);
    y = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx1), tv4, y);
    tv4 = fp2_mul_inl(y, C3);
    y = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx1), tv4, y);
    tv4 = fp2_mul_inl(tv4, C2);
    y = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx1), tv4, y);
    Fp2 gx2 = fp2_mul_inl(fp2_mul_inl(gx1, tv5), tv3);
    tv5 = fp2_mul_inl(fp2_mul_inl(y, tv1), u);
    tv1 = fp2_mul_inl(tv5, C4);
    tv4 = fp2_mul_inl(tv1, C2);
    tv1 = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx2), tv4, tv1);
    tv4 = fp2_mul_inl(tv5, C5);
    tv1 = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx2), tv4, tv1);
    tv4 = fp2_mul_inl(tv4, C2);
    tv1 = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx2), tv4, tv1);
    const bool e8 = v_eq(fp2_mul_inl(v_sqr(y), gxd), gx1);
    y = v_sel(e8, y, tv1);
    const Fp2 xn = v_sel(e8, x1n, fp2_mul_inl(tv3, x1n));
    const bool e9 = !(v_sgn0(u) ^ v_sgn0(y));
    y = v_sel(e9, y, fp2_neg(y));
    // to_projective_short (hasher.rs:551-559), to_affine_unchecked (:569-583), isogeny_map (:294-348)
    const Fp2 xd3 = fp2_mul_inl(v_sqr(xd), xd);
    const Fp2 jx = fp2_mul_inl(xn, xd), jy = fp2_mul_inl(y, xd3);
    const bool is_infinity = fp2_is_zero(xd);
    const Fp2 zi = fp2_inv_inl(xd), zi2 = v_sqr(zi);
    const Fp2 ax = fp2_mul_inl(jx, zi2), ay = fp2_mul_inl(jy, fp2_mul_inl(zi2, zi));
    const Fp2 kxd[3] = {K_ISO_XDEN0(), K_ISO_XDEN1(), K_ISO_XDEN2()};
    const Fp2 kyd[4] = {K_ISO_YDEN0(), K_ISO_YDEN1(), K_ISO_YDEN2(), K_ISO_YDEN3()};
    const Fp2 kxn[4] = {K_ISO_XNUM0(), K_ISO_XNUM1(), K_ISO_XNUM2(), K_ISO_XNUM3()};
    const Fp2 kyn[4] = {K_ISO_YNUM0(), K_ISO_YNUM1(), K_ISO_YNUM2(), K_ISO_YNUM3()};
    const Fp2 x_den_inv = fp2_inv_inl(v_poly(kxd, 3, ax));
    const Fp2 y_den_inv = fp2_inv_inl(v_poly(kyd, 4, ax));
    const Fp2 img_x = fp2_mul_inl(v_poly(kxn, 4, ax), x_den_inv);
    const Fp2 img_y = fp2_mul_inl(fp2_mul_inl(v_poly(kyn, 4, ax), ay), y_den_inv);
    Proj<OpsFp2> q;
    q.x = v_sel(is_infinity, fp2_zero(), img_x);
    q.y = v_sel(is_infinity, fp2_zero(), img_y);
    q.z = is_infinity ? fp2_zero() : fp2_one();
    return q;
}


// ---- cofactor clearing through the endomorphism psi (Budroni-Pintore; RFC 9380 App. G.4):
//     h_eff * P = [x^2 - x - 1] P + [x - 1] psi(P) + psi^2(2 P),   x = -0xd201000000010000
// two 64-bit ladders of weight 6 (63 doublings + 5 additions each) instead of the 636-bit ladder over h_eff (635 + 303). The
// reference's in-circuit clear_cofactor2 multiplies by h_eff (hasher.rs:664-673) and its native hash_to_g2 goes through arkworks'
// clear_cofactor, which is this formula (bls.rs:483-490); hasher.rs:1004-1026 asserts that both give the same point.
// general Jacobian addition (add-2007-bl); p = 0, q = 0, p = +-q handled
BLSW_HD Jac2 v_add(const Jac2& p, const Jac2& q) {
    if (fp2_is_zero(p.z)) return q;
    if (fp2_is_zero(q.z)) return p;
    const Fp2 z1z1 = v_sqr(p.z), z2z2 = v_sqr(q.z);
    const Fp2 u1 = fp2_mul_inl(p.x, z2z2), u2 = fp2_mul_inl(q.x, z1z1);
    const Fp2 s1 = fp2_mul_inl(fp2_mul_inl(p.y, q.z), z2z2), s2 = fp2_mul_inl(fp2_mul_inl(q.y, p.z), z1z1);
    const Fp2 h = fp2_sub(u2, u1), rr = fp2_dbl(fp2_sub(s2, s1));
    if (fp2_is_zero(h)) {
        if (fp2_is_zero(rr)) return v_dbl(p);
        return {fp2_one(), fp2_one(), fp2_zero()};
    }
    const Fp2 i = v_sqr(fp2_dbl(h)), j = fp2_mul_inl(h, i), v = fp2_mul_inl(u1, i);
    const Fp2 x3 = fp2_sub(fp2_sub(v_sqr(rr), j), fp2_dbl(v));
    const Fp2 y3 = fp2_sub(fp2_mul_inl(rr, fp2_sub(v, x3)), fp2_dbl(fp2_mul_inl(s1, j)));
    const Fp2 z3 = fp2_mul_inl(fp2_sub(fp2_sub(v_sqr(fp2_add(p.z, q.z)), z1z1), z2z2), h);
    return {x3, y3, z3};
}
BLSW_HD Jac2 v_neg(const Jac2& p) { return {p.x, fp2_neg(p.y), p.z}; }
// psi(X, Y, Z) = (c1 conj(X), c2 conj(Y), conj(Z));  psi^2(X, Y, Z) = (k2 X, -Y, Z)
BLSW_HD Jac2 v_psi(const Jac2& p) { return {fp2_mul_inl(fp2_conj(p.x), K_PSI_C1()), fp2_mul_inl(fp2_conj(p.y), K_PSI_C2()), fp2_conj(p.z)}; }
BLSW_HD Jac2 v_psi2(const Jac2& p) { return {fp2_mul_fp(p.x, K_PSI2_C1()), fp2_neg(p.y), p.z}; }
// PARK: a few Jacobian points per lane kept out of the registers (workspace rows on the device, an array on the host):
// st(slot, point), ld(slot). The whole computation is ONE loop over a step table with a single inlined copy of the doubling and
// of the general addition (each inlined copy of v_add holds two points and its temporaries: nine of them were 2 600 spilled
// registers). Steps: low 3 bits = kind, bits 3-4 = slot, bit 5 = operand through psi, bit 6 = operand negated.
enum : uint8_t { VS_DBL = 0, VS_ADD = 1, VS_NEG = 2, VS_PSI2 = 3, VS_ST = 4, VS_LD = 5, VS_SLOT = 8, VS_OPPSI = 32, VS_OPNEG = 64 };
struct VCofactorProgram {
    uint8_t step[160];
    uint32_t n;
};
constexpr VCofactorProgram v_cofactor_program() {
    VCofactorProgram p = {};
    uint32_t n = 0;
    p.step[n++] = VS_ST | 0 * VS_SLOT;  // park[0] = P
    for (int pass = 0; pass < 2; pass++) {
        const uint8_t slot = pass == 0 ? 0 : 2;
        for (int i = 62; i >= 0; i--) {  // acc = |x| * park[slot] (acc holds park[slot] on entry)
            p.step[n++] = VS_DBL;
            if ((BLSW_X_ABS >> i) & 1) p.step[n++] = VS_ADD | slot * VS_SLOT;
        }
        p.step[n++] = VS_NEG;  // x is negative
        if (pass == 0) {
            p.step[n++] = VS_ST | 1 * VS_SLOT;             // park[1] = t1 = x P
            p.step[n++] = VS_ADD | 0 * VS_SLOT | VS_OPPSI;  // t1 + psi(P)
            p.step[n++] = VS_ST | 2 * VS_SLOT;
        } else {
            p.step[n++] = VS_ST | 2 * VS_SLOT;  // park[2] = x (x P + psi(P))
        }
    }
    p.step[n++] = VS_LD | 0 * VS_SLOT;
    p.step[n++] = VS_DBL;
    p.step[n++] = VS_PSI2;                                      // psi^2(2 P)
    p.step[n++] = VS_ADD | 0 * VS_SLOT | VS_OPPSI | VS_OPNEG;  // - psi(P)
    p.step[n++] = VS_ADD | 2 * VS_SLOT;                        // + x (x P + psi(P))
    p.step[n++] = VS_ADD | 1 * VS_SLOT | VS_OPNEG;             // - x P
    p.step[n++] = VS_ADD | 0 * VS_SLOT | VS_OPNEG;             // - P
    p.n = n;
    return p;
}
template <class PARK>
BLSW_HD Jac2 v_clear_cofactor(const PARK& park, const Jac2& p) {
    constexpr VCofactorProgram prog = v_cofactor_program();
    Jac2 acc = p;
#pragma unroll 1
    for (uint32_t k = 0; k < prog.n; k++) {
        const uint32_t st = prog.step[k], kind = st & 7, slot = (st >> 3) & 3;
        if (kind == VS_DBL) {
            acc = v_dbl(acc);
        } else if (kind == VS_ADD) {
            Jac2 q = park.ld(slot);
            if (st & VS_OPPSI) q = v_psi(q);
            if (st & VS_OPNEG) q.y = fp2_neg(q.y);
            acc = v_add(acc, q);
        } else if (kind == VS_NEG) {
            acc.y = fp2_neg(acc.y);
        } else if (kind == VS_PSI2) {
            acc = v_psi2(acc);
        } else if (kind == VS_ST) {
            park.st(slot, acc);
        } else {
            acc = park.ld(slot);
        }
    }
    return acc;
}

}  // namespace blsw
