// Input decode, one point per lane: ZCash-format compressed G1 / G2 points as ark-bls12-381 ^0.4.0
// (de)serialises them (`PublicKey::try_from` / `Signature::try_from` -> deserialize_compressed, src/bls.rs:219-242,
// 316-339) -> affine Montgomery coordinates + a status code. Checks: flags, x < p, on curve, prime-order subgroup.
// Pinned by tests/test_cases/deserialization_G1/*.json (10) and deserialization_G2/*.json (12).
#pragma once
#include "constants.hpp"

namespace blsw {

#define BLSW_SQRT_EXP_WORDS                                                                                                               \
    {                                                                                                                                     \
        0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u, 0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, \
            0x8e5ff9a6u, 0x0680447au                                                                                                      \
    }
#define BLSW_R_WORDS                                                                                             \
    {                                                                                                            \
        0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u \
    }
enum { DEC_OK = 0, DEC_BAD_ENCODING = 1, DEC_NOT_ON_CURVE = 2, DEC_NOT_IN_SUBGROUP = 3, DEC_IDENTITY = 4 };

// 48 big-endian bytes (top three bits already masked by the caller) -> canonical limbs; false if >= p
BLSW_FN bool fp_from_be48(const uint8_t* in, uint8_t mask0, Fp& out_mont) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    constexpr uint32_t R2[12] = BLSW_R2_LIMBS;
    Fp c;
    for (int w = 0; w < 12; w++) {
        const uint8_t* b = in + 44 - 4 * w;
        uint32_t b0 = b[0];
        if (w == 11) b0 &= mask0;
        c.l[w] = (b0 << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    }
    uint32_t borrow = 0;
    for (int i = 0; i < 12; i++) (void)subb32(c.l[i], P[i], borrow);
    if (!borrow) return false;  // c >= p
    Fp r2;
    for (int i = 0; i < 12; i++) r2.l[i] = R2[i];
    out_mont = fp_mul(c, r2);
    return true;
}
// a^((p+1)/4); true iff a is a square (then out^2 == a)
BLSW_FN bool fp_sqrt(const Fp& a, Fp& out) {
    constexpr uint32_t E[12] = BLSW_SQRT_EXP_WORDS;
    Fp r = a;  // bit 378 is the top bit of the exponent
#pragma unroll 1
    for (int i = 377; i >= 0; i--) {
        r = fp_sqr(r);
        if ((E[i >> 5] >> (i & 31)) & 1) r = fp_mul(r, a);
    }
    out = r;
    return fp_eq(fp_sqr(r), a);
}
// canonical comparison a > -a  (ark-serialize "lexicographically largest" flag)
BLSW_FN bool fp_lex_largest(const Fp& a) {
    Fp ca = fp_to_canonical(a), cn = fp_to_canonical(fp_neg(a));
    for (int i = 11; i >= 0; i--) {
        if (ca.l[i] > cn.l[i]) return true;
        if (ca.l[i] < cn.l[i]) return false;
    }
    return false;
}
BLSW_FN int fp_cmp_canonical(const Fp& a, const Fp& b) {
    Fp ca = fp_to_canonical(a), cb = fp_to_canonical(b);
    for (int i = 11; i >= 0; i--) {
        if (ca.l[i] > cb.l[i]) return 1;
        if (ca.l[i] < cb.l[i]) return -1;
    }
    return 0;
}
BLSW_HD bool fp2_lex_largest(const Fp2& a) {  // c1 is the most significant component
    Fp2 n = fp2_neg(a);
    int c = fp_cmp_canonical(a.c1, n.c1);
    if (c != 0) return c > 0;
    return fp_cmp_canonical(a.c0, n.c0) > 0;
}
BLSW_FN bool fp2_sqrt(const Fp2& a, Fp2& out) {
    if (fp2_is_zero(a)) {
        out = a;
        return true;
    }
    Fp r;
    if (fp_is_zero(a.c1)) {
        if (fp_sqrt(a.c0, r)) {
            out = {r, fp_zero()};
            return true;
        }
        if (fp_sqrt(fp_neg(a.c0), r)) {
            out = {fp_zero(), r};
            return true;
        }
        return false;
    }
    Fp alpha;
    if (!fp_sqrt(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)), alpha)) return false;
    // two_inv in Montgomery form
    constexpr uint32_t TI[12] = {0x00015554u, 0x18040000u, 0x3ab00001u, 0x85500005u, 0x253c276fu, 0x633cb57cu,
                                 0x31ebb502u, 0x6e22d1ecu, 0xf2d14ca2u, 0xd3916126u, 0x1a006596u, 0x17fbb857u};
    Fp two_inv;
    for (int i = 0; i < 12; i++) two_inv.l[i] = TI[i];
    Fp delta = fp_mul(fp_add(a.c0, alpha), two_inv);
    Fp x0;
    if (!fp_sqrt(delta, x0)) {
        delta = fp_mul(fp_sub(a.c0, alpha), two_inv);
        if (!fp_sqrt(delta, x0)) return false;
    }
    Fp x1 = fp_mul(a.c1, fp_inv(fp_dbl(x0)));
    Fp2 rr = {x0, x1};
    Fp2 chk = fp2_sqr(rr);
    if (!(fp_eq(chk.c0, a.c0) && fp_eq(chk.c1, a.c1))) return false;
    out = rr;
    return true;
}

// ---- value-only Jacobian arithmetic over Fp2 (a = 0) for the G2 subgroup check
struct Jac2 {
    Fp2 x, y, z;
};
BLSW_FN Jac2 jac2_dbl(const Jac2& p) {
    Fp2 A = fp2_sqr(p.x), B = fp2_sqr(p.y), C = fp2_sqr(B);
    Fp2 t = fp2_add(p.x, B);
    Fp2 D = fp2_dbl(fp2_sub(fp2_sub(fp2_sqr(t), A), C));
    Fp2 E = fp2_add(fp2_dbl(A), A);
    Fp2 F = fp2_sqr(E);
    Fp2 x3 = fp2_sub(F, fp2_dbl(D));
    Fp2 c8 = fp2_dbl(fp2_dbl(fp2_dbl(C)));
    Fp2 y3 = fp2_sub(fp2_mul(E, fp2_sub(D, x3)), c8);
    Fp2 z3 = fp2_dbl(fp2_mul(p.y, p.z));
    return {x3, y3, z3};
}
// complete-enough mixed addition for the double-and-add ladder [r]P: handles p = identity, p = +-q
BLSW_FN Jac2 jac2_add_mixed(const Jac2& p, const Fp2& qx, const Fp2& qy) {
    if (fp2_is_zero(p.z)) return {qx, qy, fp2_one()};
    Fp2 z1z1 = fp2_sqr(p.z);
    Fp2 u2 = fp2_mul(qx, z1z1);
    Fp2 s2 = fp2_mul(fp2_mul(qy, p.z), z1z1);
    Fp2 h = fp2_sub(u2, p.x);
    Fp2 rr = fp2_dbl(fp2_sub(s2, p.y));
    if (fp2_is_zero(h)) {
        if (fp2_is_zero(rr)) return jac2_dbl(p);
        return {fp2_one(), fp2_one(), fp2_zero()};
    }
    Fp2 hh = fp2_sqr(h);
    Fp2 i = fp2_dbl(fp2_dbl(hh));
    Fp2 j = fp2_mul(h, i);
    Fp2 v = fp2_mul(p.x, i);
    Fp2 x3 = fp2_sub(fp2_sub(fp2_sqr(rr), j), fp2_dbl(v));
    Fp2 y3 = fp2_sub(fp2_mul(rr, fp2_sub(v, x3)), fp2_dbl(fp2_mul(p.y, j)));
    Fp2 z3 = fp2_sub(fp2_sub(fp2_sqr(fp2_add(p.z, h)), z1z1), hh);
    return {x3, y3, z3};
}
struct Jac1v {
    Fp x, y, z;
};
BLSW_FN Jac1v jac1v_dbl(const Jac1v& p) {
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
BLSW_FN Jac1v jac1v_add_mixed(const Jac1v& p, const Fp& qx, const Fp& qy) {
    if (fp_is_zero(p.z)) return {qx, qy, fp_one()};
    Fp z1z1 = fp_sqr(p.z);
    Fp u2 = fp_mul(qx, z1z1);
    Fp s2 = fp_mul(fp_mul(qy, p.z), z1z1);
    Fp h = fp_sub(u2, p.x);
    Fp rr = fp_dbl(fp_sub(s2, p.y));
    if (fp_is_zero(h)) {
        if (fp_is_zero(rr)) return jac1v_dbl(p);
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

// ---- prime-order subgroup membership of a point ON the curve (not the identity), as ark-bls12-381 0.4 decides it in
// is_in_correct_subgroup_assuming_on_curve (what PublicKey::try_from / Signature::try_from reach through deserialize_compressed): the
// endomorphism tests of Scott, eprint 2021/1130 — two / one 64-bit ladders by |x| instead of one 255-bit ladder by r. The [r]P == O ladders
// stay as the definition they are checked against (tests/hostsim: random curve points outside the subgroup, cofactor multiples inside).
BLSW_FN bool g1_in_subgroup_ladder(const Fp& px, const Fp& py) {
    constexpr uint32_t RW[8] = BLSW_R_WORDS;
    Jac1v acc = {px, py, fp_one()};
#pragma unroll 1
    for (int i = 253; i >= 0; i--) {
        acc = jac1v_dbl(acc);
        if ((RW[i >> 5] >> (i & 31)) & 1) acc = jac1v_add_mixed(acc, px, py);
    }
    return fp_is_zero(acc.z);
}
BLSW_FN bool g2_in_subgroup_ladder(const Fp2& px, const Fp2& py) {
    constexpr uint32_t RW[8] = BLSW_R_WORDS;
    Jac2 acc = {px, py, fp2_one()};
#pragma unroll 1
    for (int i = 253; i >= 0; i--) {
        acc = jac2_dbl(acc);
        if ((RW[i >> 5] >> (i & 31)) & 1) acc = jac2_add_mixed(acc, px, py);
    }
    return fp2_is_zero(acc.z);
}
// G1: phi(P) = (beta x, y) acts on G1 as multiplication by -x^2 (x^4 - x^2 + 1 = r), so P in G1 <=> phi(P) = -[x^2] P; arkworks' early exit
// [|x|]P == +-P -> not in the subgroup keeps the second ladder off degenerate inputs.
BLSW_FN bool g1_in_subgroup(const Fp& px, const Fp& py) {
    Jac1v t = {px, py, fp_one()};
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        t = jac1v_dbl(t);
        if ((BLSW_X_ABS >> i) & 1) t = jac1v_add_mixed(t, px, py);
    }
    if (fp_is_zero(t.z)) return false;  // [|x|]P = O: the order of P divides |x| < r
    // u = [|x|]P as an affine point, then [|x|]u = [x^2]P
    const Fp zi = fp_inv(t.z), zi2 = fp_sqr(zi);
    const Fp ux = fp_mul(t.x, zi2), uy = fp_mul(t.y, fp_mul(zi2, zi));
    if (fp_eq(ux, px)) return false;  // [|x|]P = +-P
    Jac1v w = {ux, uy, fp_one()};
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        w = jac1v_dbl(w);
        if ((BLSW_X_ABS >> i) & 1) w = jac1v_add_mixed(w, ux, uy);
    }
    if (fp_is_zero(w.z)) return false;
    // (beta px, py) == -(w.x / z^2, w.y / z^3)
    const Fp z2 = fp_sqr(w.z), z3 = fp_mul(z2, w.z);
    return fp_eq(fp_mul(fp_mul(px, K_G1_BETA()), z2), w.x) && fp_eq(fp_mul(py, z3), fp_neg(w.y));
}
// G2: psi = untwist-Frobenius-twist acts on G2 as multiplication by x (negative): P in G2 <=> psi(P) = [x] P = -[|x|] P
BLSW_FN bool g2_in_subgroup(const Fp2& px, const Fp2& py) {
    Jac2 t = {px, py, fp2_one()};
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        t = jac2_dbl(t);
        if ((BLSW_X_ABS >> i) & 1) t = jac2_add_mixed(t, px, py);
    }
    if (fp2_is_zero(t.z)) return false;
    const Fp2 qx = fp2_mul(fp2_conj(px), K_PSI_C1()), qy = fp2_mul(fp2_conj(py), K_PSI_C2());  // psi(P), affine
    const Fp2 z2 = fp2_sqr(t.z), z3 = fp2_mul(z2, t.z);
    const Fp2 lx = fp2_mul(qx, z2), ly = fp2_mul(qy, z3), ny = fp2_neg(t.y);
    return fp_eq(lx.c0, t.x.c0) && fp_eq(lx.c1, t.x.c1) && fp_eq(ly.c0, ny.c0) && fp_eq(ly.c1, ny.c1);
}

// G1: 48 bytes -> (x, y) Montgomery ((0,0) for the identity) + status
BLSW_FN int g1_decode(const uint8_t* in, Fp& x, Fp& y) {
    x = fp_zero();
    y = fp_zero();
    const bool c = in[0] >> 7, inf = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    if (sort && (!c || inf)) return DEC_BAD_ENCODING;
    if (!c) return DEC_BAD_ENCODING;
    if (inf) return DEC_IDENTITY;
    Fp px;
    if (!fp_from_be48(in, 0x1f, px)) return DEC_BAD_ENCODING;
    Fp rhs = fp_add(fp_mul(fp_sqr(px), px), fp_from_u32(4));
    Fp py;
    if (!fp_sqrt(rhs, py)) return DEC_NOT_ON_CURVE;
    if (fp_lex_largest(py) != sort) py = fp_neg(py);
    if (!g1_in_subgroup(px, py)) return DEC_NOT_IN_SUBGROUP;
    x = px;
    y = py;
    return DEC_OK;
}
// G2: 96 bytes = x.c1 || x.c0 -> (x.c0, x.c1, y.c0, y.c1) + status
BLSW_FN int g2_decode(const uint8_t* in, Fp2& x, Fp2& y) {
    x = fp2_zero();
    y = fp2_zero();
    const bool c = in[0] >> 7, inf = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    if (sort && (!c || inf)) return DEC_BAD_ENCODING;
    if (!c) return DEC_BAD_ENCODING;
    if (inf) return DEC_IDENTITY;
    Fp2 px;
    if (!fp_from_be48(in, 0x1f, px.c1)) return DEC_BAD_ENCODING;
    if (!fp_from_be48(in + 48, 0xff, px.c0)) return DEC_BAD_ENCODING;
    Fp four = fp_from_u32(4);
    Fp2 b = {four, four};
    Fp2 rhs = fp2_add(fp2_mul(fp2_sqr(px), px), b);
    Fp2 py;
    if (!fp2_sqrt(rhs, py)) return DEC_NOT_ON_CURVE;
    if (fp2_lex_largest(py) != sort) py = fp2_neg(py);
    if (!g2_in_subgroup(px, py)) return DEC_NOT_IN_SUBGROUP;
    x = px;
    y = py;
    return DEC_OK;
}

// ------------------------------------------------------------------------------------------------ signer
// Native BLS::sign / PublicKey::from(&sk) for a batch (src/bls.rs:411-425, 183-195): sig = sk * H(msg), pk = sk * g1,
// serialised as ark-serialize compressed points (ZCash flags). Value-only Jacobian ladders, one key per lane.
enum { SIGN_OK = 0, SIGN_BAD_ENCODING = 1, SIGN_INVALID_SECRET_KEY = 5 };

// 32 little-endian bytes (PrivateKey::try_from, bls.rs:97-103: deserialize_compressed of an Fr) -> 8 words.
// SIGN_BAD_ENCODING if >= r, SIGN_INVALID_SECRET_KEY if zero (bls.rs:417-419)
BLSW_FN int sk_from_le32(const uint8_t* in, uint32_t* w) {
    constexpr uint32_t RW[8] = BLSW_R_WORDS;
    uint32_t any = 0;
    for (int i = 0; i < 8; i++) {
        w[i] = (uint32_t)in[4 * i] | ((uint32_t)in[4 * i + 1] << 8) | ((uint32_t)in[4 * i + 2] << 16) | ((uint32_t)in[4 * i + 3] << 24);
        any |= w[i];
    }
    uint32_t borrow = 0;
    for (int i = 0; i < 8; i++) (void)subb32(w[i], RW[i], borrow);
    if (!borrow) return SIGN_BAD_ENCODING;
    if (!any) return SIGN_INVALID_SECRET_KEY;
    return SIGN_OK;
}
// canonical big-endian bytes of a Montgomery-form element
BLSW_FN void fp_to_be48(const Fp& a, uint8_t* out) {
    Fp c = fp_to_canonical(a);
    for (int w = 0; w < 12; w++) {
        uint8_t* b = out + 44 - 4 * w;
        b[0] = (uint8_t)(c.l[w] >> 24);
        b[1] = (uint8_t)(c.l[w] >> 16);
        b[2] = (uint8_t)(c.l[w] >> 8);
        b[3] = (uint8_t)c.l[w];
    }
}
BLSW_FN void g1_encode(const Fp& x, const Fp& y, bool infinity, uint8_t* out) {
    if (infinity) {
        for (int i = 0; i < 48; i++) out[i] = 0;
        out[0] = 0xc0;
        return;
    }
    fp_to_be48(x, out);
    out[0] |= 0x80 | (fp_lex_largest(y) ? 0x20 : 0);
}
BLSW_FN void g2_encode(const Fp2& x, const Fp2& y, bool infinity, uint8_t* out) {
    if (infinity) {
        for (int i = 0; i < 96; i++) out[i] = 0;
        out[0] = 0xc0;
        return;
    }
    fp_to_be48(x.c1, out);
    fp_to_be48(x.c0, out + 48);
    out[0] |= 0x80 | (fp2_lex_largest(y) ? 0x20 : 0);
}
// [k]Q for an affine Q in G2, k given as 8 words; returns affine (x, y); false if the result is the identity
BLSW_FN bool g2_mul_affine(const Fp2& qx, const Fp2& qy, const uint32_t* k, Fp2& rx, Fp2& ry) {
    Jac2 acc = {fp2_one(), fp2_one(), fp2_zero()};
#pragma unroll 1
    for (int i = 254; i >= 0; i--) {
        acc = jac2_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = jac2_add_mixed(acc, qx, qy);
    }
    if (fp2_is_zero(acc.z)) return false;
    Fp2 zi = fp2_inv(acc.z), zi2 = fp2_sqr(zi);
    rx = fp2_mul(acc.x, zi2);
    ry = fp2_mul(acc.y, fp2_mul(zi2, zi));
    return true;
}
BLSW_FN bool g1_mul_affine(const Fp& qx, const Fp& qy, const uint32_t* k, Fp& rx, Fp& ry) {
    Jac1v acc = {fp_one(), fp_one(), fp_zero()};
#pragma unroll 1
    for (int i = 254; i >= 0; i--) {
        acc = jac1v_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = jac1v_add_mixed(acc, qx, qy);
    }
    if (fp_is_zero(acc.z)) return false;
    Fp zi = fp_inv(acc.z), zi2 = fp_sqr(zi);
    rx = fp_mul(acc.x, zi2);
    ry = fp_mul(acc.y, fp_mul(zi2, zi));
    return true;
}


}  // namespace blsw
