// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// CPU restatement of the BLS12-381 base-field tower used by the reference
// (lightec-xyz/bls-verify-gadget). The arithmetic itself lives in third-party
// crates that are NOT vendored in /root/reference: ark-ff ^0.4.0 and
// ark-bls12-381 ^0.4.0 (Cargo.toml:17,22). This file restates their published
// algorithms (6x64-bit Montgomery Fp, Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3-(1+u)),
// Fp12 = Fp6[w]/(w^2-v)); constants are those of SURVEY.md App. B.
// Parity status: end results pinned by the reference's fixtures (tests/golden);
// witness layout vs real arkworks is UNPINNED.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <cassert>

namespace orc {

typedef unsigned __int128 u128;

struct Fp {
    uint64_t l[6];
};

static const uint64_t P_LIMBS[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                    0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t R1_LIMBS[6] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                                     0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
static const uint64_t R2_LIMBS[6] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                                     0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};
static const uint64_t P_INV = 0x89f3fffcfffcfffdULL;  // -p^{-1} mod 2^64

// global op counters (the roofline's algorithmic figure comes from here)
struct OpCount {
    uint64_t fp_mul = 0, fp_inv = 0, sha_blocks = 0;
};
inline OpCount& opcount() {
    static thread_local OpCount c;
    return c;
}

inline bool fp_is_zero(const Fp& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
inline bool fp_eq(const Fp& a, const Fp& b) { return memcmp(a.l, b.l, 48) == 0; }
inline Fp fp_zero() {
    Fp r;
    memset(r.l, 0, 48);
    return r;
}
inline Fp fp_one() {
    Fp r;
    memcpy(r.l, R1_LIMBS, 48);
    return r;
}
// raw compare of limb vectors: a >= b
inline bool limbs_geq(const uint64_t* a, const uint64_t* b) {
    for (int i = 5; i >= 0; i--) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}
inline uint64_t limbs_sub(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 6; i++) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
    }
    return borrow;
}
inline uint64_t limbs_add(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    uint64_t carry = 0;
    for (int i = 0; i < 6; i++) {
        u128 t = (u128)a[i] + b[i] + carry;
        r[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
    return carry;
}
inline Fp fp_add(const Fp& a, const Fp& b) {
    Fp r;
    limbs_add(r.l, a.l, b.l);  // p < 2^381 so no carry-out
    if (limbs_geq(r.l, P_LIMBS)) limbs_sub(r.l, r.l, P_LIMBS);
    return r;
}
inline Fp fp_sub(const Fp& a, const Fp& b) {
    Fp r;
    if (limbs_sub(r.l, a.l, b.l)) limbs_add(r.l, r.l, P_LIMBS);
    return r;
}
inline Fp fp_neg(const Fp& a) {
    if (fp_is_zero(a)) return a;
    Fp r;
    limbs_sub(r.l, P_LIMBS, a.l);
    return r;
}
inline Fp fp_dbl(const Fp& a) { return fp_add(a, a); }

// Montgomery multiplication (CIOS), result fully reduced. [ark-ff MontBackend::mul_assign semantics]
inline Fp fp_mul(const Fp& a, const Fp& b) {
    opcount().fp_mul++;
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 6; j++) {
            u128 x = (u128)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        u128 x = (u128)t[6] + c;
        t[6] = (uint64_t)x;
        t[7] = (uint64_t)(x >> 64);
        uint64_t m = t[0] * P_INV;
        x = (u128)m * P_LIMBS[0] + t[0];
        c = (uint64_t)(x >> 64);
        for (int j = 1; j < 6; j++) {
            x = (u128)m * P_LIMBS[j] + t[j] + c;
            t[j - 1] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        x = (u128)t[6] + c;
        t[5] = (uint64_t)x;
        t[6] = t[7] + (uint64_t)(x >> 64);
    }
    Fp r;
    memcpy(r.l, t, 48);
    if (t[6] || limbs_geq(r.l, P_LIMBS)) limbs_sub(r.l, r.l, P_LIMBS);
    return r;
}
inline Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }

inline Fp fp_from_raw(const uint64_t* limbs) {  // canonical integer (< p) -> Montgomery
    Fp a, r2;
    memcpy(a.l, limbs, 48);
    memcpy(r2.l, R2_LIMBS, 48);
    return fp_mul(a, r2);
}
inline void fp_to_raw(uint64_t* out, const Fp& a) {  // Montgomery -> canonical integer
    Fp one;
    memset(one.l, 0, 48);
    one.l[0] = 1;
    Fp r = fp_mul(a, one);
    memcpy(out, r.l, 48);
}
inline Fp fp_from_u64(uint64_t v) {
    uint64_t l[6] = {v, 0, 0, 0, 0, 0};
    return fp_from_raw(l);
}

// exponentiation by a little-endian limb array (square & multiply, MSB first)
inline Fp fp_pow(const Fp& a, const uint64_t* e, int nlimbs) {
    Fp r = fp_one();
    bool started = false;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        if (started) r = fp_sqr(r);
        if ((e[i / 64] >> (i % 64)) & 1) {
            r = started ? fp_mul(r, a) : a;
            started = true;
        }
    }
    return r;
}

// Fermat inverse (independent cross-check of fp_inv below)
inline Fp fp_inv_fermat(const Fp& a) {
    uint64_t e[6];
    memcpy(e, P_LIMBS, 48);
    e[0] -= 2;
    return fp_pow(a, e, 6);
}

// Binary extended Euclid on the Montgomery representation, as ark-ff's
// Fp::inverse does (Guajardo-Kumar-Paar-Pelzl Alg. 16). Returns 0 for 0.
inline Fp fp_inv(const Fp& a) {
    opcount().fp_inv++;
    if (fp_is_zero(a)) return a;
    uint64_t u[6], v[6];
    memcpy(u, a.l, 48);
    memcpy(v, P_LIMBS, 48);
    Fp b, c = fp_zero();
    memcpy(b.l, R2_LIMBS, 48);  // b = R^2 so that result is a^{-1} R
    auto is_one = [](const uint64_t* x) { return x[0] == 1 && (x[1] | x[2] | x[3] | x[4] | x[5]) == 0; };
    auto shr1 = [](uint64_t* x, uint64_t top) {
        for (int i = 0; i < 5; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
        x[5] = (x[5] >> 1) | (top << 63);
    };
    auto halve = [&](Fp& x) {
        if (x.l[0] & 1) {
            uint64_t carry = limbs_add(x.l, x.l, P_LIMBS);
            shr1(x.l, carry);
        } else
            shr1(x.l, 0);
    };
    while (!is_one(u) && !is_one(v)) {
        while ((u[0] & 1) == 0) {
            shr1(u, 0);
            halve(b);
        }
        while ((v[0] & 1) == 0) {
            shr1(v, 0);
            halve(c);
        }
        if (limbs_geq(u, v)) {
            limbs_sub(u, u, v);
            b = fp_sub(b, c);
        } else {
            limbs_sub(v, v, u);
            c = fp_sub(c, b);
        }
    }
    return is_one(u) ? b : c;
}

// canonical big-endian 48 bytes
inline void fp_to_bytes_be(uint8_t* out, const Fp& a) {
    uint64_t raw[6];
    fp_to_raw(raw, a);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 8; j++) out[47 - (i * 8 + j)] = (uint8_t)(raw[i] >> (8 * j));
}
// returns false if value >= p
inline bool fp_from_bytes_be(Fp& out, const uint8_t* in) {
    uint64_t raw[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 48; i++) raw[(47 - i) / 8] |= (uint64_t)in[i] << (8 * ((47 - i) % 8));
    if (limbs_geq(raw, P_LIMBS)) return false;
    out = fp_from_raw(raw);
    return true;
}
// arbitrary-length big-endian bytes reduced mod p (Horner)
inline Fp fp_from_be_bytes_mod_order(const uint8_t* in, size_t n) {
    Fp acc = fp_zero();
    Fp c256 = fp_from_u64(256);
    for (size_t i = 0; i < n; i++) acc = fp_add(fp_mul(acc, c256), fp_from_u64(in[i]));
    return acc;
}
inline Fp fp_from_hex(const char* hex) {  // big-endian hex, any length, reduced mod p
    Fp acc = fp_zero();
    Fp c16 = fp_from_u64(16);
    for (const char* s = hex; *s; s++) {
        char ch = *s;
        int v = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : (ch >= 'A' && ch <= 'F') ? ch - 'A' + 10 : -1;
        if (v < 0) continue;
        acc = fp_add(fp_mul(acc, c16), fp_from_u64((uint64_t)v));
    }
    return acc;
}
inline Fp fp_from_dec(const char* dec) {
    Fp acc = fp_zero();
    Fp c10 = fp_from_u64(10);
    for (const char* s = dec; *s; s++) acc = fp_add(fp_mul(acc, c10), fp_from_u64((uint64_t)(*s - '0')));
    return acc;
}
// canonical-integer comparison: a > (p-1)/2  <=>  a > -a   (ark-serialize "is largest" flag)
inline bool fp_is_lexicographically_largest(const Fp& a) {
    uint64_t ra[6], rn[6];
    fp_to_raw(ra, a);
    fp_to_raw(rn, fp_neg(a));
    if (memcmp(ra, rn, 48) == 0) return false;
    return limbs_geq(ra, rn);
}
inline int fp_cmp_canonical(const Fp& a, const Fp& b) {
    uint64_t ra[6], rb[6];
    fp_to_raw(ra, a);
    fp_to_raw(rb, b);
    for (int i = 5; i >= 0; i--) {
        if (ra[i] > rb[i]) return 1;
        if (ra[i] < rb[i]) return -1;
    }
    return 0;
}
// sqrt for p = 3 mod 4: a^((p+1)/4); returns false if a is a non-residue
inline bool fp_sqrt(Fp& out, const Fp& a) {
    uint64_t e[6];
    memcpy(e, P_LIMBS, 48);
    // (p+1)/4
    uint64_t carry = 1;
    for (int i = 0; i < 6 && carry; i++) {
        e[i] += carry;
        carry = (e[i] == 0);
    }
    for (int k = 0; k < 2; k++) {
        for (int i = 0; i < 5; i++) e[i] = (e[i] >> 1) | (e[i + 1] << 63);
        e[5] >>= 1;
    }
    Fp r = fp_pow(a, e, 6);
    if (!fp_eq(fp_sqr(r), a)) return false;
    out = r;
    return true;
}

// ---------------------------------------------------------------- Fp2
struct Fp2 {
    Fp c0, c1;
};
inline Fp2 fp2_zero() { return {fp_zero(), fp_zero()}; }
inline Fp2 fp2_one() { return {fp_one(), fp_zero()}; }
inline bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
inline bool fp2_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
inline Fp2 fp2_add(const Fp2& a, const Fp2& b) { return {fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
inline Fp2 fp2_sub(const Fp2& a, const Fp2& b) { return {fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
inline Fp2 fp2_neg(const Fp2& a) { return {fp_neg(a.c0), fp_neg(a.c1)}; }
inline Fp2 fp2_dbl(const Fp2& a) { return fp2_add(a, a); }
inline Fp2 fp2_conj(const Fp2& a) { return {a.c0, fp_neg(a.c1)}; }
inline Fp2 fp2_mul(const Fp2& a, const Fp2& b) {
    Fp v0 = fp_mul(a.c0, b.c0), v1 = fp_mul(a.c1, b.c1);
    Fp s = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    return {fp_sub(v0, v1), fp_sub(fp_sub(s, v0), v1)};
}
inline Fp2 fp2_sqr(const Fp2& a) {
    Fp v = fp_mul(a.c0, a.c1);
    Fp t = fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1));
    return {t, fp_dbl(v)};
}
inline Fp2 fp2_mul_fp(const Fp2& a, const Fp& b) { return {fp_mul(a.c0, b), fp_mul(a.c1, b)}; }
// multiply by the Fp6 non-residue xi = 1 + u
inline Fp2 fp2_mul_xi(const Fp2& a) { return {fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)}; }
inline Fp2 fp2_inv(const Fp2& a) {
    Fp n = fp_add(fp_sqr(a.c0), fp_sqr(a.c1));
    Fp ni = fp_inv(n);
    return {fp_mul(a.c0, ni), fp_neg(fp_mul(a.c1, ni))};
}
inline Fp2 fp2_pow(const Fp2& a, const uint64_t* e, int nlimbs) {
    Fp2 r = fp2_one();
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        r = fp2_sqr(r);
        if ((e[i / 64] >> (i % 64)) & 1) r = fp2_mul(r, a);
    }
    return r;
}
// ark-ff QuadExtField ordering: c1 is the most significant component
inline bool fp2_is_lexicographically_largest(const Fp2& a) {
    Fp2 n = fp2_neg(a);
    int c = fp_cmp_canonical(a.c1, n.c1);
    if (c != 0) return c > 0;
    return fp_cmp_canonical(a.c0, n.c0) > 0;
}
inline bool fp2_sqrt(Fp2& out, const Fp2& a) {
    if (fp2_is_zero(a)) {
        out = a;
        return true;
    }
    if (fp_is_zero(a.c1)) {
        Fp r;
        if (fp_sqrt(r, a.c0)) {
            out = {r, fp_zero()};
            return true;
        }
        if (fp_sqrt(r, fp_neg(a.c0))) {
            out = {fp_zero(), r};
            return true;
        }
        return false;
    }
    Fp alpha;
    if (!fp_sqrt(alpha, fp_add(fp_sqr(a.c0), fp_sqr(a.c1)))) return false;
    Fp two_inv = fp_inv(fp_from_u64(2));
    Fp delta = fp_mul(fp_add(a.c0, alpha), two_inv);
    Fp x0;
    if (!fp_sqrt(x0, delta)) {
        delta = fp_mul(fp_sub(a.c0, alpha), two_inv);
        if (!fp_sqrt(x0, delta)) return false;
    }
    Fp x1 = fp_mul(a.c1, fp_inv(fp_dbl(x0)));
    Fp2 r = {x0, x1};
    if (!fp2_eq(fp2_sqr(r), a)) return false;
    out = r;
    return true;
}

// ---------------------------------------------------------------- Fp6 / Fp12 (native; used for the Fp12 inverse hint and Frobenius)
struct Fp6 {
    Fp2 c0, c1, c2;
};
struct Fp12 {
    Fp6 c0, c1;
};
inline Fp6 fp6_zero() { return {fp2_zero(), fp2_zero(), fp2_zero()}; }
inline Fp6 fp6_one() { return {fp2_one(), fp2_zero(), fp2_zero()}; }
inline Fp6 fp6_add(const Fp6& a, const Fp6& b) { return {fp2_add(a.c0, b.c0), fp2_add(a.c1, b.c1), fp2_add(a.c2, b.c2)}; }
inline Fp6 fp6_sub(const Fp6& a, const Fp6& b) { return {fp2_sub(a.c0, b.c0), fp2_sub(a.c1, b.c1), fp2_sub(a.c2, b.c2)}; }
inline Fp6 fp6_neg(const Fp6& a) { return {fp2_neg(a.c0), fp2_neg(a.c1), fp2_neg(a.c2)}; }
inline Fp6 fp6_mul(const Fp6& a, const Fp6& b) {
    Fp2 v0 = fp2_mul(a.c0, b.c0), v1 = fp2_mul(a.c1, b.c1), v2 = fp2_mul(a.c2, b.c2);
    Fp2 t0 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c1, a.c2), fp2_add(b.c1, b.c2)), v1), v2);
    Fp2 t1 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1)), v0), v1);
    Fp2 t2 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2)), v0), v2);
    return {fp2_add(fp2_mul_xi(t0), v0), fp2_add(t1, fp2_mul_xi(v2)), fp2_add(t2, v1)};
}
// multiply by v: (c0,c1,c2) -> (xi*c2, c0, c1)
inline Fp6 fp6_mul_v(const Fp6& a) { return {fp2_mul_xi(a.c2), a.c0, a.c1}; }
inline Fp6 fp6_inv(const Fp6& a) {
    Fp2 t0 = fp2_sub(fp2_sqr(a.c0), fp2_mul_xi(fp2_mul(a.c1, a.c2)));
    Fp2 t1 = fp2_sub(fp2_mul_xi(fp2_sqr(a.c2)), fp2_mul(a.c0, a.c1));
    Fp2 t2 = fp2_sub(fp2_sqr(a.c1), fp2_mul(a.c0, a.c2));
    Fp2 n = fp2_add(fp2_mul(a.c0, t0), fp2_mul_xi(fp2_add(fp2_mul(a.c2, t1), fp2_mul(a.c1, t2))));
    Fp2 ni = fp2_inv(n);
    return {fp2_mul(t0, ni), fp2_mul(t1, ni), fp2_mul(t2, ni)};
}
inline Fp12 fp12_one() { return {fp6_one(), fp6_zero()}; }
inline Fp12 fp12_mul(const Fp12& a, const Fp12& b) {
    Fp6 v0 = fp6_mul(a.c0, b.c0), v1 = fp6_mul(a.c1, b.c1);
    Fp6 s = fp6_mul(fp6_add(a.c0, a.c1), fp6_add(b.c0, b.c1));
    return {fp6_add(v0, fp6_mul_v(v1)), fp6_sub(fp6_sub(s, v0), v1)};
}
inline Fp12 fp12_inv(const Fp12& a) {
    Fp6 n = fp6_sub(fp6_mul(a.c0, a.c0), fp6_mul_v(fp6_mul(a.c1, a.c1)));
    Fp6 ni = fp6_inv(n);
    return {fp6_mul(a.c0, ni), fp6_neg(fp6_mul(a.c1, ni))};
}
inline bool fp6_eq(const Fp6& a, const Fp6& b) { return fp2_eq(a.c0, b.c0) && fp2_eq(a.c1, b.c1) && fp2_eq(a.c2, b.c2); }
inline bool fp12_eq(const Fp12& a, const Fp12& b) { return fp6_eq(a.c0, b.c0) && fp6_eq(a.c1, b.c1); }

// Frobenius coefficients, derived (not tabulated): gamma = xi^((p-1)/6).
// FROB12_C1[k] = xi^((p^k-1)/6), FROB6_C1[k] = xi^((p^k-1)/3), FROB6_C2[k] = xi^((2p^k-2)/3)
struct FrobTables {
    Fp2 f12c1[4], f6c1[4], f6c2[4];
    FrobTables() {
        // e = (p-1)/6
        uint64_t e[6];
        memcpy(e, P_LIMBS, 48);
        e[0] -= 1;
        // divide by 6
        u128 rem = 0;
        for (int i = 5; i >= 0; i--) {
            u128 cur = (rem << 64) | e[i];
            e[i] = (uint64_t)(cur / 6);
            rem = cur % 6;
        }
        assert(rem == 0);
        Fp2 xi = {fp_one(), fp_one()};
        Fp2 g1 = fp2_pow(xi, e, 6);
        Fp2 g[4];
        g[0] = fp2_one();
        g[1] = g1;
        g[2] = fp2_mul(g1, fp2_conj(g1));           // gamma^(p+1)
        g[3] = fp2_mul(g[2], g1);                   // gamma^(p^2+p+1) with gamma^(p^2)=gamma
        for (int k = 0; k < 4; k++) {
            f12c1[k] = g[k];
            f6c1[k] = fp2_sqr(g[k]);
            f6c2[k] = fp2_sqr(f6c1[k]);
        }
    }
};
inline const FrobTables& frob_tables() {
    static FrobTables t;
    return t;
}

}  // namespace orc
