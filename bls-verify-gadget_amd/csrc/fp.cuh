// BLS12-381 base field for gfx950: 12 x 32-bit limbs, Montgomery form (R = 2^384), one element per lane.
// Values are always fully reduced (< p) so that every emitted witness is the canonical arkworks
// in-memory representation (6 little-endian u64 limbs == 12 little-endian u32 limbs).
// Replaces, on the device, ark-ff ^0.4.0's Fp<MontBackend<FqConfig,6>,6> used by the reference
// (src/hasher.rs:32, src/constraints.rs:18).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BLSW_HD __host__ __device__ __forceinline__
#define BLSW_HD_NOINLINE __host__ __device__ __noinline__
#define BLSW_FN __host__ __device__ __noinline__
#else
#define BLSW_HD inline
#define BLSW_HD_NOINLINE
#define BLSW_FN inline
#endif

namespace blsw {

struct Fp {
    uint32_t l[12];
};
struct Fp2 {
    Fp c0, c1;
};

#define BLSW_P_LIMBS                                                                                                                  \
    {                                                                                                                                 \
        0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, \
            0x397fe69au, 0x1a0111eau                                                                                                  \
    }
#define BLSW_R1_LIMBS                                                                                                                 \
    {                                                                                                                                 \
        0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, \
            0xfa80e493u, 0x15f65ec3u                                                                                                  \
    }
#define BLSW_R2_LIMBS                                                                                                                 \
    {                                                                                                                                 \
        0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, \
            0x92cae3aau, 0x11988fe5u                                                                                                  \
    }
#define BLSW_R3_LIMBS                                                                                                                 \
    {                                                                                                                                 \
        0xd94ca1e0u, 0xed48ac6bu, 0x03a7adf8u, 0x315f831eu, 0x615e29ddu, 0x9a53352au, 0x921e1761u, 0x34c04e5eu, 0x65724728u, 0x2512d435u, \
            0x91755d4du, 0x0aa63460u                                                                                                  \
    }
#define BLSW_INV32 0xfffcfffdu

BLSW_HD uint32_t p_limb(int i) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    return P[i];
}
BLSW_HD Fp fp_zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = 0;
    return r;
}
BLSW_HD Fp fp_one() {
    constexpr uint32_t R1[12] = BLSW_R1_LIMBS;
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = R1[i];
    return r;
}
BLSW_HD bool fp_is_zero(const Fp& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) o |= a.l[i];
    return o == 0;
}
BLSW_HD bool fp_eq(const Fp& a, const Fp& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}
// r = a - p if a >= p else a   (a < 2p, given with its 13th carry word)
BLSW_HD Fp fp_cond_sub_p(const Fp& a, uint32_t top) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    Fp d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t x = (uint64_t)a.l[i] - P[i] - borrow;
        d.l[i] = (uint32_t)x;
        borrow = (x >> 32) & 1;
    }
    bool ge = (top != 0) || (borrow == 0);
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = ge ? d.l[i] : a.l[i];
    return r;
}
BLSW_HD Fp fp_add(const Fp& a, const Fp& b) {
    Fp s;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t x = (uint64_t)a.l[i] + b.l[i] + c;
        s.l[i] = (uint32_t)x;
        c = x >> 32;
    }
    return fp_cond_sub_p(s, (uint32_t)c);
}
BLSW_HD Fp fp_sub(const Fp& a, const Fp& b) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    Fp d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t x = (uint64_t)a.l[i] - b.l[i] - borrow;
        d.l[i] = (uint32_t)x;
        borrow = (x >> 32) & 1;
    }
    uint32_t mask = borrow ? 0xffffffffu : 0u;
    uint64_t c = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t x = (uint64_t)d.l[i] + (P[i] & mask) + c;
        r.l[i] = (uint32_t)x;
        c = x >> 32;
    }
    return r;
}
BLSW_HD Fp fp_neg(const Fp& a) { return fp_sub(fp_zero(), a); }
BLSW_HD Fp fp_dbl(const Fp& a) { return fp_add(a, a); }

// Montgomery product a*b*R^-1 mod p, CIOS over 32-bit limbs: 144 + 144 + 12 v_mad_u64_u32-class
// multiply-adds (the "300 MAD per Fp-mul" of SURVEY §8d). Kept out of line: one body per code object.
BLSW_HD_NOINLINE Fp fp_mul(const Fp& a, const Fp& b) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    uint32_t t[12];
#pragma unroll
    for (int i = 0; i < 12; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t c = 0;
        uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < 12; j++) {
            uint64_t x = (uint64_t)a.l[j] * bi + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        uint32_t t12 = (uint32_t)c;
        uint32_t m = t[0] * BLSW_INV32;
        uint64_t x = (uint64_t)m * P[0] + t[0];
        c = x >> 32;
#pragma unroll
        for (int j = 1; j < 12; j++) {
            x = (uint64_t)m * P[j] + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        t[11] = t12 + (uint32_t)c;  // t < 2p < 2^383: no carry out
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = t[i];
    return fp_cond_sub_p(r, 0);
}
BLSW_HD Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }

BLSW_HD Fp fp_from_limbs(const uint32_t* p) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = p[i];
    return r;
}
// small integer -> Montgomery form
BLSW_HD Fp fp_from_u32(uint32_t v) {
    constexpr uint32_t R2[12] = BLSW_R2_LIMBS;
    Fp a = fp_zero(), r2;
    a.l[0] = v;
#pragma unroll
    for (int i = 0; i < 12; i++) r2.l[i] = R2[i];
    return fp_mul(a, r2);
}
// Montgomery form -> canonical integer limbs
BLSW_HD Fp fp_to_canonical(const Fp& a) {
    Fp one = fp_zero();
    one.l[0] = 1;
    return fp_mul(a, one);
}

// Inversion, returns 0 for 0 (arkworks `inverse().unwrap_or(zero)` hint semantics).
// Round-1 implementation: Fermat a^(p-2) with a fixed (uniform) exponent: 380 squarings + 226 products.
BLSW_HD_NOINLINE Fp fp_inv(const Fp& a) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    Fp r = a;  // top bit (bit 380) of p-2 is set
#pragma unroll 1
    for (int i = 379; i >= 0; i--) {
        r = fp_sqr(r);
        uint32_t w = P[i >> 5];
        if (i < 32) w -= 2;  // p - 2: only the lowest limb changes (0xffffaaab - 2, no borrow)
        if ((w >> (i & 31)) & 1) r = fp_mul(r, a);
    }
    return r;
}

// ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
BLSW_HD Fp2 fp2_zero() { return {fp_zero(), fp_zero()}; }
BLSW_HD Fp2 fp2_one() { return {fp_one(), fp_zero()}; }
BLSW_HD bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
BLSW_FN Fp2 fp2_add(const Fp2& a, const Fp2& b) { return {fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
BLSW_FN Fp2 fp2_sub(const Fp2& a, const Fp2& b) { return {fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
BLSW_FN Fp2 fp2_neg(const Fp2& a) { return {fp_neg(a.c0), fp_neg(a.c1)}; }
BLSW_FN Fp2 fp2_dbl(const Fp2& a) { return {fp_dbl(a.c0), fp_dbl(a.c1)}; }
BLSW_HD Fp2 fp2_conj(const Fp2& a) { return {a.c0, fp_neg(a.c1)}; }
BLSW_FN Fp2 fp2_mul_xi(const Fp2& a) { return {fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)}; }  // * (1+u)
// value-only products (used where the circuit has a constant operand: linear combination, no witness)
BLSW_FN Fp2 fp2_mul(const Fp2& a, const Fp2& b) {
    Fp v0 = fp_mul(a.c0, b.c0), v1 = fp_mul(a.c1, b.c1);
    Fp s = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    return {fp_sub(v0, v1), fp_sub(fp_sub(s, v0), v1)};
}
BLSW_FN Fp2 fp2_sqr(const Fp2& a) {
    Fp v = fp_mul(a.c0, a.c1);
    Fp t = fp_mul(fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1));
    return {t, fp_dbl(v)};
}
BLSW_FN Fp2 fp2_mul_fp(const Fp2& a, const Fp& b) { return {fp_mul(a.c0, b), fp_mul(a.c1, b)}; }
BLSW_FN Fp2 fp2_inv(const Fp2& a) {
    Fp n = fp_add(fp_sqr(a.c0), fp_sqr(a.c1));
    Fp ni = fp_inv(n);
    return {fp_mul(a.c0, ni), fp_neg(fp_mul(a.c1, ni))};
}

}  // namespace blsw
