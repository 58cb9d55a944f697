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
#define BLSW_HD_NOINLINE inline __host__ __device__ __noinline__  // `inline`: one definition per program across the translation units
// BLSW_FN: the witness programs above the Fp product. Out of line by default (one body per code object); a translation unit that
// defines BLSW_INLINE_CHAINS inlines them into its kernels, so that the kernel's register budget (amdgpu_waves_per_eu) governs
// all of the chain's code — a separate function is compiled against the full 512-register file whatever its caller asks for.
#if defined(BLSW_INLINE_CHAINS) && defined(__HIP_DEVICE_COMPILE__)
#define BLSW_FN __host__ __device__ __forceinline__
#else
#define BLSW_FN inline __host__ __device__ __noinline__
#endif
#else
#define BLSW_HD inline
#define BLSW_HD_NOINLINE inline  // host-only translation units (test harness, csrc/r1cs.cpp): no second strong definition
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
// add / subtract with carry. clang lowers __builtin_addc/__builtin_subc to v_add_co_u32 / v_addc_co_u32 chains.
#if defined(__clang__)
BLSW_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t& carry) {
    unsigned co;
    uint32_t r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
}
BLSW_HD uint32_t subb32(uint32_t a, uint32_t b, uint32_t& borrow) {
    unsigned bo;
    uint32_t r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
}
#else
BLSW_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t& carry) {
    uint64_t x = (uint64_t)a + b + carry;
    carry = (uint32_t)(x >> 32);
    return (uint32_t)x;
}
BLSW_HD uint32_t subb32(uint32_t a, uint32_t b, uint32_t& borrow) {
    uint64_t x = (uint64_t)a - b - borrow;
    borrow = (uint32_t)(x >> 32) & 1u;
    return (uint32_t)x;
}
#endif
// r = a - p if a >= p else a   (a < 2p, given with its 13th carry word)
BLSW_HD Fp fp_cond_sub_p(const Fp& a, uint32_t top) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    Fp d;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) d.l[i] = subb32(a.l[i], P[i], borrow);
    bool ge = (top != 0) || (borrow == 0);
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = ge ? d.l[i] : a.l[i];
    return r;
}
BLSW_HD Fp fp_add(const Fp& a, const Fp& b) {
    Fp s;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) s.l[i] = addc32(a.l[i], b.l[i], c);
    return fp_cond_sub_p(s, c);
}
BLSW_HD Fp fp_sub(const Fp& a, const Fp& b) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    Fp d;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) d.l[i] = subb32(a.l[i], b.l[i], borrow);
    uint32_t mask = 0u - borrow;
    uint32_t c = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = addc32(d.l[i], P[i] & mask, c);
    return r;
}
BLSW_HD Fp fp_neg(const Fp& a) { return fp_sub(fp_zero(), a); }
BLSW_HD Fp fp_dbl(const Fp& a) { return fp_add(a, a); }

// Montgomery product a*b*R^-1 mod p, CIOS over 32-bit limbs: 144 + 144 v_mad_u64_u32 + 12 v_mul_lo_u32
// (the "300 MAD per Fp-mul" of SURVEY §8d). Per row the 12 products are independent (no carry in the mad
// chain); the high halves are folded in by one add-with-carry pass. Kept out of line: one body per code object.
// Operands travel as six 16-byte vectors so that BOTH stay in VGPRs across the call (a by-value struct pair
// would send the second operand through the stack).
#if defined(__HIPCC__)
typedef uint4 blsw_u4;
#else
struct blsw_u4 {
    uint32_t x, y, z, w;
};
#endif
BLSW_HD_NOINLINE Fp fp_mul_v(blsw_u4 a0, blsw_u4 a1, blsw_u4 a2, blsw_u4 b0, blsw_u4 b1, blsw_u4 b2) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    const uint32_t al[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
    const uint32_t bl[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
    uint32_t t[13];
#pragma unroll
    for (int i = 0; i < 13; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint32_t bi = bl[i];
        uint64_t x[12];
#pragma unroll
        for (int j = 0; j < 12; j++) x[j] = (uint64_t)al[j] * bi + t[j];
        uint32_t c = 0;
        t[0] = (uint32_t)x[0];
#pragma unroll
        for (int j = 1; j < 12; j++) t[j] = addc32((uint32_t)x[j], (uint32_t)(x[j - 1] >> 32), c);
        t[12] = addc32(t[12], (uint32_t)(x[11] >> 32), c);
        const uint32_t m = t[0] * BLSW_INV32;
#pragma unroll
        for (int j = 0; j < 12; j++) x[j] = (uint64_t)m * P[j] + t[j];
        c = 0;
#pragma unroll
        for (int j = 1; j < 12; j++) t[j - 1] = addc32((uint32_t)x[j], (uint32_t)(x[j - 1] >> 32), c);
        t[11] = addc32(t[12], (uint32_t)(x[11] >> 32), c);
        t[12] = c;  // 0: t < 2p < 2^383
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = t[i];
    return fp_cond_sub_p(r, 0);
}
// The same product on 14 limbs of 28 bits with 64-bit column accumulators: no carry chain inside the 14 x 14 + 14 x 14
// multiply-adds (a column collects at most 28 products of 56 bits), one carry per row (the cleared low limb), one
// normalisation pass at the end. 392 v_mad_u64_u32 instead of 288, but ~320 other instructions instead of ~716 (the 32-bit
// CIOS spends two thirds of its issue slots on carries and zero-extension moves). 14 x 28 = 392 bits, so the reduction
// divides by 2^392: the second operand enters shifted left by 8 bits (b < p < 2^381, so 256 b still fits), which leaves
// a * b * 2^-384 exactly as in fp_mul_v. Result < 1.125 p before the final conditional subtraction.
#define BLSW_P28                                                                                                                               \
    {                                                                                                                                          \
        0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u, 0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, \
            0xe69a4b1u, 0x1ea397fu, 0x001a011u                                                                                                 \
    }
#define BLSW_PINV28 0xffcfffdu
BLSW_HD_NOINLINE Fp fp_mul_v28(blsw_u4 a0, blsw_u4 a1, blsw_u4 a2, blsw_u4 b0, blsw_u4 b1, blsw_u4 b2) {
    constexpr uint32_t P28[14] = BLSW_P28;
    const uint32_t M28 = 0x0fffffffu;
    const uint32_t al[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
    const uint32_t bl[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
    uint32_t A[14], B[14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        // A[i] = bits [28 i, 28 i + 28) of a
        const int bp = 28 * i, w = bp >> 5, sh = bp & 31;
        uint32_t v = al[w] >> sh;
        if (sh > 4 && w + 1 < 12) v |= al[w + 1] << (32 - sh);
        A[i] = v & M28;
        // B[i] = bits [28 i - 8, 28 i + 20) of b
        const int bq = 28 * i - 8;
        uint32_t u;
        if (bq < 0)
            u = bl[0] << 8;
        else {
            const int wq = bq >> 5, sq = bq & 31;
            u = bl[wq] >> sq;
            if (sq > 4 && wq + 1 < 12) u |= bl[wq + 1] << (32 - sq);
        }
        B[i] = u & M28;
    }
    uint64_t t[15];
#pragma unroll
    for (int j = 0; j < 15; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t bi = B[i];
#pragma unroll
        for (int j = 0; j < 14; j++) t[j] += (uint64_t)A[j] * bi;
        const uint32_t m = ((uint32_t)t[0] * BLSW_PINV28) & M28;
#pragma unroll
        for (int j = 0; j < 14; j++) t[j] += (uint64_t)m * P28[j];
        t[1] += t[0] >> 28;  // t[0] is a multiple of 2^28 now
#pragma unroll
        for (int j = 0; j < 14; j++) t[j] = t[j + 1];
        t[14] = 0;
    }
    uint32_t r28[15];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 14; j++) {
        const uint64_t v = t[j] + c;
        r28[j] = (uint32_t)v & M28;
        c = v >> 28;
    }
    r28[14] = 0;
    Fp r;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int bp = 32 * k, i = bp / 28, off = bp - 28 * i;  // off <= 24: two limbs cover the word
        r.l[k] = (r28[i] >> off) | (r28[i + 1] << (28 - off));
    }
    return fp_cond_sub_p(r, 0);
}
#ifndef BLSW_FP_MUL28
#define BLSW_FP_MUL28 1
#endif
BLSW_HD Fp fp_mul(const Fp& a, const Fp& b) {
    blsw_u4 a0 = {a.l[0], a.l[1], a.l[2], a.l[3]}, a1 = {a.l[4], a.l[5], a.l[6], a.l[7]}, a2 = {a.l[8], a.l[9], a.l[10], a.l[11]};
    blsw_u4 b0 = {b.l[0], b.l[1], b.l[2], b.l[3]}, b1 = {b.l[4], b.l[5], b.l[6], b.l[7]}, b2 = {b.l[8], b.l[9], b.l[10], b.l[11]};
#if BLSW_FP_MUL28
    return fp_mul_v28(a0, a1, a2, b0, b1, b2);
#else
    return fp_mul_v(a0, a1, a2, b0, b1, b2);
#endif
}
BLSW_HD Fp fp_mul32(const Fp& a, const Fp& b) {  // the 12 x 32-bit CIOS, kept as the cross-check of fp_mul_v28
    blsw_u4 a0 = {a.l[0], a.l[1], a.l[2], a.l[3]}, a1 = {a.l[4], a.l[5], a.l[6], a.l[7]}, a2 = {a.l[8], a.l[9], a.l[10], a.l[11]};
    blsw_u4 b0 = {b.l[0], b.l[1], b.l[2], b.l[3]}, b1 = {b.l[4], b.l[5], b.l[6], b.l[7]}, b2 = {b.l[8], b.l[9], b.l[10], b.l[11]};
    return fp_mul_v(a0, a1, a2, b0, b1, b2);
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
// Fermat a^(p-2) with a fixed (uniform) exponent: 380 squarings + 226 products. Kept as the independent
// cross-check of fp_inv (tests) — not used on the hot path.
BLSW_HD_NOINLINE Fp fp_inv_fermat(const Fp& a) {
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

// ---- Bernstein-Yang "safegcd" inversion (half-delta divsteps), 13 signed 30-bit limbs, batches of 30 divsteps.
// Per batch: 30 cheap steps on the low words build a 2x2 transition matrix, which is then applied to the full
// (f, g) and, modulo p, to (d, e): 10 x 13 v_mad_i64_i32 instead of 30 full-width shift/add passes.
// e starts at R^2 mod p, so for a Montgomery input aR the output is a^-1 R directly. All lanes run the same
// instruction stream; the loop ends when every lane's g is 0 (<= 30 batches for 381-bit inputs).
#define BLSW_P30                                                                                                                              \
    {                                                                                                                                         \
        0x3fffaaab, 0x27fbffff, 0x153ffffb, 0x2affffac, 0x30f6241e, 0x034a83da, 0x112bf673, 0x12e13ce1, 0x2cd76477, 0x1ed90d2e, 0x29a4b1ba, \
            0x3a8e5ff9, 0x001a0111                                                                                                            \
    }
#define BLSW_R2_30                                                                                                                            \
    {                                                                                                                                         \
        0x1c341746, 0x137c7cd0, 0x1d104f1f, 0x1db9a982, 0x15b6d50a, 0x151db132, 0x183c08de, 0x222a64e7, 0x152d67eb, 0x3a16d466, 0x3aa9a793, \
            0x3964b2b8, 0x0011988f                                                                                                            \
    }
#define BLSW_PINV30 0x30003u
#if defined(__HIP_DEVICE_COMPILE__)
#define BLSW_WAVE_ANY(x) (__any((x)) != 0)
#else
#define BLSW_WAVE_ANY(x) (x)
#endif
// signed 32 x 32 + 64 multiply-add: one v_mad_i64_i32 (the compiler otherwise lowers products of masked, provably
// non-negative limbs to unsigned mads plus sign fix-ups)
BLSW_HD int64_t mad_i64(int32_t a, int32_t b, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
#else
    return (int64_t)a * b + c;
#endif
}
BLSW_HD int64_t mad_i64_k(int32_t konst, int32_t b, int64_t c) {  // `konst` is a compile-time constant: kept in an SGPR
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(d) : "s"(konst), "v"(b), "v"(c) : "vcc");
    return d;
#else
    return (int64_t)konst * b + c;
#endif
}
BLSW_HD_NOINLINE Fp fp_inv(const Fp& a) {
    constexpr int32_t P30[13] = BLSW_P30;
    constexpr int32_t R2_30[13] = BLSW_R2_30;
    const int32_t M30 = 0x3fffffff;
    int32_t f[13], g[13], d[13], e[13];
    // g = a as 13 x 30-bit limbs
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const int bp = 30 * k, w = bp >> 5, sh = bp & 31;
        uint32_t v = (w < 12) ? (a.l[w] >> sh) : 0u;
        if (sh > 2 && w + 1 < 12) v |= a.l[w + 1] << (32 - sh);
        g[k] = (int32_t)(v & (uint32_t)M30);
        f[k] = P30[k];
        d[k] = 0;
        e[k] = R2_30[k];
    }
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 37; it++) {
        uint32_t gnz = 0;
#pragma unroll
        for (int k = 0; k < 13; k++) gnz |= (uint32_t)g[k];
        if (!BLSW_WAVE_ANY(gnz != 0)) break;
        // ---- 30 divsteps on the low words
        uint32_t u = 1, v = 0, q = 0, r = 1;
        uint32_t f0 = (uint32_t)f[0] | ((uint32_t)f[1] << 30), g0 = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
#pragma unroll 6
        for (int i = 0; i < 30; i++) {
            uint32_t c1 = (uint32_t)(zeta >> 31);
            uint32_t c2 = 0u - (g0 & 1u);
            uint32_t x = (f0 ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
            g0 += x & c2;
            q += y & c2;
            r += z & c2;
            c1 &= c2;
            zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
            f0 += g0 & c1;
            u += q & c1;
            v += r & c1;
            g0 >>= 1;
            u <<= 1;
            v <<= 1;
        }
        const int32_t tu = (int32_t)u, tv = (int32_t)v, tq = (int32_t)q, tr = (int32_t)r;
        // ---- (d, e) <- t * (d, e) / 2^30 mod p
        {
            int32_t sd = d[12] >> 31, se = e[12] >> 31;
            int32_t md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
            int64_t cd = mad_i64(tv, e[0], mad_i64(tu, d[0], 0));
            int64_t ce = mad_i64(tr, e[0], mad_i64(tq, d[0], 0));
            md -= (int32_t)((BLSW_PINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
            me -= (int32_t)((BLSW_PINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
            cd = mad_i64_k(P30[0], md, cd);
            ce = mad_i64_k(P30[0], me, ce);
            cd >>= 30;
            ce >>= 30;
#pragma unroll
            for (int k = 1; k < 13; k++) {
                int32_t dk = d[k], ek = e[k];
                cd = mad_i64_k(P30[k], md, mad_i64(tv, ek, mad_i64(tu, dk, cd)));
                ce = mad_i64_k(P30[k], me, mad_i64(tr, ek, mad_i64(tq, dk, ce)));
                d[k - 1] = (int32_t)cd & M30;
                e[k - 1] = (int32_t)ce & M30;
                cd >>= 30;
                ce >>= 30;
            }
            d[12] = (int32_t)cd;
            e[12] = (int32_t)ce;
        }
        // ---- (f, g) <- t * (f, g) / 2^30
        {
            int64_t cf = mad_i64(tv, g[0], mad_i64(tu, f[0], 0));
            int64_t cg = mad_i64(tr, g[0], mad_i64(tq, f[0], 0));
            cf >>= 30;
            cg >>= 30;
#pragma unroll
            for (int k = 1; k < 13; k++) {
                int32_t fk = f[k], gk = g[k];
                cf = mad_i64(tv, gk, mad_i64(tu, fk, cf));
                cg = mad_i64(tr, gk, mad_i64(tq, fk, cg));
                f[k - 1] = (int32_t)cf & M30;
                g[k - 1] = (int32_t)cg & M30;
                cf >>= 30;
                cg >>= 30;
            }
            f[12] = (int32_t)cf;
            g[12] = (int32_t)cg;
        }
    }
    // ---- normalize d: into (-p, p), negate if f < 0, then into [0, p)
    {
        int32_t cond_add = d[12] >> 31;
#pragma unroll
        for (int k = 0; k < 13; k++) d[k] += P30[k] & cond_add;
        int32_t cond_neg = f[12] >> 31;
#pragma unroll
        for (int k = 0; k < 13; k++) d[k] = (d[k] ^ cond_neg) - cond_neg;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            d[k + 1] += d[k] >> 30;
            d[k] &= M30;
        }
        cond_add = d[12] >> 31;
#pragma unroll
        for (int k = 0; k < 13; k++) d[k] += P30[k] & cond_add;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            d[k + 1] += d[k] >> 30;
            d[k] &= M30;
        }
    }
    // back to 12 x 32-bit limbs
    Fp r;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bp = 32 * w, k = bp / 30, sh = bp - 30 * k;  // bit bp lives in 30-bit limb k at offset sh
        uint32_t v = (uint32_t)d[k] >> sh;
        v |= (uint32_t)d[k + 1] << (30 - sh);
        if (30 - sh + 30 < 32 && k + 2 < 13) v |= (uint32_t)d[k + 2] << (60 - sh);
        r.l[w] = v;
    }
    return r;
}

// ---------------------------------------------------------------- quads (latency compilation, -DBLSW_QUAD)
// A translation unit compiled with BLSW_QUAD runs every chain on FOUR adjacent lanes: all four hold the same values and execute the same
// program, and the independent Fp products of an Fp2 operation (three of a Karatsuba product, two of a square, two of a product by an Fp
// element, the two squares of a norm) go to different lanes; the products come back to all four lanes through DPP quad broadcasts (twelve
// v_mov_b32_dpp per element). The programs above (chains.hpp, cofactor_vf.hpp, ...) do not change: only the Fp2 primitives below and in
// gadgets.hpp do. Used for small launch groups that start a pipeline, whose latency is one wave's instruction stream (engine.hip).
#if defined(BLSW_QUAD) && defined(__HIP_DEVICE_COMPILE__)
#define BLSW_QUAD_DEV 1
BLSW_HD uint32_t quad_role() { return threadIdx.x & 3u; }
template <int K>
BLSW_HD Fp quad_bcast(const Fp& v) {  // lane K of the quad -> all four lanes
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], K * 0x55, 0xf, 0xf, true);
    return r;
}
template <int K>
BLSW_HD uint32_t quad_bcast_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, K * 0x55, 0xf, 0xf, true);
}
// selects on VALUES (a conditional on two lvalues is an lvalue: the compiler would select addresses, and the operands could not leave memory)
BLSW_HD Fp quad_sel2(uint32_t role, const Fp& a, const Fp& b) {  // even lanes a, odd lanes b
    const bool odd = (role & 1u) != 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint32_t x = a.l[i], y = b.l[i];
        r.l[i] = odd ? y : x;
    }
    return r;
}
BLSW_HD Fp quad_sel3(uint32_t role, const Fp& a, const Fp& b, const Fp& c) {  // lane 0 (and 3) a, lane 1 b, lane 2 c
    const bool is1 = role == 1u, is2 = role == 2u;
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint32_t x = a.l[i], y = b.l[i], z = c.l[i];
        const uint32_t t = is2 ? z : x;
        r.l[i] = is1 ? y : t;
    }
    return r;
}
#endif

// ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
BLSW_HD Fp2 fp2_zero() { return {fp_zero(), fp_zero()}; }
BLSW_HD Fp2 fp2_one() { return {fp_one(), fp_zero()}; }
BLSW_HD bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
BLSW_HD Fp2 fp2_add(const Fp2& a, const Fp2& b) { return {fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
BLSW_HD Fp2 fp2_sub(const Fp2& a, const Fp2& b) { return {fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
BLSW_HD Fp2 fp2_neg(const Fp2& a) { return {fp_neg(a.c0), fp_neg(a.c1)}; }
BLSW_HD Fp2 fp2_dbl(const Fp2& a) { return {fp_dbl(a.c0), fp_dbl(a.c1)}; }
BLSW_HD Fp2 fp2_conj(const Fp2& a) { return {a.c0, fp_neg(a.c1)}; }
BLSW_HD Fp2 fp2_mul_xi(const Fp2& a) { return {fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)}; }  // * (1+u)
// value-only products (used where the circuit has a constant operand: linear combination, no witness)
#ifdef BLSW_QUAD_DEV
// one Fp product per lane: lane 0 a0 b0, lane 1 a1 b1, lane 2 (a0 + a1)(b0 + b1); `prod` = this lane's product (the witness of role < 3)
BLSW_HD Fp2 fp2_mul_quad(const Fp2& a, const Fp2& b, Fp& prod) {
    const uint32_t role = quad_role();
    prod = fp_mul(quad_sel3(role, a.c0, a.c1, fp_add(a.c0, a.c1)), quad_sel3(role, b.c0, b.c1, fp_add(b.c0, b.c1)));
    const Fp v0 = quad_bcast<0>(prod), v1 = quad_bcast<1>(prod), s = quad_bcast<2>(prod);
    return {fp_sub(v0, v1), fp_sub(fp_sub(s, v0), v1)};
}
// lane 0 a0 a1, lane 1 (a0 - a1)(a0 + a1)
BLSW_HD Fp2 fp2_sqr_quad(const Fp2& a, Fp& prod) {
    const uint32_t role = quad_role();
    prod = fp_mul(quad_sel2(role, a.c0, fp_sub(a.c0, a.c1)), quad_sel2(role, a.c1, fp_add(a.c0, a.c1)));
    return {quad_bcast<1>(prod), fp_dbl(quad_bcast<0>(prod))};
}
BLSW_HD Fp2 fp2_mul_inl(const Fp2& a, const Fp2& b) {
    Fp prod;
    return fp2_mul_quad(a, b, prod);
}
BLSW_FN Fp2 fp2_mul(const Fp2& a, const Fp2& b) { return fp2_mul_inl(a, b); }
BLSW_FN Fp2 fp2_sqr(const Fp2& a) {
    Fp prod;
    return fp2_sqr_quad(a, prod);
}
BLSW_FN Fp2 fp2_mul_fp(const Fp2& a, const Fp& b) {
    const Fp prod = fp_mul(quad_sel2(quad_role(), a.c0, a.c1), b);
    return {quad_bcast<0>(prod), quad_bcast<1>(prod)};
}
BLSW_HD Fp2 fp2_inv_inl(const Fp2& a) {
    const uint32_t role = quad_role();
    const Fp sq = fp_sqr(quad_sel2(role, a.c0, a.c1));
    const Fp ni = fp_inv(fp_add(quad_bcast<0>(sq), quad_bcast<1>(sq)));
    const Fp prod = fp_mul(quad_sel2(role, a.c0, a.c1), ni);
    return {quad_bcast<0>(prod), fp_neg(quad_bcast<1>(prod))};
}
#else
BLSW_HD Fp2 fp2_mul_inl(const Fp2& a, const Fp2& b) {
    Fp v0 = fp_mul(a.c0, b.c0), v1 = fp_mul(a.c1, b.c1);
    Fp s = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    return {fp_sub(v0, v1), fp_sub(fp_sub(s, v0), v1)};
}
BLSW_FN Fp2 fp2_mul(const Fp2& a, const Fp2& b) { return fp2_mul_inl(a, b); }
BLSW_FN Fp2 fp2_sqr(const Fp2& a) {
    Fp v = fp_mul(a.c0, a.c1);
    Fp t = fp_mul(fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1));
    return {t, fp_dbl(v)};
}
BLSW_FN Fp2 fp2_mul_fp(const Fp2& a, const Fp& b) { return {fp_mul(a.c0, b), fp_mul(a.c1, b)}; }
BLSW_HD Fp2 fp2_inv_inl(const Fp2& a) {
    Fp n = fp_add(fp_sqr(a.c0), fp_sqr(a.c1));
    Fp ni = fp_inv(n);
    return {fp_mul(a.c0, ni), fp_neg(fp_mul(a.c1, ni))};
}
#endif
BLSW_FN Fp2 fp2_inv(const Fp2& a) { return fp2_inv_inl(a); }

}  // namespace blsw
