// Scalar multiplications of the native signer (bls.rs:411-425: sig = sk * H(m); bls.rs:183-195: pk = sk * g1), value only.
//   G2: psi acts on the subgroup as multiplication by x = -|x| (|x| = 0xd201000000010000, r = x^4 - x^2 + 1), so
//       sk = k0 + k1 |x| + k2 |x|^2 + k3 |x|^3 (four 64-bit digits) and
//       sk * Q = k0 Q + k1 (-psi Q) + k2 (psi^2 Q) + k3 (-psi^3 Q): one joint ladder of 64 doublings and at most 64 additions from a
//       table of the 15 non-empty sums of the four bases (Straus), instead of 255 doublings and ~128 additions.
//   G1: the base is the fixed generator: pk = sum_w T[w][digit_w] with T[w][d - 1] = d 16^w g1 (g1_table.hpp, generated): 64 mixed
//       additions, no doubling.
// Compiles for the host as well (tests/hostsim: the sign fixtures and the oracle's signer).
#pragma once
#include "g1_table.hpp"
#include "vcurve.hpp"

namespace blsw {

// k (8 little-endian words, k < 2^255) -> its four base-|x| digits, by restoring division
BLSW_HD void v_digits_x(const uint32_t* k, uint64_t d[4]) {
    uint64_t n[4] = {k[0] | ((uint64_t)k[1] << 32), k[2] | ((uint64_t)k[3] << 32), k[4] | ((uint64_t)k[5] << 32), k[6] | ((uint64_t)k[7] << 32)};
#pragma unroll 1
    for (int j = 0; j < 3; j++) {
        uint64_t q[4] = {0, 0, 0, 0}, rem = 0;
#pragma unroll 1
        for (int i = 255; i >= 0; i--) {
            const uint64_t top = rem >> 63;
            rem = (rem << 1) | ((n[i >> 6] >> (i & 63)) & 1);
            if (top || rem >= BLSW_X_ABS) {
                rem -= BLSW_X_ABS;
                q[i >> 6] |= 1ull << (i & 63);
            }
        }
        d[j] = rem;
        for (int w = 0; w < 4; w++) n[w] = q[w];
    }
    d[3] = n[0];  // k < |x|^4: the third quotient is the last digit
}
// [k] q for q in G2 (Jacobian, not the identity). PARK slots 1..15 hold the table (slot = subset of the four bases).
template <class PARK>
BLSW_HD Jac2 v_g2_mul_gls(const PARK& park, const Jac2& q, const uint32_t* k) {
    uint64_t d[4];
    v_digits_x(k, d);
    {
        Jac2 b = q;
        park.st(1, b);
        b = v_neg(v_psi(b));  // |x| q
        park.st(2, b);
        b = v_psi2(q);  // |x|^2 q
        park.st(4, b);
        b = v_neg(v_psi(b));  // |x|^3 q
        park.st(8, b);
    }
    // steps 3..15 without the powers of two: table entries T[s] = T[s without its lowest base] + T[lowest base]; then 64 ladder steps.
    // One loop, one inlined copy of the doubling and of the addition.
    Jac2 acc = {fp2_one(), fp2_one(), fp2_zero()};
#pragma unroll 1
    for (int step = 3; step < 16 + 64; step++) {
        const bool build = step < 16;
        uint32_t idx;
        if (build) {
            if ((step & (step - 1)) == 0) continue;  // a single base: stored above
            const uint32_t low = (uint32_t)step & (0u - (uint32_t)step);
            acc = park.ld((uint32_t)step ^ low);
            idx = low;
        } else {
            const int i = 63 - (step - 16);
            acc = v_dbl(acc);
            idx = (uint32_t)((d[0] >> i) & 1) | (uint32_t)(((d[1] >> i) & 1) << 1) | (uint32_t)(((d[2] >> i) & 1) << 2) | (uint32_t)(((d[3] >> i) & 1) << 3);
        }
        if (idx) acc = v_add(acc, park.ld(idx));
        if (build) {
            park.st((uint32_t)step, acc);
            if (step == 15) acc = {fp2_one(), fp2_one(), fp2_zero()};
        }
    }
    return acc;
}
// [k] g1, k < 2^256 as 8 words
BLSW_HD Jac1v v1_mul_g1_fixed(const uint32_t* k) {
    Jac1v acc = {fp_one(), fp_one(), fp_zero()};
#pragma unroll 1
    for (int w = 0; w < BLSW_G1_TABLE_WINDOWS; w++) {
        const uint32_t dgt = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
        if (dgt) {
            const uint32_t* t = K_G1_TABLE + ((uint32_t)w * 15 + dgt - 1) * 24;
            acc = v1_add_mixed(acc, fp_from_limbs(t), fp_from_limbs(t + 12));
        }
    }
    return acc;
}

}  // namespace blsw
