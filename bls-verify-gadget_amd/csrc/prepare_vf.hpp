// G2PreparedVar::from_group_var (constraints.rs:118, 120) "values first": the latency form of the prepare segments.
//
// The circuit walks the 63 doublings and 5 additions of the Miller loop's point chain in AFFINE coordinates: each step inverts its slope's
// denominator (a witness) and everything else of the step is products of that inverse with the running point. As in cofactor_vf.hpp the
// chain of points is computed as VALUES in Jacobian coordinates (Z' = 2 Y Z for a doubling, Z' = 2 Z H for the mixed addition of the
// prepared point), ONE inversion of the last Z gives every 1 / Z_k by the backward recurrence 1 / Z_k = m_k / Z_{k+1} (m_k = 2 Y_k or 2 H_k),
// and a lane per step recovers the step's affine point (X / Z^2, Y / Z^3) and the inverse it needs — 1 / y = 2 Z^4 / Z' for a doubling,
// 1 / (q.x - x) = 2 Z^3 / Z' for an addition — and runs the step's statements (chains.hpp: prepare_dbl_step / prepare_add_step) at the step's
// place in the segment. Field elements are canonical residues: the same bytes as the affine chain.
// Serial work: 63 x 19 + 5 x 32 products + 2 inversions (to_affine, last Z) instead of 68 x ~48 with 69 inversions.
// Degenerate inputs: the identity (to_affine gives (0, 0), every Z, inverse and slope 0) runs as the circuit does; a point of order two
// inside the chain or r = +-q at an addition (impossible for points of prime order r) would differ from the circuit's zero-hint arithmetic.
#pragma once
#include "cofactor_vf.hpp"

namespace blsw {

#define BLSW_PREPV_STEPS 68
struct PrepvPlan {
    uint8_t is_add[BLSW_PREPV_STEPS];
    uint16_t pos[BLSW_PREPV_STEPS];  // first witness of step k, relative to the prepare segment
    uint16_t total;
};
constexpr PrepvPlan prepv_plan() {
    PrepvPlan p = {};
    uint32_t pos = 18, k = 0;  // g2_to_affine_w
    for (int i = 62; i >= 0; i--) {
        p.is_add[k] = 0;
        p.pos[k++] = (uint16_t)pos;
        pos += 16;
        if ((BLSW_X_ABS >> i) & 1) {
            p.is_add[k] = 1;
            p.pos[k++] = (uint16_t)pos;
            pos += 14;
        }
    }
    p.total = (uint16_t)pos;
    return p;
}
static_assert(prepv_plan().total == 1096, "prepare segment: the plan must count what chain_prepare_g2 emits");
// scratch of one point, in field elements: Q = the prepared point, affine (4); ST(k) = X, Y, Z of the running point before step k, and for an
// addition H and r (10); ZI(k) = 1 / Z before step k, k = 0 .. 68
#define BLSW_PREPV_Q 0u
#define BLSW_PREPV_ST(k) (4u + 10u * (uint32_t)(k))
#define BLSW_PREPV_ZI(k) (4u + 10u * BLSW_PREPV_STEPS + 2u * (uint32_t)(k))
#define BLSW_PREPV_ELEMS (4u + 10u * BLSW_PREPV_STEPS + 2u * (BLSW_PREPV_STEPS + 1))

// ---- phase 1 (one lane, or one quad, per point): to_affine with its witnesses, then the chain of points as values
template <class S>
BLSW_FN void prepv_chain(Emitter e, const Proj<OpsFp2>& q_, const S& scr) {
    constexpr PrepvPlan plan = prepv_plan();
    const Aff2Inf q = g2_to_affine_w(e, q_);
    cofv_st2(scr, BLSW_PREPV_Q, q.x);
    cofv_st2(scr, BLSW_PREPV_Q + 2, q.y);
    Fp2 X = q.x, Y = q.y, Z = fp2_one();
#pragma unroll 1
    for (uint32_t k = 0; k < BLSW_PREPV_STEPS; k++) {
        cofv_st2(scr, BLSW_PREPV_ST(k), X);
        cofv_st2(scr, BLSW_PREPV_ST(k) + 2, Y);
        cofv_st2(scr, BLSW_PREPV_ST(k) + 4, Z);
        if (!plan.is_add[k]) {  // dbl-2009-l, a = 0
            v_dbl_inplace(X, Y, Z);
        } else {  // madd-2007-bl with Z3 = 2 Z1 H
            const Fp2 z1z1 = v_sqr(Z);
            const Fp2 H = fp2_sub(fp2_mul_inl(q.x, z1z1), X);
            const Fp2 rr = fp2_dbl(fp2_sub(fp2_mul_inl(fp2_mul_inl(q.y, Z), z1z1), Y));
            cofv_st2(scr, BLSW_PREPV_ST(k) + 6, H);
            cofv_st2(scr, BLSW_PREPV_ST(k) + 8, rr);
            const Fp2 hh = v_sqr(H);
            const Fp2 I = fp2_dbl(fp2_dbl(hh));
            const Fp2 J = fp2_mul_inl(H, I);
            const Fp2 V = fp2_mul_inl(X, I);
            const Fp2 x3 = fp2_sub(fp2_sub(v_sqr(rr), J), fp2_dbl(V));
            const Fp2 y3 = fp2_sub(fp2_mul_inl(rr, fp2_sub(V, x3)), fp2_dbl(fp2_mul_inl(Y, J)));
            Z = fp2_dbl(fp2_mul_inl(Z, H));
            X = x3;
            Y = y3;
        }
    }
    Fp2 zi = fp2_inv_inl(Z);
    cofv_st2(scr, BLSW_PREPV_ZI(BLSW_PREPV_STEPS), zi);
#pragma unroll 1
    for (int k = BLSW_PREPV_STEPS - 1; k >= 1; k--) {
        const Fp2 m = cofv_ld2(scr, BLSW_PREPV_ST(k) + (plan.is_add[k] ? 6u : 2u));
        zi = fp2_mul_inl(fp2_dbl(m), zi);
        cofv_st2(scr, BLSW_PREPV_ZI(k), zi);
    }
    cofv_st2(scr, BLSW_PREPV_ZI(0), fp2_one());
}

// ---- phase 2 (one lane per step k of a point): the step's 16 / 14 witnesses and its line coefficients
template <class S, class C>
BLSW_FN void prepv_step_w(Emitter e, uint32_t k, const S& scr, const C& out) {
    constexpr PrepvPlan plan = prepv_plan();
    const Fp2 X = cofv_ld2(scr, BLSW_PREPV_ST(k)), Y = cofv_ld2(scr, BLSW_PREPV_ST(k) + 2), Z = cofv_ld2(scr, BLSW_PREPV_ST(k) + 4);
    const Fp2 zi = cofv_ld2(scr, BLSW_PREPV_ZI(k)), zn = cofv_ld2(scr, BLSW_PREPV_ZI(k + 1));
    const Fp2 zi2 = v_sqr(zi);
    Fp2 rx = fp2_mul_inl(X, zi2), ry = fp2_mul_inl(Y, fp2_mul_inl(zi2, zi));
    const Fp2 z2 = v_sqr(Z);
    e.pos += plan.pos[k];
    if (!plan.is_add[k]) {
        const Fp2 ry_inv = fp2_mul_inl(fp2_dbl(v_sqr(z2)), zn);  // 1 / y = Z^3 / Y = 2 Z^4 / Z'
        prepare_dbl_step(e, rx, ry, ry_inv, out, k);
    } else {
        const Fp2 dx_inv = fp2_mul_inl(fp2_dbl(fp2_mul_inl(z2, Z)), zn);  // 1 / (q.x - x) = Z^2 / H = 2 Z^3 / Z'
        prepare_add_step(e, cofv_ld2(scr, BLSW_PREPV_Q), cofv_ld2(scr, BLSW_PREPV_Q + 2), rx, ry, dx_inv, out, k);
    }
}

}  // namespace blsw
