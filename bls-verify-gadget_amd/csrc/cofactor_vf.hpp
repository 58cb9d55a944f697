// clear_cofactor2 (hasher.rs:664-673) "values first": the latency form of the cofactor segment.
//
// In the circuit the scalar multiplication by h_eff is an AFFINE double-and-add: 636 doublings mopt_{i+1} = 2 mopt_i and
// 304 additions acc += mopt_i, each with a slope lambda = num / den whose quotient is a witness — 940 dependent steps with an inversion
// each (chains.hpp: chain_cofactor, the statement of the segment; cofactor_par.hpp: the same on three lanes). But every witness of a step
// is a function of the step's affine operands and its slope, field elements are canonical residues, and
//   * the doubling chain 2^i P does not depend on the additions: in Jacobian coordinates (dbl-2009-l, Z_{i+1} = 2 Y_i Z_i) it costs 16
//     products per step and NO inversion; 1 / Z_i follows from ONE inversion of the last Z by the backward recurrence
//     1 / Z_i = 2 Y_i / Z_{i+1}, and the slope of doubling i is 3 X_i^2 / Z_{i+1} (3 x^2 / 2 y with x = X / Z^2, y = Y / Z^3);
//   * the additions of a chunk are a chain of mixed Jacobian additions (madd-2007-bl, Z3 = 2 Z1 H) with the affine 2^i P as the second
//     operand: again one inversion per chunk, 1 / Z1_j = 2 H_j / Z1_{j+1}, and the slope of addition j is r_j / Z1_{j+1}.
// So: two short serial value programs (phase 1: the doubling chain of an instance; phase 3: the addition chain of a chunk) and
// embarrassingly parallel programs (phase 2a: the affine 2^D P, 2b: one lane per doubling; phase 4: one lane per addition) that recompute the affine
// operands, multiply out the slope and emit exactly the witnesses of curve.hpp's nz_double_pre_inl / nz_add_unchecked_pre_inl at the
// step's place in the segment (cofv_plan: a compile-time walk over the bits of h_eff). Phase 5 folds the chunks (the statements of
// cofactor_par.hpp's join with the tail doublings already emitted).
// Serial work per instance: 636 x 19 + 304 x 32 products + 4 inversions instead of 940 x ~50 with 636 inversions; with the Fp2 products of
// the serial phases split over the lanes of a quad (BLSW_QUAD, gadgets.hpp) the critical path is ~7 k product-times instead of 21 k.
// Degenerate inputs: Q0 + Q1 = identity runs as the circuit does (all-zero chain: Z = 0, every hint 0). A point of order two inside the
// doubling chain or acc = +-mopt inside an addition chain (probability ~2^-250 for hash outputs) would differ from the circuit's zero-hint
// arithmetic, exactly as cofactor_par.hpp's Jacobian chunk starts already do.
// Pipelined in SEGMENTS of the doubling chain (cofv_seg: six ranges (D0, D1] that end at the chunks' ends): the forward doublings of a segment are the
// only thing the next segment waits for; the segment's inversion + backward recurrence (cofv_bwd), its affine points and the part of its chunk's
// addition chain run BESIDE the following segments (kcommon.hpp: launch_cofactor puts them on other streams), so the critical path is the forward
// doubling chain plus the last segment's short tail instead of chain + recurrence + affine points + the longest addition chain.
// Compiles for the host as well: tests/hostsim runs the phases in order against the oracle.
#pragma once
#include "cofactor_par.hpp"

namespace blsw {

#define BLSW_COFV_MAX_ADDS 144  // additions inside one chunk's loop (checked against the plan)
struct CofvPlan {
    uint16_t pos_dbl[BLSW_H_EFF_NBITS];         // first witness of doubling D (of 2^D P), relative to the cofactor segment
    uint16_t add_bit[3][BLSW_COFV_MAX_ADDS];    // chunk c, addition j: the doubling index D whose operand 2^D P is added
    uint16_t pos_add[3][BLSW_COFV_MAX_ADDS];    // its first witness
    uint16_t n_adds[3];
    uint16_t total;
};
constexpr CofvPlan cofv_plan() {
    CofvPlan p = {};
    uint32_t pos = 18;  // g2_to_affine_w
    int mr_state = -1;
    int c = 0;
    for (int off = 0; off < BLSW_H_EFF_NBITS; off += 255, c++) {
        const int n = BLSW_H_EFF_NBITS - off < 255 ? BLSW_H_EFF_NBITS - off : 255;
        const int split = n < 253 ? n : 253;
        p.pos_dbl[off] = (uint16_t)pos;
        pos += 10;
        uint16_t j = 0;
        for (int i = 1; i < split; i++) {
            if (h_eff_bit(off + i)) {
                p.add_bit[c][j] = (uint16_t)(off + i);
                p.pos_add[c][j] = (uint16_t)pos;
                j++;
                pos += 8;
            }
            p.pos_dbl[off + i] = (uint16_t)pos;
            pos += 10;
        }
        p.n_adds[c] = j;
        int diff_state = 2;
        if (!h_eff_bit(off)) {
            pos += 33;
            diff_state = 0;
        }
        if (mr_state < 0)
            mr_state = diff_state;
        else {
            pos += (mr_state == 2 || diff_state == 2) ? 33 : 36;
            mr_state = 0;
        }
        for (int i = split; i < n; i++) {
            if (h_eff_bit(off + i)) {
                pos += 33;
                mr_state = 0;
            }
            p.pos_dbl[off + i] = (uint16_t)pos;
            pos += 10;
        }
    }
    pos += mr_state == 2 ? 4 : 6;
    p.total = (uint16_t)pos;
    return p;
}
// segments of the doubling chain: segment s covers the points D in (bnd[s], bnd[s + 1]] (segment 0 also D = 0) and belongs to chunk chunk[s]; a chunk's
// loop additions, its first point and its tail operands all lie in its own segments. [j_lo[s], j_hi[s]): the additions of that chunk in segment s.
// The later chunks are cut finer: what follows the last doubling (the last segment's inversion, affine points and additions) is the exposed tail.
#define BLSW_COFV_NSEG 8
struct CofvSeg {
    uint16_t bnd[BLSW_COFV_NSEG + 1];
    uint8_t chunk[BLSW_COFV_NSEG];
    uint16_t j_lo[BLSW_COFV_NSEG], j_hi[BLSW_COFV_NSEG];
    constexpr bool first(int s) const { return s == 0 || chunk[s - 1] != chunk[s]; }
    constexpr bool last(int s) const { return s == BLSW_COFV_NSEG - 1 || chunk[s + 1] != chunk[s]; }
};
constexpr CofvSeg cofv_seg() {
    CofvSeg g = {{0, 128, 254, 340, 425, 509, 552, 594, BLSW_H_EFF_NBITS}, {0, 0, 1, 1, 1, 2, 2, 2}, {}, {}};
    const CofvPlan p = cofv_plan();
    for (int s = 0; s < BLSW_COFV_NSEG; s++) {
        const int c = g.chunk[s];
        uint16_t lo = 0, hi = 0;
        for (uint16_t j = 0; j < p.n_adds[c]; j++) {
            if (p.add_bit[c][j] <= g.bnd[s]) lo = (uint16_t)(j + 1);
            if (p.add_bit[c][j] <= g.bnd[s + 1]) hi = (uint16_t)(j + 1);
        }
        g.j_lo[s] = lo;
        g.j_hi[s] = hi;
    }
    return g;
}
constexpr bool cofv_seg_ok() {
    const CofvSeg g = cofv_seg();
    const CofvPlan p = cofv_plan();
    for (int s = 0; s < BLSW_COFV_NSEG; s++) {
        const int c = g.chunk[s], off = 255 * c;
        if (g.first(s) && (g.j_lo[s] != 0 || g.bnd[s] >= off || g.bnd[s + 1] < off) && s != 0) return false;  // the chunk's first point lies in its first segment
        if (g.last(s) && g.j_hi[s] != p.n_adds[c]) return false;
        if (g.last(s) && c < 2 && g.bnd[s + 1] != off + 254) return false;  // ... and its tail operands in its last
        if (!g.first(s) && g.j_lo[s] != g.j_hi[s - 1]) return false;
    }
    return g.bnd[0] == 0 && g.bnd[BLSW_COFV_NSEG] == BLSW_H_EFF_NBITS;
}
static_assert(cofv_seg_ok(), "cofv: every addition, first point and tail operand of a chunk lies in the chunk's own segments");
static_assert(cofv_plan().total == 8979, "cofactor segment: the plan must count what chain_cofactor emits");
static_assert(cofv_plan().n_adds[0] < BLSW_COFV_MAX_ADDS && cofv_plan().n_adds[1] < BLSW_COFV_MAX_ADDS && cofv_plan().n_adds[2] < BLSW_COFV_MAX_ADDS, "cofv: additions per chunk");

// scratch of one instance, in field elements; S::st(elem, Fp) / S::ld(elem) (element-major rows on the device, an array on the host)
//   XY(D)   Jacobian (X, Y) of 2^D P (phase 1);  AF(D) the affine point (phase 2a)
//   ZI(D)   1 / Z_D, D = 0 .. 636 (ZI(0) = 1)
//   AC(c,j) addition j of chunk c: X1, Y1 (the accumulator before it), r, H; AZ(c,j) = 1 / Z1 before addition j (AZ(c, 0) = 1)
//   RES, ZE(s), ACS(c): what one segment's programs leave for the next
#define BLSW_COFV_XY(D) (4u * (uint32_t)(D))
#define BLSW_COFV_ZI(D) (4u * BLSW_H_EFF_NBITS + 2u * (uint32_t)(D))
#define BLSW_COFV_ACC0 (4u * BLSW_H_EFF_NBITS + 2u * (BLSW_H_EFF_NBITS + 1))
#define BLSW_COFV_AC(c, j) (BLSW_COFV_ACC0 + (uint32_t)(c) * 10u * (BLSW_COFV_MAX_ADDS + 1) + 10u * (uint32_t)(j))
#define BLSW_COFV_AZ(c, j) (BLSW_COFV_AC(c, j) + 8u)
#define BLSW_COFV_AFF0 (BLSW_COFV_ACC0 + 3u * 10u * (BLSW_COFV_MAX_ADDS + 1))
#define BLSW_COFV_AF(D) (BLSW_COFV_AFF0 + 4u * (uint32_t)(D))
#define BLSW_COFV_RES (BLSW_COFV_AFF0 + 4u * BLSW_H_EFF_NBITS)           // X, Y, Z of the doubling chain between two segments
#define BLSW_COFV_ZE(s) (BLSW_COFV_RES + 6u + 2u * (uint32_t)(s))        // Z of segment s's last point
#define BLSW_COFV_ACS(c) (BLSW_COFV_RES + 6u + 2u * BLSW_COFV_NSEG + 6u * (uint32_t)(c))  // X1, Y1, Z1 of chunk c's addition chain between two segments
#define BLSW_COFV_ELEMS (BLSW_COFV_RES + 6u + 2u * BLSW_COFV_NSEG + 18u)

template <class S>
BLSW_HD void cofv_st2(const S& s, uint32_t el, const Fp2& v) {
    s.st(el, v.c0);
    s.st(el + 1, v.c1);
}
template <class S>
BLSW_HD Fp2 cofv_ld2(const S& s, uint32_t el) {
    return {s.ld(el), s.ld(el + 1)};
}

// ---- phase 1, segment s (one lane, or one quad, per instance): [s = 0: Q0 + Q1 and to_affine with their witnesses, then] the segment's Jacobian doublings
// as values. `rows`: what the join reads (cofactor_par.hpp: row 36 = infinity flag)
template <class S, class ST, class Q>
BLSW_FN void cofv_chain_seg(int s, Emitter e_add, Emitter e, const Q& load_q, const S& scr, const ST& rows) {
    constexpr CofvSeg seg = cofv_seg();
    Fp2 X, Y, Z;
    if (s == 0) {
        Proj<OpsFp2> q0, q1;
        load_q(q0, q1);
        Proj<OpsFp2> r = proj_add_w<OpsFp2, 0>(e_add, q0, q1);
        Aff2Inf ra = g2_to_affine_w(e, r);
        rows.st(36, ra.infinity ? fp_one() : fp_zero());
        X = ra.x;
        Y = ra.y;
        Z = fp2_one();
        cofv_st2(scr, BLSW_COFV_XY(0), X);
        cofv_st2(scr, BLSW_COFV_XY(0) + 2, Y);
    } else {
        X = cofv_ld2(scr, BLSW_COFV_RES);
        Y = cofv_ld2(scr, BLSW_COFV_RES + 2);
        Z = cofv_ld2(scr, BLSW_COFV_RES + 4);
    }
    const int D1 = seg.bnd[s + 1];
#pragma unroll 1
    for (int D = seg.bnd[s]; D < D1; D++) {  // dbl-2009-l, a = 0 (vcurve.hpp: on a quad, four product rounds per step)
        v_dbl_inplace(X, Y, Z);
        if (D + 1 < BLSW_H_EFF_NBITS) {
            cofv_st2(scr, BLSW_COFV_XY(D + 1), X);
            cofv_st2(scr, BLSW_COFV_XY(D + 1) + 2, Y);
        }
    }
    cofv_st2(scr, BLSW_COFV_ZE(s), Z);
    if (s + 1 < BLSW_COFV_NSEG) {
        cofv_st2(scr, BLSW_COFV_RES, X);
        cofv_st2(scr, BLSW_COFV_RES + 2, Y);
        cofv_st2(scr, BLSW_COFV_RES + 4, Z);
    }
}
// ---- phase 1b, segment s (one lane, or one quad, per instance): 1 / Z_D for D in (D0, D1], backwards from the segment's last point (Z_{D+1} = 2 Y_D Z_D)
template <class S>
BLSW_FN void cofv_bwd(int s, const S& scr) {
    constexpr CofvSeg seg = cofv_seg();
    const int D0 = seg.bnd[s], D1 = seg.bnd[s + 1];
    Fp2 zi = fp2_inv_inl(cofv_ld2(scr, BLSW_COFV_ZE(s)));
    cofv_st2(scr, BLSW_COFV_ZI(D1), zi);
#pragma unroll 1
    for (int D = D1 - 1; D > D0; D--) {
        zi = fp2_mul_inl(fp2_dbl(cofv_ld2(scr, BLSW_COFV_XY(D) + 2)), zi);
        cofv_st2(scr, BLSW_COFV_ZI(D), zi);
    }
    if (s == 0) cofv_st2(scr, BLSW_COFV_ZI(0), fp2_one());
}

// ---- phase 2a (one lane per doubling index D of a segment, of an instance): the affine 2^D P — what the addition chains wait for
template <class S>
BLSW_FN void cofv_affine(uint32_t D, const S& scr) {
    const Fp2 X = cofv_ld2(scr, BLSW_COFV_XY(D)), Y = cofv_ld2(scr, BLSW_COFV_XY(D) + 2), zi = cofv_ld2(scr, BLSW_COFV_ZI(D));
    const Fp2 zi2 = v_sqr(zi);
    cofv_st2(scr, BLSW_COFV_AF(D), fp2_mul_inl(X, zi2));
    cofv_st2(scr, BLSW_COFV_AF(D) + 2, fp2_mul_inl(Y, fp2_mul_inl(zi2, zi)));
}
// ---- phase 2b (one lane per doubling D of an instance): the ten witnesses of nz_double_pre_inl on 2^D P. Nothing waits for it but the
// placement of the segment: the engine runs it beside the addition chains
template <class S>
BLSW_FN void cofv_dbl_w(Emitter e, uint32_t D, const S& scr) {
    constexpr CofvPlan plan = cofv_plan();
    const Fp2 X = cofv_ld2(scr, BLSW_COFV_XY(D)), zn = cofv_ld2(scr, BLSW_COFV_ZI(D + 1));
    const Aff2 p = {cofv_ld2(scr, BLSW_COFV_AF(D)), cofv_ld2(scr, BLSW_COFV_AF(D) + 2)};
    const Fp2 A = v_sqr(X);
    const Fp2 lambda = fp2_mul_inl(fp2_add(fp2_dbl(A), A), zn);
    e.pos += plan.pos_dbl[D];
    (void)fp2_sqr_w(e, p.x);
    e.put(lambda.c0);
    e.put(lambda.c1);
    fp_mul_w(e, lambda.c1, fp_dbl(p.y.c1));
    const Fp2 l2 = fp2_sqr_w(e, lambda);
    const Fp2 x3 = fp2_sub(l2, fp2_dbl(p.x));
    (void)fp2_mul_w(e, lambda, fp2_sub(p.x, x3));
}

// ---- phase 3, segment s (one lane, or one quad, per instance): the additions of the segment's chunk whose operand lies in the segment, as a mixed
// Jacobian chain; the chunk's last segment ends with the inversion and (acc, init) in the join's rows 12 c + {0.., 4..}
template <class S, class ST>
BLSW_FN void cofv_acc_seg(int s, const S& scr, const ST& rows) {
    constexpr CofvPlan plan = cofv_plan();
    constexpr CofvSeg seg = cofv_seg();
    const int c = seg.chunk[s];
    const int off = 255 * c;
    Fp2 X1, Y1, Z1;
    if (seg.first(s)) {
        X1 = cofv_ld2(scr, BLSW_COFV_AF(off));
        Y1 = cofv_ld2(scr, BLSW_COFV_AF(off) + 2);
        Z1 = fp2_one();
    } else {
        X1 = cofv_ld2(scr, BLSW_COFV_ACS(c));
        Y1 = cofv_ld2(scr, BLSW_COFV_ACS(c) + 2);
        Z1 = cofv_ld2(scr, BLSW_COFV_ACS(c) + 4);
    }
    const uint32_t j1 = seg.j_hi[s];
#pragma unroll 1
    for (uint32_t j = seg.j_lo[s]; j < j1; j++) {  // madd-2007-bl with Z3 = 2 Z1 H
        const uint32_t D = plan.add_bit[c][j];
        const Fp2 x2 = cofv_ld2(scr, BLSW_COFV_AF(D)), y2 = cofv_ld2(scr, BLSW_COFV_AF(D) + 2);
        const Fp2 z1z1 = v_sqr(Z1);
        const Fp2 u2 = fp2_mul_inl(x2, z1z1);
        const Fp2 s2 = fp2_mul_inl(fp2_mul_inl(y2, Z1), z1z1);
        const Fp2 H = fp2_sub(u2, X1);
        const Fp2 rr = fp2_dbl(fp2_sub(s2, Y1));
        cofv_st2(scr, BLSW_COFV_AC(c, j), X1);
        cofv_st2(scr, BLSW_COFV_AC(c, j) + 2, Y1);
        cofv_st2(scr, BLSW_COFV_AC(c, j) + 4, rr);
        cofv_st2(scr, BLSW_COFV_AC(c, j) + 6, H);
        const Fp2 hh = v_sqr(H);
        const Fp2 I = fp2_dbl(fp2_dbl(hh));
        const Fp2 J = fp2_mul_inl(H, I);
        const Fp2 V = fp2_mul_inl(X1, I);
        const Fp2 x3 = fp2_sub(fp2_sub(v_sqr(rr), J), fp2_dbl(V));
        const Fp2 y3 = fp2_sub(fp2_mul_inl(rr, fp2_sub(V, x3)), fp2_dbl(fp2_mul_inl(Y1, J)));
        Z1 = fp2_dbl(fp2_mul_inl(Z1, H));
        X1 = x3;
        Y1 = y3;
    }
    if (!seg.last(s)) {
        cofv_st2(scr, BLSW_COFV_ACS(c), X1);
        cofv_st2(scr, BLSW_COFV_ACS(c) + 2, Y1);
        cofv_st2(scr, BLSW_COFV_ACS(c) + 4, Z1);
        return;
    }
    const uint32_t na = plan.n_adds[c];
    const Aff2 init = {cofv_ld2(scr, BLSW_COFV_AF(off)), cofv_ld2(scr, BLSW_COFV_AF(off) + 2)};
    Fp2 zi = fp2_inv_inl(Z1);
    {  // the accumulator after the loop, affine
        const Fp2 zi2 = v_sqr(zi);
        const Aff2 acc = {fp2_mul_inl(X1, zi2), fp2_mul_inl(Y1, fp2_mul_inl(zi2, zi))};
        cof_st_aff(rows, 12 * c, na ? acc : init);
    }
    cof_st_aff(rows, 12 * c + 4, init);  // (the join takes the tail's operands from AF)
    cofv_st2(scr, BLSW_COFV_AZ(c, na), zi);
}
// ---- phase 3b (one lane, or one quad, per chunk of an instance): 1 / Z1 before every addition of the chunk, backwards from the one after the last
// (Z1_{j+1} = 2 Z1_j H_j). Only the additions' witnesses read it: off the critical path
template <class S>
BLSW_FN void cofv_acc_az(int c, const S& scr) {
    constexpr CofvPlan plan = cofv_plan();
    const uint32_t na = plan.n_adds[c];
    if (na == 0) return;
    Fp2 zi = cofv_ld2(scr, BLSW_COFV_AZ(c, na));
#pragma unroll 1
    for (int j = (int)na - 1; j >= 1; j--) {
        zi = fp2_mul_inl(fp2_dbl(cofv_ld2(scr, BLSW_COFV_AC(c, j) + 6)), zi);
        cofv_st2(scr, BLSW_COFV_AZ(c, j), zi);
    }
    cofv_st2(scr, BLSW_COFV_AZ(c, 0), fp2_one());
}

// ---- phase 4 (one lane per addition j of chunk c of an instance): the eight witnesses of nz_add_unchecked_pre_inl
template <class S>
BLSW_FN void cofv_add_w(Emitter e, int c, uint32_t j, const S& scr) {
    constexpr CofvPlan plan = cofv_plan();
    const uint32_t D = plan.add_bit[c][j];
    const Fp2 X1 = cofv_ld2(scr, BLSW_COFV_AC(c, j)), Y1 = cofv_ld2(scr, BLSW_COFV_AC(c, j) + 2), rr = cofv_ld2(scr, BLSW_COFV_AC(c, j) + 4);
    const Fp2 zi = cofv_ld2(scr, BLSW_COFV_AZ(c, j)), zn = cofv_ld2(scr, BLSW_COFV_AZ(c, j + 1));
    const Fp2 zi2 = v_sqr(zi);
    const Aff2 p = {fp2_mul_inl(X1, zi2), fp2_mul_inl(Y1, fp2_mul_inl(zi2, zi))};
    const Fp2 qx = cofv_ld2(scr, BLSW_COFV_AF(D));
    const Fp2 lambda = fp2_mul_inl(rr, zn);
    e.pos += plan.pos_add[c][j];
    e.put(lambda.c0);
    e.put(lambda.c1);
    fp_mul_w(e, lambda.c1, fp_sub(qx.c1, p.x.c1));
    const Fp2 l2 = fp2_sqr_w(e, lambda);
    const Fp2 x3 = fp2_sub(fp2_sub(l2, p.x), qx);
    (void)fp2_mul_w(e, lambda, fp2_sub(p.x, x3));
}

// ---- phase 5 (one lane per instance): chain_cofactor_join's statements; the tail doublings of chunks 0 and 1 were emitted by phase 2,
// their operands are the affine points of phase 2a
template <class S, class LD>
BLSW_FN Proj<OpsFp2> cofv_join(Emitter e, const S& scr, const LD& load) {
    constexpr CofactorPlan plan = cofactor_plan();
    constexpr uint32_t HE[20] = BLSW_H_EFF_WORDS;
    const uint32_t pos0 = e.pos;
    const bool infinity = !fp_is_zero(load.ld(36));
    Proj<OpsFp2> mul_result = {fp2_zero(), fp2_one(), fp2_zero()};
    int mr_state = -1;
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        const int off = 255 * c;
        const int n = BLSW_H_EFF_NBITS - off < 255 ? BLSW_H_EFF_NBITS - off : 255;
        const int split = n < 253 ? n : 253;
        e.pos = pos0 + plan.start[c] + plan.loop[c];
        const Aff2 acc = cof_ld_aff(load, 12 * c), init = cof_ld_aff(load, 12 * c + 4);
        Proj<OpsFp2> diff = {acc.x, acc.y, fp2_one()};
        int diff_state = 2;
        if (!bit_of(HE, off)) {
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
                Proj<OpsFp2> m = {cofv_ld2(scr, BLSW_COFV_AF(off + i)), cofv_ld2(scr, BLSW_COFV_AF(off + i) + 2), fp2_one()};
                mul_result = proj_add_zstate_w(e, mul_result, mr_state, m, 2);
                mr_state = 0;
            }
            e.pos += 10;  // nz_double_w(mopt): phase 2
        }
    }
    Proj<OpsFp2> h;
    h.x = fp2_select_w(e, infinity, fp2_zero(), mul_result.x);
    h.y = fp2_select_w(e, infinity, fp2_one(), mul_result.y);
    if (mr_state == 2)
        h.z = infinity ? fp2_zero() : fp2_one();
    else
        h.z = fp2_select_w(e, infinity, fp2_zero(), mul_result.z);
    return h;
}

}  // namespace blsw
