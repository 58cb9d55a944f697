// clear_cofactor2 (hasher.rs:664-673) with its three 255-bit chunks on three lanes.
//
// scalar_mul_le walks the 636 bits of h_eff in chunks of 255: within a chunk an affine double-and-add loop over all
// but the last two bits, then a handful of projective additions that fold the chunk into the running result and the chunk's last two
// doublings. The loop of chunk c starts from mopt_c = 2^(255 c) * P and is otherwise independent of the other chunks; only the folding
// additions chain the chunks together. As ONE chain (chains.hpp: chain_cofactor, the statement of the segment) that is 636 dependent
// affine steps with an inversion each — 53 ms alone, the longest kernel of every launch group. Here:
//   phase A (lane per chunk)   chunk 0 emits Q0 + Q1 and the to_affine of the sum; chunks 1, 2 compute their start point as a VALUE
//                              (the same sum and affine point without a cursor, then 255 c Jacobian doublings and one inversion:
//                              field elements are canonical, so it is bit for bit the point the affine chain reaches); every lane then
//                              runs its chunk's loop with the witness cursor at the chunk's place and leaves (acc, init, mopt);
//   phase B (lane per instance) the folding additions, the two tail doublings of chunks 0 and 1 and the final select, ~700 products.
// Critical path: 4.1 k (start of chunk 1) + 12.7 k products instead of 32 k. The witness offsets of the chunks are a compile-time walk
// over the bits of h_eff (cofactor_plan), checked against the segment length.
// Compiles for the host as well: tests/hostsim runs the phases in any order against the oracle.
#pragma once
#include "chains.hpp"
#include "vcurve.hpp"

namespace blsw {

struct CofactorPlan {
    uint32_t start[3];  // first witness of chunk c's loop, relative to the cofactor segment
    uint32_t loop[3];   // witnesses of the loop part of chunk c (phase A)
    uint32_t total;     // whole segment (to_affine + chunks + final select)
};
constexpr bool h_eff_bit(int i) {
    constexpr uint32_t HE[20] = BLSW_H_EFF_WORDS;
    return (HE[i >> 5] >> (i & 31)) & 1;
}
constexpr CofactorPlan cofactor_plan() {
    CofactorPlan p = {};
    uint32_t pos = 18;  // g2_to_affine_w: is_zero 5, z_inv 2 + 1, two products 6, two selects 4
    int mr_state = -1;
    int c = 0;
    for (int off = 0; off < BLSW_H_EFF_NBITS; off += 255, c++) {
        const int n = BLSW_H_EFF_NBITS - off < 255 ? BLSW_H_EFF_NBITS - off : 255;
        const int split = n < 253 ? n : 253;
        p.start[c] = pos;
        pos += 10;  // nz_double_w of bit 0
        for (int i = 1; i < split; i++) pos += 10 + (h_eff_bit(off + i) ? 8 : 0);
        p.loop[c] = pos - p.start[c];
        int diff_state = 2;
        if (!h_eff_bit(off)) {
            pos += 33;  // proj_add_w<2>(acc, -init)
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
                pos += 33;  // + (mopt, 1): z2 is the constant one
                mr_state = 0;
            }
            pos += 10;
        }
    }
    pos += mr_state == 2 ? 4 : 6;  // infinity.select(zero, mul_result)
    p.total = pos;
    return p;
}
static_assert(cofactor_plan().total == 8979, "cofactor segment: the plan must count what chain_cofactor emits (layout.h: SEG_COFACTOR)");

// what phase A leaves for phase B: ST / LD are row stores with st(idx, Fp) / ld(idx) (CoeffStrided on the device, CoeffLinear on the host);
// rows 12 c + {0..3 acc, 4..7 init, 8..11 mopt after the loop}, row 36 = the infinity flag of the sum (0 / one)
#define BLSW_COFACTOR_ROWS 37
template <class ST>
BLSW_HD void cof_st_aff(const ST& s, uint32_t row, const Aff2& a) {
    s.st(row, a.x.c0);
    s.st(row + 1, a.x.c1);
    s.st(row + 2, a.y.c0);
    s.st(row + 3, a.y.c1);
}
template <class LD>
BLSW_HD Aff2 cof_ld_aff(const LD& s, uint32_t row) {
    return {{s.ld(row), s.ld(row + 1)}, {s.ld(row + 2), s.ld(row + 3)}};
}

// phase A, chunk c of one instance. e_add: cursor of the "add" segment, e: cursor at the start of the cofactor segment.
template <class ST>
BLSW_FN void chain_cofactor_chunk(Emitter e_add, Emitter e, const Proj<OpsFp2>& q0, const Proj<OpsFp2>& q1, int c, const ST& store) {
    constexpr CofactorPlan plan = cofactor_plan();
    constexpr uint32_t HE[20] = BLSW_H_EFF_WORDS;
    Emitter ev_add = e_add, ev = e;  // cursors of the sum and its affine form: real for chunk 0 (they are its witnesses), dummies otherwise
    if (c != 0) {
        ev_add.base = nullptr;
        ev.base = nullptr;
    }
    Proj<OpsFp2> r = proj_add_w<OpsFp2, 0>(ev_add, q0, q1);
    Aff2Inf ra = g2_to_affine_w(ev, r);
    Aff2 mopt = {ra.x, ra.y};
    if (c == 0) {
        Fp flag = ra.infinity ? fp_one() : fp_zero();
        store.st(36, flag);
    } else if (!ra.infinity) {  // mopt = 2^(255 c) * (x, y): Jacobian doublings, one inversion
        Jac2 j = {ra.x, ra.y, fp2_one()};
#pragma unroll 1
        for (int i = 0; i < 255 * c; i++) j = v_dbl(j);
        const Fp2 zi = fp2_inv_inl(j.z), zi2 = v_sqr(zi);
        mopt = {fp2_mul_inl(j.x, zi2), fp2_mul_inl(j.y, fp2_mul_inl(zi2, zi))};
    } else {  // the sum is the identity: the circuit's chain runs on (0, 0) with zero hints — follow it step by step as values
        Emitter dummy = {nullptr, 0};
#pragma unroll 1
        for (int i = 0; i < 255 * c; i++) mopt = nz_double_w(dummy, mopt);
    }
    // the chunk's loop, witnesses at the chunk's place
    Emitter w = e;
    w.pos = e.pos + plan.start[c];
    const int off = 255 * c;
    const int n = BLSW_H_EFF_NBITS - off < 255 ? BLSW_H_EFF_NBITS - off : 255;
    const int split = n < 253 ? n : 253;
    Aff2 acc = mopt;
    cof_st_aff(store, 12 * c + 4, mopt);  // init
    mopt = nz_double_w(w, mopt);
#pragma unroll 1
    for (int i = 1; i < split; i++) {
        const bool add = bit_of(HE, off + i);
        Fp2 inv_add, inv_dbl;
        if (add) {
            fp2_inv2_inl(fp2_sub(mopt.x, acc.x), fp2_dbl(mopt.y), inv_add, inv_dbl);
            acc = nz_add_unchecked_pre_inl(w, acc, mopt, inv_add);
        } else {
            inv_dbl = fp2_inv_inl(fp2_dbl(mopt.y));
        }
        mopt = nz_double_pre_inl(w, mopt, inv_dbl);
    }
    cof_st_aff(store, 12 * c, acc);
    cof_st_aff(store, 12 * c + 8, mopt);
}

// phase B: folds the chunks (chain_cofactor's statements after each loop), cursor at the start of the cofactor segment
template <class LD>
BLSW_FN Proj<OpsFp2> chain_cofactor_join(Emitter e, const LD& load) {
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
        Aff2 mopt = cof_ld_aff(load, 12 * c + 8);
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
                Proj<OpsFp2> m = {mopt.x, mopt.y, fp2_one()};
                mul_result = proj_add_zstate_w(e, mul_result, mr_state, m, 2);
                mr_state = 0;
            }
            mopt = nz_double_w(e, mopt);
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
