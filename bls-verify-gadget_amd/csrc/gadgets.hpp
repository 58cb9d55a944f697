// Witness-emitting field "gadgets" for one instance per lane.
// Each helper computes the value(s) that ark-r1cs-std ^0.4.0 would push onto
// ConstraintSystem::witness_assignment for the corresponding Var x Var operation, in the same order,
// and stores them (48 B, Montgomery, little-endian limbs) at the lane's cursor.
// Operations with a constant operand are linear combinations in the circuit (no witness): callers use
// the plain value functions of fp.hpp for those.  Rules: SURVEY.md App. A.1, A.2.
#pragma once
#include "fp.hpp"

namespace blsw {

#if defined(__HIPCC__)  // both passes: device functions are parsed for the host as well
typedef uint32_t blsw_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) blsw_u32x4 blsw_global_u32x4;
#endif
struct Emitter {
    uint32_t* base;        // where element 0 of this instance lives (16-byte aligned); nullptr = value-only mode
    uint32_t pos;          // element index of the next witness
    uint64_t stride = 12;  // distance, in u32, between consecutive elements: 12 = dense vector (instance-major);
                           // 12 * N = element-major staging shared by N instances (coalesced across lanes)
    // element `at` of this instance := v (no cursor movement, no mode check)
    BLSW_HD void store_at(uint32_t at, const Fp& v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        // witnesses live in global memory (staging or an output tensor): say so — a generic pointer makes these FLAT stores, which count on
        // lgkmcnt as well and hold up every later wait for an LDS operation (the team kernels) until their addresses are resolved
        blsw_global_u32x4* d = (blsw_global_u32x4*)(base + (size_t)at * stride);
        d[0] = blsw_u32x4{v.l[0], v.l[1], v.l[2], v.l[3]};
        d[1] = blsw_u32x4{v.l[4], v.l[5], v.l[6], v.l[7]};
        d[2] = blsw_u32x4{v.l[8], v.l[9], v.l[10], v.l[11]};
#else
        uint32_t* d = base + (size_t)at * stride;
        for (int i = 0; i < 12; i++) d[i] = v.l[i];
#endif
    }
    BLSW_HD void put(const Fp& v) {
        if (base == nullptr) {  // value-only mode (hash_to_g2 batch): no store, cursor still advances
            pos++;
            return;
        }
#ifdef BLSW_QUAD_DEV
        if (quad_role() == 0)  // the four lanes of a quad hold the same value: one of them stores it
#endif
            store_at(pos, v);
        pos++;
    }
#ifdef BLSW_QUAD_DEV
    // every lane of the quad with role < n stores ITS value at pos + role (the split products of an Fp2 operation); cursor += n
    BLSW_HD void put_split(const Fp& mine, uint32_t n) {
        const uint32_t role = quad_role();
        if (base != nullptr && role < n) store_at(pos + role, mine);
        pos += n;
    }
#endif
    BLSW_HD void put_bool(bool b) {
        Fp one = fp_one();
        Fp v;
#pragma unroll
        for (int i = 0; i < 12; i++) v.l[i] = b ? one.l[i] : 0u;
        put(v);
    }
};

// ---- FpVar
BLSW_HD Fp fp_mul_w(Emitter& e, const Fp& a, const Fp& b) {
    Fp r = fp_mul(a, b);
    e.put(r);
    return r;
}
// AllocatedFp::is_neq(self, other): witnesses = [is_not_equal (boolean), multiplier]; returns is_eq.
// Orientation matters: (Var v).is_eq(Constant c) is evaluated as c.is_eq(v), i.e. diff = c - v.
BLSW_FN bool fp_is_eq_w(Emitter& e, const Fp& self, const Fp& other) {
    Fp diff = fp_sub(self, other);
    bool ne = !fp_is_zero(diff);
    e.put_bool(ne);
    Fp m = fp_inv(diff);  // uniform instruction stream: computed even when equal (result discarded)
    e.put(ne ? m : fp_one());
    return !ne;
}
BLSW_HD Fp fp_select_w(Emitter& e, bool cond, const Fp& t, const Fp& f) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = cond ? t.l[i] : f.l[i];
    e.put(r);
    return r;
}

// ---- Fp2Var (QuadExtVar over FpVar)
#ifdef BLSW_QUAD_DEV
// the three (two) product witnesses are computed and stored by three (two) lanes of the quad (fp.hpp: fp2_mul_quad)
BLSW_HD Fp2 fp2_mul_w(Emitter& e, const Fp2& a, const Fp2& b) {
    Fp prod;
    const Fp2 r = fp2_mul_quad(a, b, prod);
    e.put_split(prod, 3);
    return r;
}
BLSW_HD Fp2 fp2_sqr_w(Emitter& e, const Fp2& a) {
    Fp prod;
    const Fp2 r = fp2_sqr_quad(a, prod);
    e.put_split(prod, 2);
    return r;
}
#else
BLSW_HD Fp2 fp2_mul_w(Emitter& e, const Fp2& a, const Fp2& b) {
    Fp v0 = fp_mul_w(e, a.c0, b.c0);
    Fp v1 = fp_mul_w(e, a.c1, b.c1);
    Fp s = fp_mul_w(e, fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    return {fp_sub(v0, v1), fp_sub(fp_sub(s, v0), v1)};
}
BLSW_HD Fp2 fp2_sqr_w(Emitter& e, const Fp2& a) {
    Fp v2 = fp_mul_w(e, a.c0, a.c1);
    Fp t = fp_mul_w(e, fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1));
    return {t, fp_dbl(v2)};
}
#endif
// QuadExtVar::inverse: witnesses inv.c0, inv.c1, then mul_equals' v1 = a.c1 * inv.c1
BLSW_FN Fp2 fp2_inv_w(Emitter& e, const Fp2& a) {
    Fp2 inv = fp2_inv(a);
    e.put(inv.c0);
    e.put(inv.c1);
    fp_mul_w(e, a.c1, inv.c1);
    return inv;
}
// FieldVar::mul_by_inverse_unchecked: witnesses r = num/den (c0, c1), then v1 = r.c1 * den.c1
BLSW_FN Fp2 fp2_div_w(Emitter& e, const Fp2& num, const Fp2& den) {
    Fp2 r = fp2_mul(num, fp2_inv(den));
    e.put(r.c0);
    e.put(r.c1);
    fp_mul_w(e, r.c1, den.c1);
    return r;
}
// same with the inverse of `den` supplied by the caller (batched inversions): identical witnesses
BLSW_FN Fp2 fp2_div_pre_w(Emitter& e, const Fp2& num, const Fp2& den, const Fp2& den_inv) {
    Fp2 r = fp2_mul(num, den_inv);
    e.put(r.c0);
    e.put(r.c1);
    fp_mul_w(e, r.c1, den.c1);
    return r;
}
// Montgomery's trick for two independent Fp2 inversions: one Fp inversion + 9 extra Fp products instead of two inversions.
// Falls back to separate inversions when either input is zero (the hint of a zero is zero).
BLSW_HD void fp2_inv2_inl(const Fp2& a, const Fp2& b, Fp2& a_inv, Fp2& b_inv) {
    if (fp2_is_zero(a) || fp2_is_zero(b)) {
        a_inv = fp2_inv(a);
        b_inv = fp2_inv(b);
        return;
    }
    Fp2 p = fp2_mul_inl(a, b);
    Fp2 pi = fp2_inv_inl(p);
    a_inv = fp2_mul_inl(b, pi);
    b_inv = fp2_mul_inl(a, pi);
}
BLSW_FN void fp2_inv2(const Fp2& a, const Fp2& b, Fp2& a_inv, Fp2& b_inv) { fp2_inv2_inl(a, b, a_inv, b_inv); }
#ifdef BLSW_QUAD_DEV
// the two component tests (an inversion each) on two lanes of the quad: lane 0 stores [ne0, m0], lane 1 [ne1, m1]
BLSW_FN bool fp2_is_eq_w(Emitter& e, const Fp2& self, const Fp2& other) {
    const uint32_t role = quad_role();
    const Fp diff = fp_sub(quad_sel2(role, self.c0, self.c1), quad_sel2(role, other.c0, other.c1));
    const bool ne = !fp_is_zero(diff);
    const Fp m = fp_inv(diff);
    if (e.base != nullptr && role < 2) {
        const Fp one = fp_one();
        Fp flag, mult;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            flag.l[i] = ne ? one.l[i] : 0u;
            mult.l[i] = ne ? m.l[i] : one.l[i];
        }
        e.store_at(e.pos + 2 * role, flag);
        e.store_at(e.pos + 2 * role + 1, mult);
    }
    e.pos += 4;
    const bool r = !quad_bcast_u32<0>(ne ? 1u : 0u) && !quad_bcast_u32<1>(ne ? 1u : 0u);
    e.put_bool(r);
    return r;
}
#else
BLSW_FN bool fp2_is_eq_w(Emitter& e, const Fp2& self, const Fp2& other) {
    bool b0 = fp_is_eq_w(e, self.c0, other.c0);
    bool b1 = fp_is_eq_w(e, self.c1, other.c1);
    bool r = b0 && b1;  // Not(ne0) AND Not(ne1) -> nor witness
    e.put_bool(r);
    return r;
}
#endif
// v.is_eq(Constant zero) / v.is_zero(): evaluated as zero.is_eq(v)
BLSW_HD bool fp2_is_zero_w(Emitter& e, const Fp2& v) { return fp2_is_eq_w(e, fp2_zero(), v); }
BLSW_HD Fp2 fp2_select_w(Emitter& e, bool cond, const Fp2& t, const Fp2& f) {
    Fp c0 = fp_select_w(e, cond, t.c0, f.c0);
    Fp c1 = fp_select_w(e, cond, t.c1, f.c1);
    return {c0, c1};
}

// ---- FpVar::to_bits_le on a variable: 381 boolean witnesses (LSB first) followed by
// Boolean::enforce_in_field_le's AND chain against p-1. Returns bit 0.
BLSW_FN bool fp_to_bits_le_w(Emitter& e, const Fp& a) {
    constexpr uint32_t P[12] = BLSW_P_LIMBS;
    Fp c = fp_to_canonical(a);
#pragma unroll 1
    for (int i = 0; i < 381; i++) e.put_bool((c.l[i >> 5] >> (i & 31)) & 1);
    // walk p-1 from the top bit; last_run starts as the constant true
    bool last_run = true, last_is_const = true;
    bool run_acc = true;
    int run_len = 0;
#pragma unroll 1
    for (int i = 380; i >= 0; i--) {
        uint32_t w = P[i >> 5];
        if (i < 32) w -= 1;
        bool pb = (w >> (i & 31)) & 1;
        bool ab = (c.l[i >> 5] >> (i & 31)) & 1;
        if (pb) {
            // kary_and of the run is built left to right: each further element costs one AND witness
            if (run_len == 0)
                run_acc = ab;
            else {
                run_acc = run_acc && ab;
            }
            run_len++;
            // AND witnesses inside a run are emitted when the run closes (order: run elements, then last_run)
        } else {
            if (run_len > 0) {
                // kary_and([run..., last_run]): re-walk the run to emit its (run_len-1) partial ANDs in order
                bool acc = true;
                for (int k = 0; k < run_len; k++) {
                    int bi = i + run_len - k;  // bits i+run_len .. i+1, high to low
                    bool bb = (c.l[bi >> 5] >> (bi & 31)) & 1;
                    if (k == 0)
                        acc = bb;
                    else {
                        acc = acc && bb;
                        e.put_bool(acc);
                    }
                }
                if (!last_is_const) {
                    acc = acc && last_run;
                    e.put_bool(acc);
                }
                last_run = acc;
                last_is_const = false;
                run_len = 0;
            }
            // enforce_kary_nand([last_run, a]) -> one AND witness unless last_run is still the constant
            if (!last_is_const) e.put_bool(last_run && ab);
        }
    }
    (void)run_acc;
    return c.l[0] & 1;
}

}  // namespace blsw
