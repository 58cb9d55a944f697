// Six lanes per instance ("team") for the Fp12 part of the path: miller_loop, final_exponentiation, is_one
// (constraints.rs:121-127). Same witnesses, in the same order, as chains.hpp::chain_miller /
// chain_final_exp_is_one, which stay as the single-lane statement of the segment (host harness, aggregate path).
//
// Lane j of a team owns the Fp2 coefficient j of every Fp12 value, (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2), in
// registers. A tower operation is one pass over an op table (team_tables.hpp, generated from the tower formulas):
//   publish the operand coefficients in the team's slot file (LDS)  ->  rounds of <= 6 independent Fp2 products, one per
//   lane, operands gathered as small linear combinations of slots, witnesses stored at the task's offset  ->  every lane
//   gathers its coefficient of the result from the product slots.
// Conjugation, Frobenius maps and additions are lane-local. Nothing of an Fp12 value ever lives on the stack.
//
// The program (team_miller, team_final_exp_is_one) is written once against a TEAM interface; TeamLanes is the device
// implementation (one lane per thread, wave barrier between phases), tests/hostsim runs the same program and the same
// lane routines with a loop over the six lanes.
#pragma once
#include "team_tables.hpp"
#include "tower.hpp"

namespace blsw {

// ---- slot file access. Slot s of a team = one Fp2 (96 bytes, 16-byte aligned).
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t team_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) team_u32x4 team_lds_u32x4;
BLSW_HD Fp2 team_ld(const Fp2* slots, uint32_t s) {
    const team_lds_u32x4* p = (const team_lds_u32x4*)(slots + s);
    team_u32x4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3], v4 = p[4], v5 = p[5];
    Fp2 r;
    r.c0.l[0] = v0.x; r.c0.l[1] = v0.y; r.c0.l[2] = v0.z; r.c0.l[3] = v0.w;
    r.c0.l[4] = v1.x; r.c0.l[5] = v1.y; r.c0.l[6] = v1.z; r.c0.l[7] = v1.w;
    r.c0.l[8] = v2.x; r.c0.l[9] = v2.y; r.c0.l[10] = v2.z; r.c0.l[11] = v2.w;
    r.c1.l[0] = v3.x; r.c1.l[1] = v3.y; r.c1.l[2] = v3.z; r.c1.l[3] = v3.w;
    r.c1.l[4] = v4.x; r.c1.l[5] = v4.y; r.c1.l[6] = v4.z; r.c1.l[7] = v4.w;
    r.c1.l[8] = v5.x; r.c1.l[9] = v5.y; r.c1.l[10] = v5.z; r.c1.l[11] = v5.w;
    return r;
}
BLSW_HD void team_st(Fp2* slots, uint32_t s, const Fp2& v) {
    team_lds_u32x4* p = (team_lds_u32x4*)(slots + s);
    p[0] = team_u32x4{v.c0.l[0], v.c0.l[1], v.c0.l[2], v.c0.l[3]};
    p[1] = team_u32x4{v.c0.l[4], v.c0.l[5], v.c0.l[6], v.c0.l[7]};
    p[2] = team_u32x4{v.c0.l[8], v.c0.l[9], v.c0.l[10], v.c0.l[11]};
    p[3] = team_u32x4{v.c1.l[0], v.c1.l[1], v.c1.l[2], v.c1.l[3]};
    p[4] = team_u32x4{v.c1.l[4], v.c1.l[5], v.c1.l[6], v.c1.l[7]};
    p[5] = team_u32x4{v.c1.l[8], v.c1.l[9], v.c1.l[10], v.c1.l[11]};
}
#else
BLSW_HD Fp2 team_ld(const Fp2* slots, uint32_t s) { return slots[s]; }
BLSW_HD void team_st(Fp2* slots, uint32_t s, const Fp2& v) { slots[s] = v; }
#endif

// slot number k of a descriptor held in registers
BLSW_HD uint32_t team_lin_idx(const TeamLin& L, uint32_t k) {
    uint64_t lo = L.idx[0] | ((uint64_t)L.idx[1] << 32), hi = L.idx[2] | ((uint64_t)L.idx[3] << 32);
    uint64_t w = (k & 8) ? hi : lo;
    return (uint32_t)(w >> ((k & 7) * 8)) & 0xffu;
}
// (sum Lp - sum Ln) + xi * (sum Mp - sum Mn)
BLSW_HD Fp2 team_gather(const TeamLin& L, const Fp2* slots) {
    Fp2 acc = fp2_zero();
    uint32_t k = 0;
    const uint32_t n0 = L.n & 0x3f, n1 = (L.n >> 8) & 0xff, n2 = (L.n >> 16) & 0xff, n3 = L.n >> 24;
    const uint32_t mscale = (L.n >> 6) & 3;  // the xi part times 1, 12 or 24 (3b of the twist: 12 xi)
    if (n0) {
        acc = team_ld(slots, team_lin_idx(L, 0));
        k = 1;
        for (; k < n0; k++) acc = fp2_add(acc, team_ld(slots, team_lin_idx(L, k)));
    }
    for (uint32_t e = k + n1; k < e; k++) acc = fp2_sub(acc, team_ld(slots, team_lin_idx(L, k)));
    if (n2 + n3) {
        Fp2 m = fp2_zero();
        for (uint32_t e = k + n2; k < e; k++) m = fp2_add(m, team_ld(slots, team_lin_idx(L, k)));
        for (uint32_t e = k + n3; k < e; k++) m = fp2_sub(m, team_ld(slots, team_lin_idx(L, k)));
        if (mscale) {
            Fp2 m3 = fp2_add(fp2_dbl(m), m);
            Fp2 m12 = fp2_dbl(fp2_dbl(m3));
            m = mscale == 2 ? fp2_dbl(m12) : m12;
        }
        acc = fp2_add(acc, fp2_mul_xi(m));
    }
    return acc;
}

// one product task of one lane. `e` = cursor of the op (pos = first witness of the op)
BLSW_HD void team_task(const TeamTask& t, Fp2* slots, const Emitter& e) {
    const uint32_t kind = t.hdr & 0xff, dst = (t.hdr >> 8) & 0xff;
    if (kind == TK_NONE) return;
    Fp2 a = team_gather(t.a, slots), b = team_gather(t.b, slots);
    Emitter w = e;
    w.pos += t.hdr >> 16;
    if (kind == TK_K3V || kind == TK_K2V) w.base = nullptr;  // same arithmetic, a linear combination in the circuit
    // the witnesses of a task are contiguous in the instance's segment: computed first, then stored back to back
    // (144 contiguous bytes per lane and 864 per team and round reach L2 within one burst)
    Fp2 r = fp2_zero();
    if (kind == TK_K3 || kind == TK_K3V || kind == TK_K2S) {
        // one instruction stream for the kinds that share rounds:
        //   K3 / K3V  Karatsuba: a0*b0, a1*b1, (a0+a1)*(b0+b1)           -> (p0 - p1, p2 - p0 - p1), 3 witnesses
        //   K2S       Fp2 square: a0*a1, (a0-a1)*(a0+a1), (third unused) -> (p1, 2 p0),              2 witnesses
        const bool sq = kind == TK_K2S;
        const Fp sum = fp_add(a.c0, a.c1), dif = fp_sub(a.c0, a.c1), bsum = fp_add(b.c0, b.c1);
        Fp y0, x1, y1;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            y0.l[i] = sq ? a.c1.l[i] : b.c0.l[i];
            x1.l[i] = sq ? dif.l[i] : a.c1.l[i];
            y1.l[i] = sq ? sum.l[i] : b.c1.l[i];
        }
        Fp p0 = fp_mul(a.c0, y0), p1 = fp_mul(x1, y1);
        Fp p2 = fp_mul(sum, bsum);
        w.put(p0);
        w.put(p1);
        if (!sq) w.put(p2);
        Fp2 rk = {fp_sub(p0, p1), fp_sub(fp_sub(p2, p0), p1)};
        Fp2 rs = {p1, fp_dbl(p0)};
#pragma unroll
        for (int i = 0; i < 12; i++) {
            r.c0.l[i] = sq ? rs.c0.l[i] : rk.c0.l[i];
            r.c1.l[i] = sq ? rs.c1.l[i] : rk.c1.l[i];
        }
    } else if (kind == TK_K2 || kind == TK_K2V || kind == TK_K2B) {
        // two Fp products by the same y = b.c0, one instruction stream for the three kinds (they share rounds):
        //   K2 / K2V  Fp2 x (y, 0): a.c0*y, (a.c0+a.c1)*y -> (v0, s - v0)      K2B  a.c0*y, a.c1*y -> (v0, s)
        const bool sep = kind == TK_K2B;
        Fp x1 = fp_add(a.c0, a.c1);
#pragma unroll
        for (int i = 0; i < 12; i++) x1.l[i] = sep ? a.c1.l[i] : x1.l[i];
        Fp v0 = fp_mul(a.c0, b.c0);
        Fp s = fp_mul(x1, b.c0);
        w.put(v0);
        w.put(s);
        Fp d = fp_sub(s, v0);
#pragma unroll
        for (int i = 0; i < 12; i++) d.l[i] = sep ? s.l[i] : d.l[i];
        r = {v0, d};
    } else {  // TK_K1E: QuadExtVar::mul_equals over Fp, only a.c1*b.c1 is a witness
        w.put(fp_mul(a.c1, b.c1));
    }
    if (dst != 0xff) team_st(slots, dst, r);
}

// lane-local maps
BLSW_HD Fp2 team_conj(uint32_t j, const Fp2& v) { return j >= 3 ? fp2_neg(v) : v; }
// coefficient j of frobenius^POWER: conjugate (odd powers), then one constant per lane
BLSW_FN Fp2 team_frob(uint32_t j, const Fp2& v, int power) {
    Fp2 x = (power & 1) ? fp2_conj(v) : v;
    const uint32_t k = j % 3;
    Fp2 c6 = fp2_one();
    if (k == 1) c6 = power == 1 ? K_FROB6_C1_1() : (power == 2 ? K_FROB6_C1_2() : K_FROB6_C1_3());
    if (k == 2) c6 = power == 1 ? K_FROB6_C2_1() : (power == 2 ? K_FROB6_C2_2() : K_FROB6_C2_3());
    if (k != 0) x = fp2_mul(x, c6);
    if (j >= 3) x = fp2_mul(x, power == 1 ? K_FROB12_C1_1() : (power == 2 ? K_FROB12_C1_2() : K_FROB12_C1_3()));
    return x;
}
// is_one, per lane: one.is_eq(result) on this lane's coefficient: 5 witnesses at h*17 + k*5 of the segment
BLSW_HD bool team_is_one_coeff(uint32_t j, const Fp2& v, const Emitter& e_one) {
    Emitter w = e_one;
    w.pos += (j / 3) * 17 + (j % 3) * 5;
    return fp2_is_eq_w(w, j == 0 ? fp2_one() : fp2_zero(), v);
}
// AND tree of fp6_is_eq x2 + the final AND, from the six per-coefficient flags: lane 0 / lane 3 write their half
BLSW_HD bool team_is_one_tree(uint32_t j, const bool* b, const Emitter& e_one) {
    bool t0 = b[0] && b[1], r0 = t0 && b[2], t1 = b[3] && b[4], r1 = t1 && b[5];
    bool res = r0 && r1;
    Emitter w = e_one;
    if (j == 0) {
        w.pos += 15;
        w.put_bool(t0);
        w.put_bool(r0);
        w.pos = e_one.pos + 34;
        w.put_bool(res);
    } else if (j == 3) {
        w.pos += 17 + 15;
        w.put_bool(t1);
        w.put_bool(r1);
    }
    return res;
}
BLSW_FN void team_inverse_lane0(Fp2* slots) {  // IN0 = a  ->  IN1 = a^-1 (value only: the hint)
    Fp12 a = {{team_ld(slots, TS_IN0 + 0), team_ld(slots, TS_IN0 + 1), team_ld(slots, TS_IN0 + 2)},
              {team_ld(slots, TS_IN0 + 3), team_ld(slots, TS_IN0 + 4), team_ld(slots, TS_IN0 + 5)}};
    Fp12 inv = fp12_inv(a);
    team_st(slots, TS_IN1 + 0, inv.c0.c0);
    team_st(slots, TS_IN1 + 1, inv.c0.c1);
    team_st(slots, TS_IN1 + 2, inv.c0.c2);
    team_st(slots, TS_IN1 + 3, inv.c1.c0);
    team_st(slots, TS_IN1 + 4, inv.c1.c1);
    team_st(slots, TS_IN1 + 5, inv.c1.c2);
}
// coefficient j of f after the first ell on the constant f = 1 (no witnesses): (c0, c1*g1.x, 0, 0, (-g1.y, 0), 0)
BLSW_HD Fp2 team_first_f(uint32_t j, const Fp2* slots) {
    if (j == 0) return team_ld(slots, TS_XS0);
    if (j == 1) return team_ld(slots, TS_XS1);
    if (j == 4) return team_ld(slots, TS_XYC);
    return fp2_zero();
}
// the same with a VARIABLE point (ParametersVar allocated as witnesses): ell computes c1.c0 * p.x and c1.c1 * p.x — two product witnesses —
// before mul_by_014 on the constant f = 1, which is linear. Pair slots as for TEAM_OP_ELLV: XH0 = c0, XH1 = c1, XPX = (p.x, 0), XYV = (p.y, 0).
BLSW_HD Fp2 team_first_f_var(uint32_t j, const Fp2* slots, const Emitter& e) {
    if (j == 0) return team_ld(slots, TS_XH0);
    if (j == 1) {
        const Fp2 c1 = team_ld(slots, TS_XH1);
        const Fp px = team_ld(slots, TS_XPX).c0;
        Fp2 r = {fp_mul(c1.c0, px), fp_mul(c1.c1, px)};
        Emitter w = e;
        w.put(r.c0);
        w.put(r.c1);
        return r;
    }
    if (j == 4) return team_ld(slots, TS_XYV);
    return fp2_zero();
}
// line coefficients of step k into the slot file: lane j < 4 moves the sig pair, lane j - ... see TeamLanes / host
// C: coefficient storage with ld(idx), as chain_prepare_g2 wrote it (4 Fp per step: c0.c0, c0.c1, c1.c0, c1.c1)
template <class C>
BLSW_HD void team_load_coeff_lane(uint32_t j, Fp2* slots, const C& coeff_sig, const C& coeff_h, uint32_t k) {
    // lanes 0,1: sig c0 / c1 (c1 times the constant g1.x: a linear combination in the circuit); lanes 2,3: H(m) c0 / c1
    if (j == 0) team_st(slots, TS_XS0, {coeff_sig.ld(4 * k + 0), coeff_sig.ld(4 * k + 1)});
    if (j == 1) team_st(slots, TS_XS1, fp2_mul_fp({coeff_sig.ld(4 * k + 2), coeff_sig.ld(4 * k + 3)}, K_G1_GEN_X()));
    if (j == 2) team_st(slots, TS_XH0, {coeff_h.ld(4 * k + 0), coeff_h.ld(4 * k + 1)});
    if (j == 3) team_st(slots, TS_XH1, {coeff_h.ld(4 * k + 2), coeff_h.ld(4 * k + 3)});
}
// N+1-pair product: the sig pair's coefficients (lanes 0, 1) and pair j's coefficients + prepared key (lanes 2..5)
template <class C>
BLSW_HD void team_load_coeff_sig_lane(uint32_t j, Fp2* slots, const C& coeff_sig, uint32_t k) {
    if (j == 0) team_st(slots, TS_XS0, {coeff_sig.ld(4 * k + 0), coeff_sig.ld(4 * k + 1)});
    if (j == 1) team_st(slots, TS_XS1, fp2_mul_fp({coeff_sig.ld(4 * k + 2), coeff_sig.ld(4 * k + 3)}, K_G1_GEN_X()));
}
template <class C>
BLSW_HD void team_load_pair_lane(uint32_t j, Fp2* slots, const C& coeff_h, uint32_t k, const Fp& pkx, const Fp& pky) {
    if (j == 2) team_st(slots, TS_XH0, {coeff_h.ld(4 * k + 0), coeff_h.ld(4 * k + 1)});
    if (j == 3) team_st(slots, TS_XH1, {coeff_h.ld(4 * k + 2), coeff_h.ld(4 * k + 3)});
    if (j == 4) team_st(slots, TS_XYV, {pky, fp_zero()});
    if (j == 5) team_st(slots, TS_XPX, {pkx, fp_zero()});
}
BLSW_HD void team_set_consts_lane(uint32_t j, Fp2* slots, const Fp& pkx, const Fp& pky) {
    if (j == 0) team_st(slots, TS_XYC, {K_G1_GEN_NEG_Y(), fp_zero()});
    if (j == 1) team_st(slots, TS_XYV, {pky, fp_zero()});
    if (j == 2) team_st(slots, TS_XPX, {pkx, fp_zero()});
}

// ------------------------------------------------------------------------------------------------ the program
// TEAM interface: Reg; exec(op, a, b) -> Reg (advances the witness cursor); conj / frob; load_coeffs(k); first_f();
// inverse_w(a); is_one_w(a, e_one); set_cursor(e)
// The two loops that execute 97 % of the ops have ONE inlined call site of the op executor each (exec_hot): a call
// would save and restore the callee-saved registers holding the distributed values on every op (measured: 12 GB of
// scratch write-back per 16 384 instances). The straight-line remainder of the final exponentiation uses calls (exec).
template <class TEAM>
BLSW_HD typename TEAM::Reg team_miller(TEAM& t) {
    typename TEAM::Reg f = t.zero();
    uint32_t k = 0;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        // phases of one bit: 0 square, 1 ell(-g1, sig), 2 ell(pk, H), then 3, 4 = the two ell again (addition step) on set bits
        const int n_phases = ((BLSW_X_ABS >> i) & 1) ? 5 : 3;
#pragma unroll 1
        for (int ph = (i == 62 ? 1 : 0); ph < n_phases; ph++) {
            if (ph == 1 || ph == 3) t.load_coeffs(k);
            if (i == 62 && ph == 1) {
                f = t.first_f();  // f = 1 is a constant: the first ell is a linear combination
                continue;
            }
            const TeamOp& T = ph == 0 ? TEAM_OP_SQR : ((ph & 1) ? TEAM_OP_ELLC : TEAM_OP_ELLV);
            f = t.exec_hot(T, f, f);
            if (ph == 2 || ph == 4) k++;
        }
    }
    return t.conj(f);
}
// miller_loop of the single-key circuit with ParametersVar allocated as witnesses: both pairs have a variable point, every ell is TEAM_OP_ELLV.
// TEAM provides load_pair_sig(k) / load_pair_h(k) (pair slots XH0, XH1, XYV, XPX) and first_f_var() (2 witnesses, cursor advanced).
template <class TEAM>
BLSW_HD typename TEAM::Reg team_miller_pv(TEAM& t) {
    typename TEAM::Reg f = t.zero();
    uint32_t k = 0;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        const int n_phases = ((BLSW_X_ABS >> i) & 1) ? 5 : 3;
#pragma unroll 1
        for (int ph = (i == 62 ? 1 : 0); ph < n_phases; ph++) {
            if (ph == 1 || ph == 3) t.load_pair_sig(k);
            if (ph == 2 || ph == 4) t.load_pair_h(k);
            if (i == 62 && ph == 1) {
                f = t.first_f_var();
                continue;
            }
            f = t.exec_hot(ph == 0 ? TEAM_OP_SQR : TEAM_OP_ELLV, f, f);
            if (ph == 2 || ph == 4) k++;
        }
    }
    return t.conj(f);
}
// miller_loop over K + 1 pairs: (-g1 constant, sig) and (pk_j, H(m_j)), j = 0..K-1 — PairingVar::miller_loop on slices
// (constraints.rs:121-125). Per bit: f^2, then one ell per pair in slice order; the addition step repeats the ells.
// TEAM additionally provides load_coeff_sig(k) and load_pair(j, k) (line coefficients of H(m_j) at step k and pk_j).
template <class TEAM>
BLSW_HD typename TEAM::Reg team_miller_multi(TEAM& t, uint32_t K) {
    typename TEAM::Reg f = t.zero();
    uint32_t k = 0;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        const int reps = ((BLSW_X_ABS >> i) & 1) ? 2 : 1;
#pragma unroll 1
        for (int rep = 0; rep < reps; rep++) {
            // phases of one line-coefficient index k: 0 square (doubling step only), 1 ell(-g1, sig), 2 + j ell(pk_j, H_j)
#pragma unroll 1
            for (uint32_t ph = (rep == 0 && i != 62) ? 0u : 1u; ph < 2 + K; ph++) {
                if (ph == 1) t.load_coeff_sig(k);
                if (ph >= 2) t.load_pair(ph - 2, k);
                if (i == 62 && rep == 0 && ph == 1) {
                    f = t.first_f();  // f = 1 is a constant: the first ell is a linear combination
                    continue;
                }
                const TeamOp& T = ph == 0 ? TEAM_OP_SQR : (ph == 1 ? TEAM_OP_ELLC : TEAM_OP_ELLV);
                f = t.exec_hot(T, f, f);
            }
            k++;
        }
    }
    return t.conj(f);
}
template <class TEAM>
BLSW_HD typename TEAM::Reg team_exp_by_x_body(TEAM& t, const typename TEAM::Reg& f) {
    const uint64_t plus = (1ull << 16) | (1ull << 48) | (1ull << 57) | (1ull << 60);
    const uint64_t minus = (1ull << 62);
    typename TEAM::Reg res = f;
#pragma unroll 1
    for (int i = 63; i >= 0; i--) {
        const bool p = (plus >> i) & 1, m = (minus >> i) & 1;
#pragma unroll 1
        for (int ph = 0; ph < ((p || m) ? 2 : 1); ph++) {
            const TeamOp& T = ph == 0 ? TEAM_OP_CYC : TEAM_OP_MUL;
            res = t.exec_hot(T, res, ph == 0 ? res : (m ? t.conj(f) : f));
        }
    }
    return t.conj(res);
}
template <class TEAM>
BLSW_HD typename TEAM::Reg team_exp_by_x(TEAM& t, const typename TEAM::Reg& f) {
    return t.exp_by_x(f);
}
// [k] ge for the bits of `words` (big-endian, the top bit is ge itself): the in-circuit subgroup check of G2 allocation
// (curve.hpp::proj_mul_bits_be_w<OpsFp2>). Points are distributed over lanes 0..2 of a team (x, y, z).
template <class TEAM>
BLSW_HD typename TEAM::Reg team_g2_mul_bits(TEAM& t, const typename TEAM::Reg& ge, const uint32_t* words, int nbits) {
    typename TEAM::Reg result = ge;
#pragma unroll 1
    for (int i = nbits - 2; i >= 0; i--) {
        const int n_phases = ((words[i >> 5] >> (i & 31)) & 1) ? 2 : 1;
#pragma unroll 1
        for (int ph = 0; ph < n_phases; ph++) result = t.exec_hot(ph == 0 ? TEAM_OP_G2DBL : TEAM_OP_G2ADD, result, ge);
    }
    return result;
}
// final_exponentiation . is_one (chains.hpp::chain_final_exp_is_one); the cursor of `t` must be at off_final_exp
template <class TEAM>
BLSW_HD bool team_final_exp_is_one(TEAM& t, const typename TEAM::Reg& f, const Emitter& e_one) {
    typedef typename TEAM::Reg R;
    R f1 = t.conj(f);
    R f2 = t.inverse_w(f);
    R r = t.exec(TEAM_OP_MUL, f1, f2);
    f2 = r;
    r = t.frob(r, 2);
    r = t.exec(TEAM_OP_MUL, r, f2);
    R y0 = t.conj(t.exec(TEAM_OP_CYC, r, r));
    R y5 = team_exp_by_x(t, r);
    R y1 = t.exec(TEAM_OP_CYC, y5, y5);
    R y3 = t.exec(TEAM_OP_MUL, y0, y5);
    y0 = team_exp_by_x(t, y3);
    R y2 = team_exp_by_x(t, y0);
    R y4 = team_exp_by_x(t, y2);
    y4 = t.exec(TEAM_OP_MUL, y4, y1);
    y1 = team_exp_by_x(t, y4);
    y3 = t.conj(y3);
    y1 = t.exec(TEAM_OP_MUL, y1, y3);
    y1 = t.exec(TEAM_OP_MUL, y1, r);
    y3 = t.conj(r);
    y0 = t.exec(TEAM_OP_MUL, y0, r);
    y0 = t.frob(y0, 3);
    y4 = t.exec(TEAM_OP_MUL, y4, y3);
    y4 = t.frob(y4, 1);
    y5 = t.exec(TEAM_OP_MUL, y5, y2);
    y5 = t.frob(y5, 2);
    y5 = t.exec(TEAM_OP_MUL, y5, y0);
    y5 = t.exec(TEAM_OP_MUL, y5, y4);
    y5 = t.exec(TEAM_OP_MUL, y5, y1);
    return t.is_one_w(y5, e_one);
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------ device team
// One lane per thread; the workgroup is a single wave, so the barrier is an LDS fence between phases.
#define BLSW_TEAM_DEV __device__ __forceinline__
BLSW_TEAM_DEV void team_sync() { __syncthreads(); }

// descriptors are fetched into registers one round ahead (wide loads from constant memory, waited on only when used)
BLSW_TEAM_DEV TeamLin team_fetch(const TeamLin* p) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
    TeamLin r;
    r.n = w[0];
    r.idx[0] = w[1]; r.idx[1] = w[2]; r.idx[2] = w[3]; r.idx[3] = w[4];
    return r;
}
BLSW_TEAM_DEV TeamTask team_fetch(const TeamTask* p) {
    TeamTask r;
    r.hdr = p->hdr;
    r.a = team_fetch(&p->a);
    r.b = team_fetch(&p->b);
    return r;
}
BLSW_TEAM_DEV Fp2 team_exec_lane_inl(const TeamOp& T, Fp2* slots, uint32_t j, bool active, const Fp2& in0, const Fp2& in1, Emitter& e) {
    TeamTask cur = team_fetch(&T.task[0][j]);
    const TeamLin outl = team_fetch(&T.out[j]);
    if (active) {
        team_st(slots, TS_IN0 + j, in0);
        team_st(slots, TS_IN1 + j, in1);
    }
    team_sync();
    const uint32_t rounds = T.rounds;
#pragma unroll 1
    for (uint32_t r = 0; r < rounds; r++) {
        TeamTask nxt = team_fetch(&T.task[r + 1 < rounds ? r + 1 : r][j]);
        if (active) team_task(cur, slots, e);
        team_sync();
        cur = nxt;
    }
    Fp2 out = fp2_zero();
    if (active) out = team_gather(outl, slots);
    team_sync();  // the next op republishes into the same slots
    e.pos += T.n_witness;
    return out;
}

inline __device__ __noinline__ Fp2 team_exec_lane(const TeamOp& T, Fp2* slots, uint32_t j, bool active, const Fp2& in0, const Fp2& in1, Emitter& e) {
    return team_exec_lane_inl(T, slots, j, active, in0, in1, e);
}

template <class C>
struct TeamLanes {
    typedef Fp2 Reg;
    Fp2* slots;
    uint32_t j;
    bool active;
    Emitter e;
    C coeff_sig, coeff_h;
    BLSW_TEAM_DEV Reg zero() const { return fp2_zero(); }
    BLSW_TEAM_DEV Reg exec(const TeamOp& T, const Reg& a, const Reg& b) { return team_exec_lane(T, slots, j, active, a, b, e); }
    BLSW_TEAM_DEV Reg exec_hot(const TeamOp& T, const Reg& a, const Reg& b) { return team_exec_lane_inl(T, slots, j, active, a, b, e); }
    __device__ __noinline__ Reg exp_by_x(const Reg& f) { return team_exp_by_x_body(*this, f); }
    BLSW_TEAM_DEV Reg conj(const Reg& a) const { return team_conj(j, a); }
    BLSW_TEAM_DEV Reg frob(const Reg& a, int power) const { return team_frob(j, a, power); }
    BLSW_TEAM_DEV void set_consts(const Fp& pkx, const Fp& pky) {
        if (active) team_set_consts_lane(j, slots, pkx, pky);
        team_sync();
    }
    BLSW_TEAM_DEV void load_coeffs(uint32_t k) {
        if (active) team_load_coeff_lane(j, slots, coeff_sig, coeff_h, k);
        team_sync();
    }
    BLSW_TEAM_DEV Reg first_f() const { return active ? team_first_f(j, slots) : fp2_zero(); }
    BLSW_TEAM_DEV Reg inverse_w(const Reg& a) {
        if (active) team_st(slots, TS_IN0 + j, a);
        team_sync();
        if (active && j == 0) team_inverse_lane0(slots);
        team_sync();
        Reg inv = fp2_zero();
        if (active) {
            inv = team_ld(slots, TS_IN1 + j);
            Emitter w = e;
            w.pos += 2 * j;
            w.put(inv.c0);
            w.put(inv.c1);
        }
        e.pos += 12;
        team_sync();
        exec(TEAM_OP_INVCHK, a, inv);
        return inv;
    }
    BLSW_TEAM_DEV bool is_one_w(const Reg& a, const Emitter& e_one) {
        bool mine = active ? team_is_one_coeff(j, a, e_one) : false;
        if (active) {
            Fp2 flag = fp2_zero();
            flag.c0.l[0] = mine ? 1u : 0u;
            team_st(slots, TS_P + j, flag);
        }
        team_sync();
        bool res = false;
        if (active) {
            bool b[6];
#pragma unroll
            for (int q = 0; q < 6; q++) b[q] = team_ld(slots, TS_P + q).c0.l[0] != 0;
            res = team_is_one_tree(j, b, e_one);
        }
        team_sync();
        return res;
    }
};
#endif

}  // namespace blsw
