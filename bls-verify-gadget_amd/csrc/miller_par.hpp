// Miller product over K + 1 pairs with the pairs in parallel (blsw_verify_multi_batch; constraints.rs:121-125 on slices of K + 1).
//
// In the circuit the product is one chain: per line-coefficient step k (68 of them) f <- f^2 (doubling steps), f <- ell(f, sig), then
// f <- ell(f, pair j) for j = 0..K-1, and every intermediate f is a witness — 68 (K + 1) dependent sparse products. The VALUES are
// a prefix product, so the chain is cut into chunks of B pairs that run on different lanes:
//   m1  (lane per step k, chunk c)   C[k][c] = product of the chunk's B sparse line elements        (value only)
//   m1b (lane per step k)            Q[k][c] = C[k][0] ... C[k][c-1],  T[k] = product of all chunks  (value only, full Fp12 products)
//   m2  (lane per instance)          the serial spine: F'_k = f_k^2 (36 witnesses), f1_k = ell(F'_k, sig) (30 witnesses),
//                                    f_{k+1} = f1_k T[k]; stores f1_k                                 (68 steps instead of 68 (K + 1))
//   m3  (lane per step k, chunk c)   f = f1_k Q[k][c], then the chunk's B witness-emitting ell(f, pair j) at their places
// Field products are commutative and every value is a canonical residue, so the witnesses are bit for bit those of the serial
// chain (chains.hpp: chain_miller_multi; team.hpp: team_miller_multi), which stay as the statement of the segment.
// Compiles for the host as well: tests/hostsim runs the four phases as loops against the oracle.
#pragma once
#include "chains.hpp"

namespace blsw {

#define BLSW_MILLER_STEPS 68
#ifndef BLSW_MILLER_CHUNK
#define BLSW_MILLER_CHUNK 12  // pairs per lane in m1 / m3 (B below; the host harness also runs small chunks)
#endif

// step k of the line-coefficient sequence: does it square first, and where do its witnesses start (without the per-pair part)
struct MillerStepInfo {
    uint8_t dbl[BLSW_MILLER_STEPS];
    uint16_t base[BLSW_MILLER_STEPS + 1];  // witnesses in front of step k that do not belong to a variable pair: 36 per square, 30 per ell(sig)
};
constexpr MillerStepInfo miller_step_info() {
    MillerStepInfo m = {};
    uint32_t k = 0, a = 0;
    for (int i = 62; i >= 0; i--) {
        const int reps = ((BLSW_X_ABS >> i) & 1) ? 2 : 1;
        for (int rep = 0; rep < reps; rep++) {
            m.dbl[k] = (rep == 0 && i != 62) ? 1 : 0;
            m.base[k] = (uint16_t)a;
            a += (m.dbl[k] ? 36u : 0u) + (k == 0 ? 0u : 30u);
            k++;
        }
    }
    m.base[k] = (uint16_t)a;
    return m;
}
// first witness of step k in the Miller segment, for K variable pairs
BLSW_HD uint32_t miller_step_pos(const MillerStepInfo& m, uint32_t k, uint32_t K) { return m.base[k] + 38u * K * k; }

#if defined(__HIP_DEVICE_COMPILE__)
#define BLSW_MP_LD(p) ld_fp(p)
#define BLSW_MP_ST(p, v) st_fp(p, v)
#else
#define BLSW_MP_LD(p) (*(p))
#define BLSW_MP_ST(p, v) (*(p) = (v))
#endif
// Fp12 values of many tasks, element-major: element e of item t at p[e * stride + t] (coalesced across the lanes of a wave)
struct Fp12Rows {
    Fp* p;
    uint64_t stride;
    BLSW_HD Fp12 ld(uint64_t t) const {
        Fp v[12];
        for (int e = 0; e < 12; e++) v[e] = BLSW_MP_LD(p + (uint64_t)e * stride + t);
        return {{{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}}, {{v[6], v[7]}, {v[8], v[9]}, {v[10], v[11]}}};
    }
    BLSW_HD void st(uint64_t t, const Fp12& f) const {
        const Fp* v[12] = {&f.c0.c0.c0, &f.c0.c0.c1, &f.c0.c1.c0, &f.c0.c1.c1, &f.c0.c2.c0, &f.c0.c2.c1, &f.c1.c0.c0, &f.c1.c0.c1, &f.c1.c1.c0, &f.c1.c1.c1, &f.c1.c2.c0, &f.c1.c2.c1};
        for (int e = 0; e < 12; e++) BLSW_MP_ST(p + (uint64_t)e * stride + t, *v[e]);
    }
};
// value-only full product (the witness cursor is a dummy: no stores)
BLSW_FN Fp12 fp12_mul_value(const Fp12& a, const Fp12& b) {
    Emitter dummy = {nullptr, 0};
    return fp12_mul_w(dummy, a, b);
}
BLSW_HD uint32_t miller_chunks(uint32_t K, uint32_t B) { return (K + B - 1) / B; }

// P: pairs of one instance: pk(j, x, y) = prepare_g1(pk_j); coeff_h(j) = line coefficients of prepare_g2(H(m_j)) (ld(idx))
// m1: product of the sparse line elements of chunk c at step k
template <class P>
BLSW_FN Fp12 miller_m1(const P& pairs, uint32_t K, uint32_t B, uint32_t k, uint32_t c) {
    Emitter dummy = {nullptr, 0};
    Fp12 f = fp12_one();
    const uint32_t j1 = (c + 1) * B < K ? (c + 1) * B : K;
#pragma unroll 1
    for (uint32_t j = c * B; j < j1; j++) {
        Fp px, py;
        pairs.pk(j, px, py);
        f = ell_var_p_w(dummy, f, pairs.coeff_h(j), k, px, py);
    }
    return f;
}
// m1b: prefixes over the chunk products of step k. Cprod / Q hold item (k * C + c); T holds item k. Q[k][0] is not stored (= 1).
BLSW_FN void miller_m1b(const Fp12Rows& Cprod, const Fp12Rows& Q, const Fp12Rows& T, uint64_t item0, uint64_t t_item, uint32_t C) {
    Fp12 acc = Cprod.ld(item0);
#pragma unroll 1
    for (uint32_t c = 1; c < C; c++) {
        Q.st(item0 + c, acc);
        acc = fp12_mul_value(acc, Cprod.ld(item0 + c));
    }
    T.st(t_item, acc);
}
// m2: the serial spine of one instance. e: cursor at the start of the Miller segment. F1 / T hold item (t0 + k). Returns conj(f).
template <class C>
BLSW_FN Fp12 miller_m2(Emitter e, uint32_t K, const C& coeff_sig, const Fp12Rows& T, const Fp12Rows& F1, uint64_t t0) {
    constexpr MillerStepInfo info = miller_step_info();
    const uint32_t pos0 = e.pos;
    Fp12 f = fp12_one();
#pragma unroll 1
    for (uint32_t k = 0; k < BLSW_MILLER_STEPS; k++) {
        e.pos = pos0 + miller_step_pos(info, k, K);
        if (info.dbl[k]) f = fp12_sqr_w(e, f);
        f = ell_const_p_w(e, f, coeff_sig, k, k == 0);
        F1.st(t0 + k, f);
        f = fp12_mul_value(f, T.ld(t0 + k));
    }
    return fp12_conj(f);
}
// m3: the witnesses of chunk c at step k. e: cursor at the start of the Miller segment.
template <class P>
BLSW_FN void miller_m3(Emitter e, const P& pairs, uint32_t K, uint32_t B, uint32_t k, uint32_t c, const Fp12& f1, const Fp12Rows& Q, uint64_t q_item) {
    constexpr MillerStepInfo info = miller_step_info();
    Fp12 f = c == 0 ? f1 : fp12_mul_value(f1, Q.ld(q_item));
    e.pos += miller_step_pos(info, k, K) + (info.dbl[k] ? 36u : 0u) + (k == 0 ? 0u : 30u) + 38u * c * B;
    const uint32_t j1 = (c + 1) * B < K ? (c + 1) * B : K;
#pragma unroll 1
    for (uint32_t j = c * B; j < j1; j++) {
        Fp px, py;
        pairs.pk(j, px, py);
        f = ell_var_p_w(e, f, pairs.coeff_h(j), k, px, py);
    }
}

}  // namespace blsw
