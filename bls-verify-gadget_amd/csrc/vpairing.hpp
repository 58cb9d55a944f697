// The native pairing check of BLS::verify (bls.rs:427-458: Bls12::multi_pairing([-g1, pk], [sig, H(m)]).is_one()) as VALUES: no circuit, no witnesses.
//
// The in-circuit Miller loop walks the G2 points in AFFINE coordinates — an inversion per step, whose quotient is a witness.
// A verdict needs no witnesses: the points are walked in homogeneous projective coordinates (dbl-2007-bl / madd-1998-cmo, a = 0) and a step's line
// through P = (x_P, y_P) is kept up to a factor in Fp2, which the final exponentiation kills ((p^2 - 1) divides (p^12 - 1) / r):
//   doubling of R = (X, Y, Z):   w = 3 X^2, s = 2 Y Z:   line = (w X - s Y) + (-w Z) x_P [v] + (s Z) y_P [v w]
//   addition R + Q, Q affine:    u = y_Q Z - Y, v = x_Q Z - X:   line = (u x_Q - v y_Q) + (-u) x_P [v] + v y_P [v w]
// i.e. f <- f.mul_by_014(c0, c1 x_P, c2 y_P) with three general Fp2 coefficients (team_tables.hpp: TEAM_OP_ELLGS / ELLGH), the circuit's ell being
// the case c2 = 1 (its affine slopes). Phase 1 (one lane per G2 point, vline_chain) leaves the 68 coefficient triples, already multiplied by the
// pair's G1 point; phase 2 (six lanes per instance, team_miller_values + team.hpp's final exponentiation with a null cursor) folds them.
// Compiles for the host as well: tests/hostsim runs both phases against the oracle's native verify.
#pragma once
#include "team.hpp"
#include "vcurve.hpp"

namespace blsw {

#ifndef BLSW_VLINE_ROWS
#define BLSW_VLINE_ROWS (6u * 68u)  // per G2 point: 68 steps x (c0, c1 x_P, c2 y_P), two Fp each
#endif

// the 68 line-coefficient triples of G2 point Q (affine, not the identity) against the G1 point P = (px, py): out.st(6 k + j, .)
template <class C>
BLSW_FN void vline_chain(const Fp2& qx, const Fp2& qy, const Fp& px, const Fp& py, const C& out) {
    Fp2 X = qx, Y = qy, Z = fp2_one();
    uint32_t k = 0;
    auto put = [&](const Fp2& c0, const Fp2& c1, const Fp2& c2) {
        const Fp2 c1p = fp2_mul_fp(c1, px), c2p = fp2_mul_fp(c2, py);
        out.st(6 * k + 0, c0.c0);
        out.st(6 * k + 1, c0.c1);
        out.st(6 * k + 2, c1p.c0);
        out.st(6 * k + 3, c1p.c1);
        out.st(6 * k + 4, c2p.c0);
        out.st(6 * k + 5, c2p.c1);
        k++;
    };
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        {  // dbl-2007-bl, a = 0
            const Fp2 xx = v_sqr(X);
            const Fp2 w = fp2_add(fp2_dbl(xx), xx);
            const Fp2 s = fp2_dbl(fp2_mul_inl(Y, Z));
            const Fp2 ss = v_sqr(s), sss = fp2_mul_inl(s, ss);
            const Fp2 r = fp2_mul_inl(Y, s), rr = v_sqr(r);
            const Fp2 b = fp2_sub(fp2_sub(v_sqr(fp2_add(X, r)), xx), rr);
            const Fp2 h = fp2_sub(v_sqr(w), fp2_dbl(b));
            put(fp2_sub(fp2_mul_inl(w, X), r), fp2_neg(fp2_mul_inl(w, Z)), fp2_mul_inl(s, Z));  // s Y = r
            X = fp2_mul_inl(h, s);
            Y = fp2_sub(fp2_mul_inl(w, fp2_sub(b, h)), fp2_dbl(rr));
            Z = sss;
        }
        if ((BLSW_X_ABS >> i) & 1) {  // madd-1998-cmo
            const Fp2 u = fp2_sub(fp2_mul_inl(qy, Z), Y), v = fp2_sub(fp2_mul_inl(qx, Z), X);
            put(fp2_sub(fp2_mul_inl(u, qx), fp2_mul_inl(v, qy)), fp2_neg(u), v);
            const Fp2 uu = v_sqr(u), vv = v_sqr(v), vvv = fp2_mul_inl(v, vv);
            const Fp2 rr = fp2_mul_inl(vv, X);
            const Fp2 a = fp2_sub(fp2_sub(fp2_mul_inl(uu, Z), vvv), fp2_dbl(rr));
            X = fp2_mul_inl(v, a);
            Y = fp2_sub(fp2_mul_inl(u, fp2_sub(rr, a)), fp2_mul_inl(vvv, Y));
            Z = fp2_mul_inl(vvv, Z);
        }
    }
}

// lane j of a team loads its part of step k's two triples: lanes 0..2 the (-g1, sig) pair -> XS0, XS1, XYC; lanes 3..5 the (pk, H) pair -> XH0, XH1, XYV
template <class C>
BLSW_HD void team_load_lines_lane(uint32_t j, Fp2* slots, const C& lines_sig, const C& lines_h, uint32_t k) {
    const uint32_t part = j % 3;
    const Fp2 v = j < 3 ? Fp2{lines_sig.ld(6 * k + 2 * part), lines_sig.ld(6 * k + 2 * part + 1)} : Fp2{lines_h.ld(6 * k + 2 * part), lines_h.ld(6 * k + 2 * part + 1)};
    const uint32_t slot = j < 3 ? (part == 0 ? TS_XS0 : (part == 1 ? TS_XS1 : TS_XYC)) : (part == 0 ? TS_XH0 : (part == 1 ? TS_XH1 : TS_XYV));
    team_st(slots, slot, v);
}
// Miller loop over the two pairs, values only. TEAM additionally provides one() and load_lines(k)
template <class TEAM>
BLSW_HD typename TEAM::Reg team_miller_values(TEAM& t) {
    typename TEAM::Reg f = t.one();
    uint32_t k = 0;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        const int n_phases = ((BLSW_X_ABS >> i) & 1) ? 5 : 3;
#pragma unroll 1
        for (int ph = (i == 62 ? 1 : 0); ph < n_phases; ph++) {
            if (ph == 1 || ph == 3) t.load_lines(k++);
            f = t.exec_hot(ph == 0 ? TEAM_OP_SQR : ((ph & 1) ? TEAM_OP_ELLGS : TEAM_OP_ELLGH), f, f);
        }
    }
    return t.conj(f);
}

}  // namespace blsw
