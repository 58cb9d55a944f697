// Device team with per-pair loads for the N+1-pair product (k_team.hip, k_miller_par.hip).
#pragma once
#include "kcommon.hpp"
#include "team.hpp"
#include "vpairing.hpp"

namespace blsw {

struct TeamLanesMulti : TeamLanes<CoeffStrided> {
    const Fp* coeff_h_all;
    const Fp* pkaff;
    uint64_t n_h, flat0;
    BLSW_TEAM_DEV void load_coeff_sig(uint32_t k) {
        if (active) team_load_coeff_sig_lane(j, slots, coeff_sig, k);
        team_sync();
    }
    BLSW_TEAM_DEV void load_pair(uint32_t jp, uint32_t k) {
        if (active) {
            const uint64_t t = flat0 + jp;
            Fp px = fp_zero(), py = fp_zero();
            if (j == 5) px = ld_fp(pkaff + t);
            if (j == 4) py = ld_fp(pkaff + n_h + t);
            team_load_pair_lane(j, slots, CoeffStrided{const_cast<Fp*>(coeff_h_all) + t, n_h}, k, px, py);
        }
        team_sync();
    }
};

// the native pairing of blsw_verify_batch (vpairing.hpp): coeff_sig / coeff_h hold the 68 projective line triples of the two G2 points
struct TeamLanesValues : TeamLanes<CoeffStrided> {
    BLSW_TEAM_DEV Reg one() const { return j == 0 ? fp2_one() : fp2_zero(); }
    BLSW_TEAM_DEV void load_lines(uint32_t k) {
        if (active) team_load_lines_lane(j, slots, coeff_sig, coeff_h, k);
        team_sync();
    }
};

// single-key circuit with ParametersVar allocated as witnesses (team_miller_pv): both pairs go through the pair slots
struct TeamLanesPv : TeamLanes<CoeffStrided> {
    Fp pkx, pky;  // lane 5 / lane 4 hold prepare_g1(pk)
    BLSW_TEAM_DEV void load_pair_sig(uint32_t k) {
        // prepare_g1(-g1) of the allocated generator: the to_affine of k_g1's params lanes yields these canonical values
        if (active) team_load_pair_lane(j, slots, coeff_sig, k, K_G1_GEN_X(), K_G1_GEN_NEG_Y());
        team_sync();
    }
    BLSW_TEAM_DEV void load_pair_h(uint32_t k) {
        if (active) team_load_pair_lane(j, slots, coeff_h, k, pkx, pky);
        team_sync();
    }
    BLSW_TEAM_DEV Reg first_f_var() {
        Reg r = active ? team_first_f_var(j, slots, e) : fp2_zero();
        e.pos += 2;
        return r;
    }
};

}  // namespace blsw
