// Device team with per-pair loads for the N+1-pair product (k_team.hip, k_miller_par.hip).
#pragma once
#include "kcommon.cuh"
#include "team.cuh"

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

}  // namespace blsw
