// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
// N+1-pair product (blsw_verify_multi_batch) with the pairs in parallel: the four phases of miller_par.hpp as kernels, and the
// final exponentiation + is_one of the six-lane team program started from the stored Miller value.
#include "kcommon.hpp"
#include "miller_par.hpp"
#include "team_multi.hpp"

namespace blsw {

// pairs of instance I: prepared keys and line coefficients live at flat index I * K + j of the per-pair launch (n_h lanes)
struct PairsDev {
    const Fp* coeff_h_all;
    const Fp* pkaff;
    uint64_t n_h, flat0;
    __device__ __forceinline__ void pk(uint32_t j, Fp& x, Fp& y) const {
        x = ld_fp(pkaff + flat0 + j);
        y = ld_fp(pkaff + n_h + flat0 + j);
    }
    __device__ __forceinline__ CoeffStrided coeff_h(uint32_t j) const { return CoeffStrided{const_cast<Fp*>(coeff_h_all) + flat0 + j, n_h}; }
};
__device__ __forceinline__ PairsDev pairs_of(const Group& gs, const MillerParArgs& a, uint64_t inst) { return PairsDev{gs.ws.coeff_h, gs.ws.pkaff, a.n_h, inst * a.K}; }

// task t = (instance * 68 + k) * C + c
__global__ __launch_bounds__(64) void k_miller_m1(Group gs, MillerParArgs a) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, per_inst = (uint64_t)BLSW_MILLER_STEPS * a.C;
    if (t >= gs.N * per_inst) return;
    const uint64_t inst = t / per_inst;
    const uint32_t r = (uint32_t)(t - inst * per_inst), k = r / a.C, c = r - k * a.C;
    const Fp12Rows Cp = {a.cprod, gs.N * per_inst};
    Cp.st(t, miller_m1(pairs_of(gs, a, inst), a.K, a.B, k, c));
}
// task t = instance * 68 + k
__global__ __launch_bounds__(64) void k_miller_m1b(Group gs, MillerParArgs a) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= gs.N * BLSW_MILLER_STEPS) return;
    const uint64_t items = gs.N * BLSW_MILLER_STEPS * a.C;
    miller_m1b(Fp12Rows{a.cprod, items}, Fp12Rows{a.q, items}, Fp12Rows{a.t, gs.N * BLSW_MILLER_STEPS}, t * a.C, t, a.C);
}
// one lane per instance: squares, ell(sig) and the running product; leaves conj(f) for the final exponentiation
__global__ __launch_bounds__(64) void k_miller_m2(Group gs, MillerParArgs a) {
    const uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= gs.N) return;
    LaneId id = lane_id(gs, I);
    const uint64_t steps = gs.N * BLSW_MILLER_STEPS;
    const Fp12 f = miller_m2(EMIT(gs, id, off_miller), a.K, CoeffStrided{gs.ws.coeff_sig + I, gs.ws.n_sig}, Fp12Rows{a.t, steps}, Fp12Rows{a.f1, steps}, I * BLSW_MILLER_STEPS);
    Fp12Rows{a.ffinal, gs.N}.st(I, f);
}
__global__ __launch_bounds__(64) void k_miller_m3(Group gs, MillerParArgs a) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, per_inst = (uint64_t)BLSW_MILLER_STEPS * a.C;
    if (t >= gs.N * per_inst) return;
    const uint64_t inst = t / per_inst;
    const uint32_t r = (uint32_t)(t - inst * per_inst), k = r / a.C, c = r - k * a.C;
    LaneId id = lane_id(gs, inst);
    const Fp12 f1 = Fp12Rows{a.f1, gs.N * BLSW_MILLER_STEPS}.ld(inst * BLSW_MILLER_STEPS + k);
    miller_m3(EMIT(gs, id, off_miller), pairs_of(gs, a, inst), a.K, a.B, k, c, f1, Fp12Rows{a.q, gs.N * per_inst}, t);
}
// the serial spine on the six-lane team program (the single-lane k_miller_m2 takes 23 ms for its 68 steps: 62 squares, 67 ell(sig), 68
// full products; a team does the same in a fraction): squares and ell(sig) with their witnesses, f1 stored, the product with T[k]
// as a value (no cursor). Leaves conj(f) for k_final_team.
__global__ __launch_bounds__(64) void k_miller_spine_team(Group gs, MillerParArgs a) {
    __shared__ Fp2 lds[BLSW_TEAMS_PER_WAVE * TS_NSLOTS];
    constexpr MillerStepInfo info = miller_step_info();
    const uint32_t team = threadIdx.x / 6, j = threadIdx.x % 6;
    const uint64_t I0 = (uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE + team;
    const bool active = team < BLSW_TEAMS_PER_WAVE && I0 < gs.N;
    const uint64_t I = active ? I0 : 0, steps = gs.N * BLSW_MILLER_STEPS;
    LaneId id = lane_id(gs, I);
    TeamLanesMulti t;
    t.slots = lds + (active ? team : 0) * TS_NSLOTS;
    t.j = j;
    t.active = active;
    t.coeff_h = {nullptr, 0};
    t.coeff_sig = {gs.ws.coeff_sig + I, gs.ws.n_sig};
    t.coeff_h_all = nullptr;
    t.pkaff = nullptr;
    t.n_h = 0;
    t.flat0 = 0;
    t.e = EMIT(gs, id, off_miller);
    if (!active) t.e.base = nullptr;
    uint32_t* const base = t.e.base;
    const uint32_t pos0 = t.e.pos;
    if (active && j == 0) team_st(t.slots, TS_XYC, {K_G1_GEN_NEG_Y(), fp_zero()});
    team_sync();
    Fp2 f = fp2_zero();
#pragma unroll 1
    for (uint32_t k = 0; k < BLSW_MILLER_STEPS; k++) {
        const uint64_t item = I * BLSW_MILLER_STEPS + k;
        t.e.pos = pos0 + miller_step_pos(info, k, a.K);
#pragma unroll 1
        for (int ph = info.dbl[k] ? 0 : 1; ph < 3; ph++) {
            Fp2 other = f;
            if (ph == 1) {
                t.load_coeff_sig(k);
                if (k == 0) {
                    f = t.first_f();  // f = 1 is a constant: the first ell is a linear combination
                    continue;
                }
            }
            if (ph == 2) {  // f1 = value after ell(sig): what the chunks of this step start from; then the step's pairs as one value product
                if (active) {
                    st_fp(a.f1 + (uint64_t)(2 * j) * steps + item, f.c0);
                    st_fp(a.f1 + (uint64_t)(2 * j + 1) * steps + item, f.c1);
                }
                other = {ld_fp(a.t + (uint64_t)(2 * j) * steps + item), ld_fp(a.t + (uint64_t)(2 * j + 1) * steps + item)};
                t.e.base = nullptr;
            }
            const TeamOp& T = ph == 0 ? TEAM_OP_SQR : (ph == 1 ? TEAM_OP_ELLC : TEAM_OP_MUL);
            f = t.exec_hot(T, f, other);
            if (ph == 2) t.e.base = base;
        }
    }
    f = t.conj(f);
    if (active) {
        st_fp(a.ffinal + (uint64_t)(2 * j) * gs.N + I, f.c0);
        st_fp(a.ffinal + (uint64_t)(2 * j + 1) * gs.N + I, f.c1);
    }
}
// final exponentiation + is_one of the stored Miller values, six lanes per instance (team.hpp)
__global__ __launch_bounds__(64) void k_final_team(Group gs, MillerParArgs a) {
    __shared__ Fp2 lds[BLSW_TEAMS_PER_WAVE * TS_NSLOTS];
    const uint32_t team = threadIdx.x / 6, j = threadIdx.x % 6;
    const uint64_t I0 = (uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE + team;
    const bool active = team < BLSW_TEAMS_PER_WAVE && I0 < gs.N;
    const uint64_t I = active ? I0 : 0;
    LaneId id = lane_id(gs, I);
    TeamLanes<CoeffStrided> t;
    t.slots = lds + (active ? team : 0) * TS_NSLOTS;
    t.j = j;
    t.active = active;
    t.coeff_h = {nullptr, 0};
    t.coeff_sig = {nullptr, 0};
    t.e = EMIT(gs, id, off_final_exp);
    if (!active) t.e.base = nullptr;
    // lane j owns the Fp2 coefficient j of the value: (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2)
    const Fp2 f = {ld_fp(a.ffinal + (uint64_t)(2 * j) * gs.N + I), ld_fp(a.ffinal + (uint64_t)(2 * j + 1) * gs.N + I)};
    Emitter e_one = EMIT(gs, id, off_is_one);
    if (!active) e_one.base = nullptr;
    bool res = team_final_exp_is_one(t, f, e_one);
    int32_t* r = gs.desc[id.s].result;
    if (active && j == 0 && r) r[id.i] = step_result(gs.desc[id.s], id.i, res);
}

void launch_miller_par(const Group& gs, const MillerParArgs& a, hipStream_t st, hipStream_t side, hipEvent_t ev_spine, hipEvent_t ev_side) {
    const uint64_t tasks = gs.N * BLSW_MILLER_STEPS * a.C, steps = gs.N * BLSW_MILLER_STEPS;
    hipLaunchKernelGGL(k_miller_m1, dim3((unsigned)((tasks + 63) / 64)), dim3(64), 0, st, gs, a);
    hipLaunchKernelGGL(k_miller_m1b, dim3((unsigned)((steps + 63) / 64)), dim3(64), 0, st, gs, a);
    if (a.spine_lane)
        hipLaunchKernelGGL(k_miller_m2, dim3((unsigned)((gs.N + 63) / 64)), dim3(64), 0, st, gs, a);
    else
        hipLaunchKernelGGL(k_miller_spine_team, dim3((unsigned)((gs.N + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, st, gs, a);
    // the final exponentiation needs only the spine's value: it runs beside the chunks' witness pass
    hipStream_t fe = side ? side : st;
    if (side) {
        hipEventRecord(ev_spine, st);
        hipStreamWaitEvent(side, ev_spine, 0);
    }
    hipLaunchKernelGGL(k_final_team, dim3((unsigned)((gs.N + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, fe, gs, a);
    hipLaunchKernelGGL(k_miller_m3, dim3((unsigned)((tasks + 63) / 64)), dim3(64), 0, st, gs, a);
    if (side) {
        hipEventRecord(ev_side, side);
        hipStreamWaitEvent(st, ev_side, 0);
    }
}

}  // namespace blsw
