// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
#include "kcommon.hpp"
#include "team_multi.hpp"

namespace blsw {

// Miller loop + final exponentiation + is_one, SIX LANES PER INSTANCE (team.hpp): ten instances per wave, every Fp12
// value distributed over the team's registers, operands and products exchanged through the team's 3.5 KB slot file in LDS
__global__ __launch_bounds__(64) void k_pairing_team(Group g) {
    __shared__ Fp2 lds[BLSW_TEAMS_PER_WAVE * TS_NSLOTS];
    if ((uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE >= g.N) return;  // a wave without instances (the scratch pre-warm launch)
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t team = threadIdx.x / 6, j = threadIdx.x % 6;
    const uint64_t I0 = (uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE + team;
    const bool active = team < BLSW_TEAMS_PER_WAVE && I0 < g.N;
    const uint64_t I = active ? I0 : 0, N = g.N;  // idle lanes only take part in the barriers
    LaneId id = lane_id(g, I);
    TeamLanes<CoeffStrided> t;
    t.slots = lds + (active ? team : 0) * TS_NSLOTS;
    t.j = j;
    t.active = active;
    t.coeff_h = {g.ws.coeff_h + I, N};
    t.coeff_sig = {g.ws.coeff_sig + I, g.ws.n_sig};
    t.e = EMIT(g, id, off_miller);
    if (!active) t.e.base = nullptr;
    t.set_consts(ld_fp(g.ws.pkaff + I), ld_fp(g.ws.pkaff + N + I));
    Fp2 f = team_miller(t);
    Emitter e_one = EMIT(g, id, off_is_one);
    if (!active) e_one.base = nullptr;
    bool res = team_final_exp_is_one(t, f, e_one);
    int32_t* r = g.desc[id.s].result;
    if (active && j == 0 && r) r[id.i] = step_result(g.desc[id.s], id.i, res);
}
// the same for ParametersVar allocated as witnesses (L.params_mode: constraints.rs:198-211 with AllocationMode::Witness): team_miller_pv
__global__ __launch_bounds__(64) void k_pairing_team_pv(Group g) {
    __shared__ Fp2 lds[BLSW_TEAMS_PER_WAVE * TS_NSLOTS];
    if ((uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE >= g.N) return;
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t team = threadIdx.x / 6, j = threadIdx.x % 6;
    const uint64_t I0 = (uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE + team;
    const bool active = team < BLSW_TEAMS_PER_WAVE && I0 < g.N;
    const uint64_t I = active ? I0 : 0, N = g.N;
    LaneId id = lane_id(g, I);
    TeamLanesPv t;
    t.slots = lds + (active ? team : 0) * TS_NSLOTS;
    t.j = j;
    t.active = active;
    t.coeff_h = {g.ws.coeff_h + I, N};
    t.coeff_sig = {g.ws.coeff_sig + I, g.ws.n_sig};
    t.pkx = ld_fp(g.ws.pkaff + I);
    t.pky = ld_fp(g.ws.pkaff + N + I);
    t.e = EMIT(g, id, off_miller);
    if (!active) t.e.base = nullptr;
    Fp2 f = team_miller_pv(t);
    Emitter e_one = EMIT(g, id, off_is_one);
    if (!active) e_one.base = nullptr;
    bool res = team_final_exp_is_one(t, f, e_one);
    int32_t* r = g.desc[id.s].result;
    if (active && j == 0 && r) r[id.i] = step_result(g.desc[id.s], id.i, res);
}
// G2 allocation, six lanes per instance: the (r - 1) * sig chain of the subgroup check runs on the team machinery (points on
// lanes 0..2), the allocation witnesses and the enforce_equal tail are single-lane work of lane 0
__global__ __launch_bounds__(64) void k_g2_alloc_team(Group g) {
    // the G2 op tables use the operand slots and 12 product slots only: 24 slots = 23 KB per wave, six waves per CU
    constexpr uint32_t G2_SLOTS = TS_P + 12;
    __shared__ Fp2 lds[BLSW_TEAMS_PER_WAVE * G2_SLOTS];
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    constexpr uint32_t RM1[8] = BLSW_RM1_WORDS;
    const uint32_t team = threadIdx.x / 6, j = threadIdx.x % 6;
    const uint64_t I0 = (uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE + team;
    const bool active = team < BLSW_TEAMS_PER_WAVE && I0 < g.N;
    const uint64_t I = active ? I0 : 0;
    LaneId id = lane_id(g, I);
    const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].sig + (uint64_t)id.i * 24);
    Fp2 sx = {ld_fp(p), ld_fp(p + 1)}, sy = {ld_fp(p + 2), ld_fp(p + 3)};
    const bool inf = fp2_is_zero(sx) && fp2_is_zero(sy);
    Proj<OpsFp2> ge = {inf ? fp2_zero() : sx, inf ? fp2_one() : sy, inf ? fp2_zero() : fp2_one()};
    TeamLanes<CoeffStrided> t;
    t.slots = lds + (active ? team : 0) * G2_SLOTS;
    t.j = j;
    t.active = active;
    t.coeff_h = {nullptr, 0};
    t.coeff_sig = {nullptr, 0};
    t.e = EMIT(g, id, off_sig_alloc);
    if (!active) t.e.base = nullptr;
    Fp2 mine = j == 0 ? ge.x : (j == 1 ? ge.y : (j == 2 ? ge.z : fp2_zero()));
    if (j < 3) {  // the six allocation witnesses: x.c0, x.c1, y.c0, y.c1, z.c0, z.c1
        Emitter w = t.e;
        w.pos += 2 * j;
        w.put(mine.c0);
        w.put(mine.c1);
    }
    t.e.pos += 6;
    (void)team_g2_mul_bits(t, mine, RM1, BLSW_RM1_NBITS);
    if (active && j == 0) chain_g2_alloc_tail(t.e, ge);
}
// N+1-pair product (blsw_verify_multi_batch): one team per instance, K pairs per instance. `gs` is the per-signature view
// (N = instances), the per-pair values (prepare_g1(pk_j), line coefficients of H(m_j)) live at flat index I * K + j of the
// per-pair launch of n_h = N * K lanes.
__global__ __launch_bounds__(64) void k_pairing_team_multi(Group gs, uint32_t K, uint64_t n_h) {
    __shared__ Fp2 lds[BLSW_TEAMS_PER_WAVE * TS_NSLOTS];
    const uint32_t team = threadIdx.x / 6, j = threadIdx.x % 6;
    const uint64_t I0 = (uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE + team;
    const bool active = team < BLSW_TEAMS_PER_WAVE && I0 < gs.N;
    const uint64_t I = active ? I0 : 0;
    LaneId id = lane_id(gs, I);
    TeamLanesMulti t;
    t.slots = lds + (active ? team : 0) * TS_NSLOTS;
    t.j = j;
    t.active = active;
    t.coeff_h = {nullptr, 0};
    t.coeff_sig = {gs.ws.coeff_sig + I, gs.ws.n_sig};
    t.coeff_h_all = gs.ws.coeff_h;
    t.pkaff = gs.ws.pkaff;
    t.n_h = n_h;
    t.flat0 = I * K;
    t.e = EMIT(gs, id, off_miller);
    if (!active) t.e.base = nullptr;
    if (active && j == 0) team_st(t.slots, TS_XYC, {K_G1_GEN_NEG_Y(), fp_zero()});
    team_sync();
    Fp2 f = team_miller_multi(t, K);
    Emitter e_one = EMIT(gs, id, off_is_one);
    if (!active) e_one.base = nullptr;
    bool res = team_final_exp_is_one(t, f, e_one);
    int32_t* r = gs.desc[id.s].result;
    if (active && j == 0 && r) r[id.i] = step_result(gs.desc[id.s], id.i, res);
}
// blsw_verify_batch, phase 1: the projective line coefficients of the two G2 points of an instance (vpairing.hpp). Lanes [0, n): the signature against
// -g1; lanes [n, 2 n): H(m) (ws.h, homogeneous) against the public key. A point that is the identity (or failed to decode: zeros) gets zero lines —
// the verdict of such an instance is false by its status.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_vlines(uint64_t n, Workspace ws, const uint64_t* __restrict__ pk_xy, const uint64_t* __restrict__ sig_xy,
                                                                                      Fp* lines_sig, Fp* lines_h) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n) return;
    const bool is_h = t >= n;
    const uint64_t i = is_h ? t - n : t;
    Fp2 qx, qy;
    Fp px, py;
    if (!is_h) {
        const Fp* p = reinterpret_cast<const Fp*>(sig_xy + i * 24);
        qx = {ld_fp(p), ld_fp(p + 1)};
        qy = {ld_fp(p + 2), ld_fp(p + 3)};
        px = K_G1_GEN_X();
        py = K_G1_GEN_NEG_Y();
    } else {
        const Proj<OpsFp2> h = ld_proj2(ws.h + i, n);
        const Fp2 zi = fp2_inv_inl(h.z);  // 0 for the identity
        qx = fp2_mul_inl(h.x, zi);
        qy = fp2_mul_inl(h.y, zi);
        const Fp* p = reinterpret_cast<const Fp*>(pk_xy + i * 12);
        px = ld_fp(p);
        py = ld_fp(p + 1);
    }
    vline_chain(qx, qy, px, py, CoeffStrided{(is_h ? lines_h : lines_sig) + i, n});
}
// phase 2: six lanes per instance fold the 2 x 68 lines (value-only Miller loop), final exponentiation, is_one; the verdict also needs both decode
// statuses BLSW_ST_OK (bls.rs:431-447: identity key, on-curve and subgroup checks are errors, which tests/tests.rs:244-263 count as false)
__global__ __launch_bounds__(64) void k_verify_team(uint64_t n, const Fp* lines_sig, const Fp* lines_h, const int32_t* __restrict__ status, int32_t* __restrict__ result) {
    __shared__ Fp2 lds[BLSW_TEAMS_PER_WAVE * TS_NSLOTS];
    const uint32_t team = threadIdx.x / 6, j = threadIdx.x % 6;
    const uint64_t I0 = (uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE + team;
    const bool active = team < BLSW_TEAMS_PER_WAVE && I0 < n;
    const uint64_t I = active ? I0 : 0;
    TeamLanesValues t;
    t.slots = lds + (active ? team : 0) * TS_NSLOTS;
    t.j = j;
    t.active = active;
    t.coeff_sig = {const_cast<Fp*>(lines_sig) + I, n};
    t.coeff_h = {const_cast<Fp*>(lines_h) + I, n};
    t.e = {nullptr, 0};
    Fp2 f = team_miller_values(t);
    const bool one = team_final_exp_is_one(t, f, Emitter{nullptr, 0});
    if (active && j == 0) result[I] = (one && status[2 * I] == BLSW_ST_OK && status[2 * I + 1] == BLSW_ST_OK) ? 1 : 0;
}
void launch_verify_values(uint64_t n, const Workspace& ws, const uint64_t* pk_xy, const uint64_t* sig_xy, Fp* lines_sig, Fp* lines_h, const int32_t* status, int32_t* result,
                          hipStream_t st) {
    hipLaunchKernelGGL(k_vlines, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st, n, ws, pk_xy, sig_xy, lines_sig, lines_h);
    hipLaunchKernelGGL(k_verify_team, dim3((unsigned)((n + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, st, n, (const Fp*)lines_sig, (const Fp*)lines_h, status, result);
}
void launch_pairing(const Group& g, const Modes& m, hipStream_t st) {
    if (g.L.params_mode)
        hipLaunchKernelGGL(k_pairing_team_pv, dim3((unsigned)((g.N + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, st, g);
    else if (!m.pairing_team)
        hipLaunchKernelGGL(k_pairing, dim3((unsigned)((g.N + 63) / 64)), dim3(64), 0, st, g);
    else
        hipLaunchKernelGGL(k_pairing_team, dim3((unsigned)((g.N + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, st, g);
}

}  // namespace blsw
