// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
#include "kcommon.hpp"

namespace blsw {

// Miller loop + final exponentiation + is_one
__global__ __launch_bounds__(64) void k_pairing(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Fp pkx = ld_fp(g.ws.pkaff + I), pky = ld_fp(g.ws.pkaff + N + I);
    CoeffStrided ch = {g.ws.coeff_h + I, N};
    CoeffStrided cs = {g.ws.coeff_sig + I, g.ws.n_sig};
    Fp12 f = chain_miller(EMIT(g, id, off_miller), pkx, pky, cs, ch);
    bool res = chain_final_exp_is_one(EMIT(g, id, off_final_exp), EMIT(g, id, off_is_one), f);
    int32_t* r = g.desc[id.s].result;
    if (r) r[id.i] = step_result(g.desc[id.s], id.i, res);
}

}  // namespace blsw
