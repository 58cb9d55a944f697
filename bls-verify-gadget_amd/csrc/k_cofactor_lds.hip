// libblsw.so, one translation unit per kernel family (see kcommon.cuh, build.py).
// EXPERIMENT, compiled only with -DBLSW_COFACTOR_LDS (profiles/r03_ab_chain_builds.txt section 11): the loops of the chunked cofactor chain as a kernel
// of their own — programs inlined, two waves per SIMD, the accumulator in LDS and the doubling point's x parked there across the Fp inversion.
#if defined(BLSW_COFACTOR_LDS)
#define BLSW_INLINE_CHAINS 1
#include "kcommon.cuh"
#include "cofactor_par.cuh"

namespace blsw {

__global__ __launch_bounds__(64) BLSW_ATTR_W2 void k_cofactor_loop_lds(Group g) {
    __shared__ blsw_u32x4 park[18 * 64];
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, N = g.N;
    if (t >= 3 * N) return;
    const int c = t >= 2 * N ? 2 : (t >= N ? 1 : 0);
    const uint64_t I = t - (uint64_t)c * N;
    LaneId id = lane_id(g, I);
    chain_cofactor_chunk_loop_lds(EMITJ(g, id, off_cofactor, stride_hash), c, CoeffStrided{g.ws.coeff_h + I, N}, (cof_lds_u32x4*)park + threadIdx.x);
}

}  // namespace blsw
#endif
