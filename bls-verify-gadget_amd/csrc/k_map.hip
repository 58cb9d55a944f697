// libblsw.so, one translation unit per kernel family (see kcommon.cuh, build.py).
#if defined(BLSW_KVARIANT_INL) || defined(BLSW_CHAINS_INLINED) || defined(BLSW_INL_MAP)  // build variant: the chain programs inlined into the kernel (fp.cuh: BLSW_FN)
#define BLSW_INLINE_CHAINS 1
#endif
#include "kcommon.cuh"
#if defined(BLSW_W2_ALL) || defined(BLSW_W2_MAP)  // build variant: two waves per SIMD (256 registers)
#define BLSW_CHAIN_ATTR BLSW_ATTR_W2
#else
#define BLSW_CHAIN_ATTR
#endif

namespace blsw {

// lanes [0, N): u0 -> Q0 ; lanes [N, 2N): u1 -> Q1
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_map)(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * g.N) return;
    uint32_t which = t >= g.N;
    uint64_t I = which ? t - g.N : t;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Fp2 u = ld_fp2(g.ws.u + (uint64_t)(2 * which) * N + I, N);
    Proj<OpsFp2> q = chain_map_to_curve(which ? EMITJ(g, id, off_map1, stride_hash) : EMITJ(g, id, off_map0, stride_hash), u);
    Fp* o = g.ws.q + (uint64_t)(6 * which) * N + I;
    st_fp(o, q.x.c0);
    st_fp(o + N, q.x.c1);
    st_fp(o + 2 * N, q.y.c0);
    st_fp(o + 3 * N, q.y.c1);
    st_fp(o + 4 * N, q.z.c0);
    st_fp(o + 5 * N, q.z.c1);
}

}  // namespace blsw
