// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
// Two compilations (build.py, kcommon.hpp: BLSW_K). Grouped-engine compilation of THIS unit: programs out of line, one wave per SIMD with
// ~100 registers left to the streaming kernels on the same SIMD (inlined it takes the whole file and the expansion starves; inlined at
// 256 registers it spills 2 400 registers into its hot loop: profiles/r03_ab_chain_builds.txt); -DBLSW_INL_COFACTOR / -DBLSW_W2_COFACTOR for A/B runs.
// Direct-mode compilation (*_inl): inlined, the whole register file.
#if defined(BLSW_KVARIANT_INL) || defined(BLSW_INL_COFACTOR)
#define BLSW_INLINE_CHAINS 1
#endif
#include "kcommon.hpp"
#include "cofactor_par.hpp"
#if !defined(BLSW_KVARIANT_INL) && defined(BLSW_W2_COFACTOR)
#define BLSW_CHAIN_ATTR BLSW_ATTR_W2
#else
#define BLSW_CHAIN_ATTR
#endif

namespace blsw {

__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_cofactor)(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Proj<OpsFp2> q0 = ld_proj2(g.ws.q + I, N), q1 = ld_proj2(g.ws.q + 6 * N + I, N);
    Proj<OpsFp2> h = chain_cofactor(EMITJ(g, id, off_add, stride_hash), EMITJ(g, id, off_cofactor, stride_hash), q0, q1);
    Fp* o = g.ws.h + I;
    st_fp(o, h.x.c0);
    st_fp(o + N, h.x.c1);
    st_fp(o + 2 * N, h.y.c0);
    st_fp(o + 3 * N, h.y.c1);
    st_fp(o + 4 * N, h.z.c0);
    st_fp(o + 5 * N, h.z.c1);
}

// The same segment with the three 255-bit chunks of the scalar on three lanes (cofactor_par.hpp): lanes [0, N) run chunk 0 (and emit
// Q0 + Q1 and the to_affine of the sum), [N, 2N) chunk 1, [2N, 3N) chunk 2 — waves are chunk-homogeneous, so a wave still appends
// whole 3 KiB rows to its tile; what the chunks leave for the join is parked in the line-coefficient rows of prepare_g2(H(m)),
// which that kernel writes afterwards.
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_cofactor_chunk)(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, N = g.N;
    if (t >= 3 * N) return;
    const int c = t >= 2 * N ? 2 : (t >= N ? 1 : 0);
    const uint64_t I = t - (uint64_t)c * N;
    LaneId id = lane_id(g, I);
    Proj<OpsFp2> q0 = ld_proj2(g.ws.q + I, N), q1 = ld_proj2(g.ws.q + 6 * N + I, N);
    chain_cofactor_chunk(EMITJ(g, id, off_add, stride_hash), EMITJ(g, id, off_cofactor, stride_hash), q0, q1, c, CoeffStrided{g.ws.coeff_h + I, N});
}

__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_cofactor_join)(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Proj<OpsFp2> h = chain_cofactor_join(EMITJ(g, id, off_cofactor, stride_hash), CoeffStrided{g.ws.coeff_h + I, N});
    Fp* o = g.ws.h + I;
    st_fp(o, h.x.c0);
    st_fp(o + N, h.x.c1);
    st_fp(o + 2 * N, h.y.c0);
    st_fp(o + 3 * N, h.y.c1);
    st_fp(o + 4 * N, h.z.c0);
    st_fp(o + 5 * N, h.z.c1);
}

}  // namespace blsw
