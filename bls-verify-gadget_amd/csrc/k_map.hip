// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
// Two compilations (build.py, kcommon.hpp: BLSW_K). Grouped-engine compilation of THIS unit: programs inlined into the kernel and
// two waves per SIMD (<= 256 registers) — measured +4 % on the 20-step job, neutral in the steady state (profiles/r03_ab_chain_builds.txt);
// -DBLSW_OUTLINE_MAP restores the out-of-line build for A/B runs. Direct-mode compilation (*_inl): inlined, the whole register file.
// Latency compilation (*_q, -DBLSW_KVARIANT_QUAD): inlined, one chain on the four lanes of a quad (fp.hpp: quads).
#if defined(BLSW_KVARIANT_INL) || defined(BLSW_KVARIANT_QUAD) || !defined(BLSW_OUTLINE_MAP)
#define BLSW_INLINE_CHAINS 1
#endif
#include "kcommon.hpp"
#if !defined(BLSW_KVARIANT_INL) && !defined(BLSW_KVARIANT_QUAD) && !defined(BLSW_OUTLINE_MAP)
#define BLSW_CHAIN_ATTR BLSW_ATTR_W2
#else
#define BLSW_CHAIN_ATTR
#endif

namespace blsw {

// lanes [0, N): u0 -> Q0 ; lanes [N, 2N): u1 -> Q1
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_map)(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    const uint64_t t = item_index();  // latency compilation (k_map_q): four lanes per item
    if (t >= 2 * g.N) return;
    uint32_t which = t >= g.N;
    uint64_t I = which ? t - g.N : t;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Fp2 u = ld_fp2(g.ws.u + (uint64_t)(2 * which) * N + I, N);
    Proj<OpsFp2> q = chain_map_to_curve(which ? EMITJ(g, id, off_map1, stride_hash) : EMITJ(g, id, off_map0, stride_hash), u);
    if (!item_leader()) return;
    Fp* o = g.ws.q + (uint64_t)(6 * which) * N + I;
    st_fp(o, q.x.c0);
    st_fp(o + N, q.x.c1);
    st_fp(o + 2 * N, q.y.c0);
    st_fp(o + 3 * N, q.y.c1);
    st_fp(o + 4 * N, q.z.c0);
    st_fp(o + 5 * N, q.z.c1);
}

}  // namespace blsw
