// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
// Two compilations (build.py, kcommon.hpp: BLSW_K). Grouped-engine compilation of THIS unit: programs out of line, one wave per SIMD with
// ~100 registers left to the streaming kernels on the same SIMD (inlined it takes the whole file and the expansion starves; inlined at
// 256 registers it spills 2 400 registers into its hot loop: profiles/r03_ab_chain_builds.txt); -DBLSW_INL_G2 / -DBLSW_W2_G2 for A/B runs.
// Direct-mode compilation (*_inl): inlined, the whole register file. Latency compilation (*_q): inlined, one chain on the four lanes of a quad.
#if defined(BLSW_KVARIANT_INL) || defined(BLSW_KVARIANT_QUAD) || defined(BLSW_INL_G2)
#define BLSW_INLINE_CHAINS 1
#endif
#include "kcommon.hpp"
#if !defined(BLSW_KVARIANT_INL) && !defined(BLSW_KVARIANT_QUAD) && defined(BLSW_W2_G2)
#define BLSW_CHAIN_ATTR BLSW_ATTR_W2
#else
#define BLSW_CHAIN_ATTR
#endif

namespace blsw {

__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_g2_alloc)(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    const uint64_t I = item_index();  // latency compilation (k_g2_alloc_q): four lanes per item
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].sig + (uint64_t)id.i * 24);
    Fp2 sx = {ld_fp(p), ld_fp(p + 1)}, sy = {ld_fp(p + 2), ld_fp(p + 3)};
    chain_g2_alloc(EMIT(g, id, off_sig_alloc), sx, sy);
}

}  // namespace blsw
