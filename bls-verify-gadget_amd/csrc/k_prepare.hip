// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
// Two compilations (build.py, kcommon.hpp: BLSW_K). Grouped-engine compilation of THIS unit: programs inlined into the kernel and
// two waves per SIMD (<= 256 registers) — measured +4 % on the 20-step job, neutral in the steady state (profiles/r03_ab_chain_builds.txt);
// -DBLSW_OUTLINE_PREPARE restores the out-of-line build for A/B runs. Direct-mode compilation (*_inl): inlined, the whole register file.
// Latency compilation (*_q, -DBLSW_KVARIANT_QUAD): inlined, one chain on the four lanes of a quad (fp.hpp: quads).
#if defined(BLSW_KVARIANT_INL) || defined(BLSW_KVARIANT_QUAD) || !defined(BLSW_OUTLINE_PREPARE)
#define BLSW_INLINE_CHAINS 1
#endif
#include "kcommon.hpp"
#if !defined(BLSW_KVARIANT_INL) && !defined(BLSW_KVARIANT_QUAD) && !defined(BLSW_OUTLINE_PREPARE)
#define BLSW_CHAIN_ATTR BLSW_ATTR_W2
#else
#define BLSW_CHAIN_ATTR
#endif

namespace blsw {

// which = 0: prepare_g2(H(m)) ; which = 1: prepare_g2(sig)
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_prepare)(Group g, int which) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    const uint64_t I = item_index();  // latency compilation (k_prepare_q): four lanes per item
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Proj<OpsFp2> q;
    if (which == 0) {
        q = ld_proj2(g.ws.h + I, N);
    } else {
        const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].sig + (uint64_t)id.i * 24);
        Fp2 sx = {ld_fp(p), ld_fp(p + 1)}, sy = {ld_fp(p + 2), ld_fp(p + 3)};
        bool inf = fp2_is_zero(sx) && fp2_is_zero(sy);
        q.x = inf ? fp2_zero() : sx;
        q.y = inf ? fp2_one() : sy;
        q.z = inf ? fp2_zero() : fp2_one();
    }
    CoeffStrided out = which == 0 ? CoeffStrided{g.ws.coeff_h + I, N} : CoeffStrided{g.ws.coeff_sig + I, g.ws.n_sig};
    chain_prepare_g2(which == 0 ? EMITJ(g, id, off_prep_h, stride_prep_h) : EMIT(g, id, off_prep_sig), q, out);
}

}  // namespace blsw
