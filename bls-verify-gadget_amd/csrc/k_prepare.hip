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

// the point a prepare chain starts from: which = 0: H(m) from the workspace; which = 1: the signature (the identity as (0, 1, 0))
__device__ __forceinline__ Proj<OpsFp2> prepare_point(const Group& g, const LaneId& id, uint64_t I, int which) {
    Proj<OpsFp2> q;
    if (which == 0) {
        q = ld_proj2(g.ws.h + I, g.N);
    } else {
        const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].sig + (uint64_t)id.i * 24);
        Fp2 sx = {ld_fp(p), ld_fp(p + 1)}, sy = {ld_fp(p + 2), ld_fp(p + 3)};
        bool inf = fp2_is_zero(sx) && fp2_is_zero(sy);
        q.x = inf ? fp2_zero() : sx;
        q.y = inf ? fp2_one() : sy;
        q.z = inf ? fp2_zero() : fp2_one();
    }
    return q;
}
// SignatureVar::new_variable(Input) (constraints.rs:234-249): x, y, z (c0, c1 each) are public inputs after the key's; the prepare chain of the
// signature is the kernel that has the point at hand (there is no allocation chain in this mode)
__device__ __forceinline__ void put_sig_instance(const Group& g, const LaneId& id, const Proj<OpsFp2>& q) {
    if (!g.L.sig_mode || !item_leader()) return;
    const uint32_t k0 = 1 + (g.L.pk_mode ? 3 : 0);
    put_instance(g, id, k0 + 0, q.x.c0);
    put_instance(g, id, k0 + 1, q.x.c1);
    put_instance(g, id, k0 + 2, q.y.c0);
    put_instance(g, id, k0 + 3, q.y.c1);
    put_instance(g, id, k0 + 4, q.z.c0);
    put_instance(g, id, k0 + 5, q.z.c1);
}
#define BLSW_PREPARE_EMIT(g, id, which) ((which) == 0 ? EMITJ(g, id, off_prep_h, stride_prep_h) : EMIT(g, id, off_prep_sig))
#define BLSW_PREPARE_OUT(g, I, which) ((which) == 0 ? CoeffStrided{(g).ws.coeff_h + (I), (g).N} : CoeffStrided{(g).ws.coeff_sig + (I), (g).ws.n_sig})
#define BLSW_PREPARE_SCR(g, I, which) ((which) == 0 ? CoeffStrided{(g).ws.prepv_h + (I), (g).N} : CoeffStrided{(g).ws.prepv_sig + (I), (g).ws.n_sig})

// which = 0: prepare_g2(H(m)) ; which = 1: prepare_g2(sig)
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_prepare)(Group g, int which) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    const uint64_t I = item_index();  // latency compilation (k_prepare_q): four lanes per item
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const Proj<OpsFp2> q = prepare_point(g, id, I, which);
    if (which == 1) put_sig_instance(g, id, q);
    chain_prepare_g2(BLSW_PREPARE_EMIT(g, id, which), q, BLSW_PREPARE_OUT(g, I, which));
}

#ifndef BLSW_KVARIANT_INL
// values first (prepare_vf.hpp), phase 1: to_affine (witnesses) and the chain of points as Jacobian values, one lane (or one quad) per point
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_prepv_chain)(Group g, int which) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t I = item_index();
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const Proj<OpsFp2> q = prepare_point(g, id, I, which);
    if (which == 1) put_sig_instance(g, id, q);
    prepv_chain(BLSW_PREPARE_EMIT(g, id, which), q, BLSW_PREPARE_SCR(g, I, which));
}
#endif
#if !defined(BLSW_KVARIANT_INL) && !defined(BLSW_KVARIANT_QUAD)
// phase 2: thread t -> (step k = t / N, point I = t % N): the step's witnesses and line coefficients
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void k_prepv_step_w(Group g, int which) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, N = g.N;
    if (t >= (uint64_t)BLSW_PREPV_STEPS * N) return;
    const uint32_t k = (uint32_t)(t / N);
    const uint64_t I = t - (uint64_t)k * N;
    LaneId id = lane_id(g, I);
    prepv_step_w(BLSW_PREPARE_EMIT(g, id, which), k, BLSW_PREPARE_SCR(g, I, which), BLSW_PREPARE_OUT(g, I, which));
}
#endif

}  // namespace blsw
