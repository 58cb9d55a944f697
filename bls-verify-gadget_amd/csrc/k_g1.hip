// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
// Two compilations (build.py, kcommon.hpp: BLSW_K). Grouped-engine compilation of THIS unit: programs inlined into the kernel and
// two waves per SIMD (<= 256 registers) — measured +4 % on the 20-step job, neutral in the steady state (profiles/r03_ab_chain_builds.txt);
// -DBLSW_OUTLINE_G1 restores the out-of-line build for A/B runs. Direct-mode compilation (*_inl): inlined, the whole register file.
#if defined(BLSW_KVARIANT_INL) || !defined(BLSW_OUTLINE_G1)
#define BLSW_INLINE_CHAINS 1
#endif
#include "kcommon.hpp"
#if !defined(BLSW_KVARIANT_INL) && !defined(BLSW_OUTLINE_G1)
#define BLSW_CHAIN_ATTR BLSW_ATTR_W2
#else
#define BLSW_CHAIN_ATTR
#endif

namespace blsw {

// Lanes [0, N): the public key of a (pk, msg) pair. ParametersVar allocated as witnesses (L.params_mode, single-key circuit: constraints.rs:198-211
// with AllocationMode::Witness): lanes [N, 2 N) run the same chain on the generator of instance I - N — G1Var::new_variable, then g1.negate()
// (linear) and prepare_g1(&g1_neg) = to_affine; no enforce_not_equal on it (constraints.rs:97-99 is about the public key only).
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_g1)(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool params = g.L.params_mode && I >= g.N;
    if (params) I -= g.N;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].pk + (uint64_t)id.f * 12);
    Emitter e_alloc = EMITJ(g, id, off_pk_alloc, stride_pk_alloc), e_nz = EMITJ(g, id, off_pk_not_zero, stride_pk_not_zero),
            e_prep = EMITJ(g, id, off_prep_pk, stride_prep_pk);
    Fp x = ld_fp(p), y = ld_fp(p + 1);
    if (params) {
        e_alloc = EMIT(g, id, off_params_alloc);
        e_prep = EMIT(g, id, off_prep_g1);
        e_nz.base = nullptr;
        x = K_G1_GEN_X();
        y = fp_neg(K_G1_GEN_NEG_Y());
    }
    Proj<OpsFp> pk;
    if (g.L.pk_mode && !params) {
        // PublicKeyVar::new_variable(Input) (constraints.rs:214-232) = new_variable_omit_prime_order_check: x, y, z are public inputs (1 .. 3 of
        // instance_assignment), no witnesses, no in-circuit prime-order check
        const bool inf = fp_is_zero(x) && fp_is_zero(y);
        pk = {inf ? fp_zero() : x, inf ? fp_one() : y, inf ? fp_zero() : fp_one()};
        put_instance(g, id, 1, pk.x);
        put_instance(g, id, 2, pk.y);
        put_instance(g, id, 3, pk.z);
    } else {
        pk = chain_g1_alloc_only(e_alloc, x, y);
    }
    if (!params) put_instance(g, id, 0, fp_one());  // instance_assignment[0]
    if (params) pk.y = fp_neg(pk.y);
    G1ChainOut o = chain_g1_post(e_nz, e_prep, pk);
    if (params) return;  // its affine form is the constant the pairing kernel uses
    st_fp(g.ws.pkaff + I, o.ax);
    st_fp(g.ws.pkaff + g.N + I, o.ay);
}

// aggregate_verify: lane t = k * N + I allocates key k of instance I (N * n_keys lanes), result to ws.keyproj
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_agg_keys)(Group g, Fp* keyproj) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t N = g.N, nk = g.L.n_keys;
    if (t >= N * nk) return;
    uint32_t k = (uint32_t)(t / N);
    LaneId id = lane_id(g, t - (uint64_t)k * N);
    const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].keys + ((uint64_t)id.i * nk + k) * 12);
    Proj<OpsFp> r = chain_g1_alloc_only(emitter(g, id, g.L.off_keys + k * SEG_PK_ALLOC, g.LS.off_keys + k * SEG_PK_ALLOC, false), ld_fp(p), ld_fp(p + 1));
    Fp* o = keyproj + t;
    st_fp(o, r.x);
    st_fp(o + N * nk, r.y);
    st_fp(o + 2 * N * nk, r.z);
}
struct KeyProjSrc {
    const Fp* p;  // keyproj + I
    uint64_t N, total;
    __device__ __forceinline__ Proj<OpsFp> ld(uint32_t k) const {
        const Fp* q = p + (uint64_t)k * N;
        return {ld_fp(q), ld_fp(q + total), ld_fp(q + 2 * total)};
    }
};
// aggregate_verify: bitmap booleans, mapped_aggregate, then pk != 0 and prepare_g1 on the aggregated key
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_agg_sum)(Group g, const Fp* keyproj) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint32_t nk = g.L.n_keys;
    const uint8_t* bm = g.desc[id.s].bitmap + (uint64_t)id.i * nk;
    Emitter eb = EMIT(g, id, off_bitmap);
    for (uint32_t k = 0; k < nk; k++) eb.put_bool(bm[k] != 0);  // Boolean::new_witness per key (constraints.rs:414-419)
    KeyProjSrc src = {keyproj + I, g.N, g.N * nk};
    uint32_t count = 0;
    Proj<OpsFp> pk = chain_mapped_aggregate(EMIT(g, id, off_count), EMIT(g, id, off_agg), src, bm, nk, &count);
    G1ChainOut o = chain_g1_post(EMIT(g, id, off_pk_not_zero), EMIT(g, id, off_prep_pk), pk);
    st_fp(g.ws.pkaff + I, o.ax);
    st_fp(g.ws.pkaff + g.N + I, o.ay);
    uint32_t* c = g.desc[id.s].count;
    if (c) c[id.i] = count;
}

}  // namespace blsw
