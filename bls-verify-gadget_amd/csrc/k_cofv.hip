// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
// clear_cofactor2 "values first" (cofactor_vf.hpp): the latency form of the cofactor segment, for small launch groups.
//   k_cofv_chain   phase 1, one lane per (pk, msg) pair: Q0 + Q1, to_affine (witnesses), the 636 Jacobian doublings and 1 / Z_D as values
//   k_cofv_aff     phase 2a, one lane per doubling index: the affine 2^D P
//   k_cofv_dbl_w   phase 2b, one lane per doubling: its ten witnesses (off the critical path: the engine runs it beside phase 3)
//   k_cofv_acc     phase 3, one lane per chunk: the chunk's additions as a mixed Jacobian chain, 1 / Z as values
//   k_cofv_add_w   phase 4, one lane per addition: its eight witnesses
//   k_cofv_join    phase 5, one lane per pair: folds the chunks (the statements of k_cofactor_join)
// Two compilations (build.py): this one (programs inlined; the parallel phases at two waves per SIMD) and the latency compilation (-DBLSW_KVARIANT_QUAD:
// k_cofv_chain_q, k_cofv_acc_q — the two serial phases on the four lanes of a quad, fp.hpp).
#define BLSW_INLINE_CHAINS 1
#include "kcommon.hpp"
#define BLSW_CHAIN_ATTR  // the serial phases and the join: the whole register file (at 256 registers the join spills 2 000 into its additions)

namespace blsw {

__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_cofv_chain)(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t I = item_index(), N = g.N;
    if (I >= N) return;
    LaneId id = lane_id(g, I);
    Proj<OpsFp2> q0 = ld_proj2(g.ws.q + I, N), q1 = ld_proj2(g.ws.q + 6 * N + I, N);
    cofv_chain(EMITJ(g, id, off_add, stride_hash), EMITJ(g, id, off_cofactor, stride_hash), q0, q1, CoeffStrided{g.ws.cofv + I, N}, CoeffStrided{g.ws.coeff_h + I, N});
}

// lanes [0, N) chunk 0, [N, 2 N) chunk 1, [2 N, 3 N) chunk 2 (chunk-homogeneous waves)
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_cofv_acc)(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t t = item_index(), N = g.N;
    if (t >= 3 * N) return;
    const int c = t >= 2 * N ? 2 : (t >= N ? 1 : 0);
    const uint64_t I = t - (uint64_t)c * N;
    cofv_acc_chain(c, CoeffStrided{g.ws.cofv + I, N}, CoeffStrided{g.ws.coeff_h + I, N});
}

#ifndef BLSW_KVARIANT_QUAD
// thread t -> (doubling D = t / N, pair I = t % N): the lanes of a wave share D (N is a multiple of 64, or the tail wave mixes two), so its ten
// witness rows are whole 3 KiB rows of the wave's staging tile and its scratch reads are contiguous
__global__ __launch_bounds__(64) BLSW_ATTR_W2 void k_cofv_aff(Group g) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, N = g.N;
    if (t >= (uint64_t)BLSW_H_EFF_NBITS * N) return;
    const uint32_t D = (uint32_t)(t / N);
    const uint64_t I = t - (uint64_t)D * N;
    cofv_affine(D, CoeffStrided{g.ws.cofv + I, N});
}
__global__ __launch_bounds__(64) BLSW_ATTR_W2 void k_cofv_dbl_w(Group g) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, N = g.N;
    if (t >= (uint64_t)BLSW_H_EFF_NBITS * N) return;
    const uint32_t D = (uint32_t)(t / N);
    const uint64_t I = t - (uint64_t)D * N;
    LaneId id = lane_id(g, I);
    cofv_dbl_w(EMITJ(g, id, off_cofactor, stride_hash), D, CoeffStrided{g.ws.cofv + I, N});
}

// thread t -> (addition a = t / N over the three chunks' additions in order, pair I = t % N)
__global__ __launch_bounds__(64) BLSW_ATTR_W2 void k_cofv_add_w(Group g) {
    constexpr CofvPlan plan = cofv_plan();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, N = g.N;
    const uint32_t total = plan.n_adds[0] + plan.n_adds[1] + plan.n_adds[2];
    if (t >= (uint64_t)total * N) return;
    uint32_t a = (uint32_t)(t / N);
    const uint64_t I = t - (uint64_t)a * N;
    int c = 0;
    if (a >= plan.n_adds[0]) {
        a -= plan.n_adds[0];
        c = 1;
        if (a >= plan.n_adds[1]) {
            a -= plan.n_adds[1];
            c = 2;
        }
    }
    LaneId id = lane_id(g, I);
    cofv_add_w(EMITJ(g, id, off_cofactor, stride_hash), c, a, CoeffStrided{g.ws.cofv + I, N});
}

__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void k_cofv_join(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, N = g.N;
    if (I >= N) return;
    LaneId id = lane_id(g, I);
    Proj<OpsFp2> h = cofv_join(EMITJ(g, id, off_cofactor, stride_hash), CoeffStrided{g.ws.cofv + I, N}, CoeffStrided{g.ws.coeff_h + I, N});
    Fp* o = g.ws.h + I;
    st_fp(o, h.x.c0);
    st_fp(o + N, h.x.c1);
    st_fp(o + 2 * N, h.y.c0);
    st_fp(o + 3 * N, h.y.c1);
    st_fp(o + 4 * N, h.z.c0);
    st_fp(o + 5 * N, h.z.c1);
}
#endif

}  // namespace blsw
