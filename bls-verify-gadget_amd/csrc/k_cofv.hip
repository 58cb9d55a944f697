// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
// clear_cofactor2 "values first" (cofactor_vf.hpp): the latency form of the cofactor segment, for small launch groups.
//   k_cofv_chain   phase 1, segment s, one lane per (pk, msg) pair: [Q0 + Q1, to_affine (witnesses),] the segment's Jacobian doublings as values
//   k_cofv_bwd     phase 1b, segment s, one lane per pair: 1 / Z_D backwards from the segment's last point (one inversion)
//   k_cofv_aff     phase 2a, one lane per doubling index of the segment: the affine 2^D P
//   k_cofv_acc     phase 3, segment s, one lane per pair: the additions of the segment's chunk in the segment as a mixed Jacobian chain
//   k_cofv_dbl_w   phase 2b, one lane per doubling: its ten witnesses                           } off the critical path: after the join,
//   k_cofv_az      phase 3b, one lane per chunk: 1 / Z1 before every addition, backwards        } on another stream
//   k_cofv_add_w   phase 4, one lane per addition: its eight witnesses                          }
//   k_cofv_join    phase 5, one lane per pair: folds the chunks (the statements of k_cofactor_join)
// kcommon.hpp's launch_cofactor pipelines the segments over four streams: phases 1b / 2a / 3 of a segment run beside phase 1 of the next ones.
// Two compilations (build.py): this one (programs inlined; the parallel phases at two waves per SIMD) and the latency compilation (-DBLSW_KVARIANT_QUAD:
// k_cofv_chain_q, k_cofv_bwd_q, k_cofv_acc_q, k_cofv_az_q — the serial phases on the four lanes of a quad, fp.hpp).
#define BLSW_INLINE_CHAINS 1
#include "kcommon.hpp"
#define BLSW_CHAIN_ATTR  // the serial phases and the join: the whole register file (at 256 registers the join spills 2 000 into its additions)

namespace blsw {

__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_cofv_chain)(Group g, int s) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t I = item_index(), N = g.N;
    if (I >= N) return;
    LaneId id = lane_id(g, I);
    auto load_q = [&](Proj<OpsFp2>& q0, Proj<OpsFp2>& q1) {
        q0 = ld_proj2(g.ws.q + I, N);
        q1 = ld_proj2(g.ws.q + 6 * N + I, N);
    };
    cofv_chain_seg(s, EMITJ(g, id, off_add, stride_hash), EMITJ(g, id, off_cofactor, stride_hash), load_q, CoeffStrided{g.ws.cofv + I, N}, CoeffStrided{g.ws.coeff_h + I, N});
}
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_cofv_bwd)(Group g, int s) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t I = item_index(), N = g.N;
    if (I >= N) return;
    cofv_bwd(s, CoeffStrided{g.ws.cofv + I, N});
}
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_cofv_acc)(Group g, int s) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t I = item_index(), N = g.N;
    if (I >= N) return;
    cofv_acc_seg(s, CoeffStrided{g.ws.cofv + I, N}, CoeffStrided{g.ws.coeff_h + I, N});
}

// lanes [0, N) chunk 0, [N, 2 N) chunk 1, [2 N, 3 N) chunk 2 (chunk-homogeneous waves)
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_cofv_az)(Group g) {
    const uint64_t t = item_index(), N = g.N;
    if (t >= 3 * N) return;
    const int c = t >= 2 * N ? 2 : (t >= N ? 1 : 0);
    const uint64_t I = t - (uint64_t)c * N;
    cofv_acc_az(c, CoeffStrided{g.ws.cofv + I, N});
}

#ifndef BLSW_KVARIANT_QUAD
// thread t -> (doubling D = t / N, pair I = t % N): the lanes of a wave share D (N is a multiple of 64, or the tail wave mixes two), so its ten
// witness rows are whole 3 KiB rows of the wave's staging tile and its scratch reads are contiguous
__global__ __launch_bounds__(64) BLSW_ATTR_W2 void k_cofv_aff(Group g, uint32_t lo, uint32_t cnt) {  // the points D in [lo, lo + cnt)
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, N = g.N;
    if (t >= (uint64_t)cnt * N) return;
    const uint32_t d = (uint32_t)(t / N);
    const uint64_t I = t - (uint64_t)d * N;
    cofv_affine(lo + d, CoeffStrided{g.ws.cofv + I, N});
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
