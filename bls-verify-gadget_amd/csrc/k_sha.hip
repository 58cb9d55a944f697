// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
// Two compilations (build.py, kcommon.hpp: BLSW_K). Grouped-engine compilation of THIS unit: programs inlined into the kernel and
// two waves per SIMD (<= 256 registers) — measured +4 % on the 20-step job, neutral in the steady state (profiles/r03_ab_chain_builds.txt);
// -DBLSW_OUTLINE_SHA restores the out-of-line build for A/B runs. Direct-mode compilation (*_inl): inlined, the whole register file.
#if defined(BLSW_KVARIANT_INL) || !defined(BLSW_OUTLINE_SHA)
#define BLSW_INLINE_CHAINS 1
#endif
#include "kcommon.hpp"
#if !defined(BLSW_KVARIANT_INL) && !defined(BLSW_OUTLINE_SHA)
#define BLSW_CHAIN_ATTR BLSW_ATTR_W2
#else
#define BLSW_CHAIN_ATTR
#endif

namespace blsw {

// ---------------------------------------------------------------- kernels (one instance per lane)
// SHA-256 witness bits of expand_message (+ the message bits themselves)
__global__ __launch_bounds__(64) BLSW_CHAIN_ATTR void BLSW_K(k_sha)(Group g, int want_bits, int write_u) {
    __shared__ uint32_t sha_lds[BLSW_BITS_CHUNK_WORDS * 64];  // the wave's word buffer of the bit sink
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint8_t* msg = g.desc[id.s].msg + (uint64_t)id.f * g.msg_len;
    // UInt8::new_witness_vec(msg): 8 booleans per byte, little-endian
    Emitter em = EMITJ(g, id, off_msg, stride_msg);
    for (uint32_t k = 0; k < g.msg_len; k++) {
        uint32_t b = msg[k];
        for (int j = 0; j < 8; j++) em.put_bool((b >> j) & 1);
    }
    uint32_t uw[64];
    if (want_bits) {
        BitSink s;
        s.init_device(sha_lds + threadIdx.x, reinterpret_cast<uint4*>(g.ws.bits + (I >> 6) * bits_tile_words(g.ws.sha_words) + (I & 63) * BLSW_BITS_CHUNK_WORDS));
        expand_message_w(s, msg, g.msg_len, false, uw);
    } else {
        expand_message_values(msg, g.msg_len, uw);  // the device sink always stores: no bits wanted = the value-only SHA
    }
    if (write_u)
        for (int j = 0; j < 4; j++) st_fp(g.ws.u + (uint64_t)j * g.N + I, hash_to_field_elem(uw + 16 * j));
}

#ifndef BLSW_KVARIANT_INL
// value-only expand_message + hash_to_field: hands u0, u1 to k_map without waiting for the witness-bit pass
__global__ __launch_bounds__(64) void k_sha_values(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    uint32_t uw[64];
    expand_message_values(g.desc[id.s].msg + (uint64_t)id.f * g.msg_len, g.msg_len, uw);
    for (int j = 0; j < 4; j++) st_fp(g.ws.u + (uint64_t)j * g.N + I, hash_to_field_elem(uw + 16 * j));
}

#endif

}  // namespace blsw
