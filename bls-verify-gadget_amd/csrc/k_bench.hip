// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
#include "kcommon.hpp"

namespace blsw {

// ---- micro-benchmarks (roofline denominators): measured on the device, not assumed
__global__ __launch_bounds__(256) void k_bench_mad(uint32_t iters, uint32_t* out) {
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u;
    uint64_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 1, a4 = x + 2, a5 = y + 2, a6 = x + 3, a7 = y + 3;
    for (uint32_t i = 0; i < iters; i++) {  // 8 independent v_mad_u64_u32 chains per lane
        a0 = (uint64_t)(uint32_t)a0 * x + a0;
        a1 = (uint64_t)(uint32_t)a1 * y + a1;
        a2 = (uint64_t)(uint32_t)a2 * x + a2;
        a3 = (uint64_t)(uint32_t)a3 * y + a3;
        a4 = (uint64_t)(uint32_t)a4 * x + a4;
        a5 = (uint64_t)(uint32_t)a5 * y + a5;
        a6 = (uint64_t)(uint32_t)a6 * x + a6;
        a7 = (uint64_t)(uint32_t)a7 * y + a7;
    }
    uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x123456789abcdefull) out[0] = (uint32_t)r;  // keep the chains live
}
// plain fill in the expansion's shipped store geometry (384 threads x 8 pieces of 16 bytes, 6 KiB apart; k_stream.hip: k_sha_expand<384, 8, ...>):
// what THIS box's memory system gives a pure write stream — the yardstick bench.py prints beside the expansion's rate (blsw_fill_rate)
__global__ __launch_bounds__(384) void k_bench_fill(uint4* __restrict__ dst, uint64_t n16) {
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint64_t p = ((uint64_t)blockIdx.x * 8 + k) * 384 + threadIdx.x;
        if (p < n16) dst[p] = v;
    }
}
__global__ __launch_bounds__(64) void k_bench_fpmul(uint32_t iters, uint32_t* out) {
    Fp a = fp_one(), b = fp_one();
    a.l[0] ^= threadIdx.x + 1;
    b.l[1] ^= blockIdx.x + 1;
    for (uint32_t i = 0; i < iters; i++) {
        a = fp_mul(a, b);
        b = fp_mul(b, a);
    }
    if (a.l[0] == 0x12345678u && b.l[3] == 0x9abcdef0u) out[0] = a.l[1];
}

__global__ __launch_bounds__(64) void k_bench_fpmul32(uint32_t iters, uint32_t* out) {  // the 12 x 32-bit CIOS product (cross-check of fp_mul)
    Fp a = fp_one(), b = fp_one();
    a.l[0] ^= threadIdx.x + 1;
    b.l[1] ^= blockIdx.x + 1;
    for (uint32_t i = 0; i < iters; i++) {
        a = fp_mul32(a, b);
        b = fp_mul32(b, a);
    }
    if (a.l[0] == 0x12345678u && b.l[3] == 0x9abcdef0u) out[0] = a.l[1];
}
__global__ __launch_bounds__(64) void k_bench_fpinv(uint32_t iters, uint32_t* out) {
    Fp a = fp_one();
    a.l[0] ^= threadIdx.x * 2654435761u + 1;
    a.l[5] ^= blockIdx.x + 1;
    for (uint32_t i = 0; i < iters; i++) {
        a = fp_inv(a);
        a.l[0] ^= i + 1;  // stays < p: only the low limb changes
        a.l[11] &= 0x0fffffffu;
    }
    if (a.l[0] == 0x12345678u && a.l[3] == 0x9abcdef0u) out[0] = a.l[1];
}
__global__ __launch_bounds__(64) void k_bench_fp2mulw(uint32_t iters, uint32_t* out) {
    Fp2 a = fp2_one(), b = fp2_one();
    a.c0.l[0] ^= threadIdx.x + 1;
    b.c1.l[1] ^= blockIdx.x + 1;
    Emitter e = {nullptr, 0};
    for (uint32_t i = 0; i < iters; i++) {
        a = fp2_mul_w(e, a, b);
        b = fp2_sqr_w(e, b);
    }
    if (a.c0.l[0] == 0x12345678u && b.c0.l[3] == 0x9abcdef0u) out[0] = a.c1.l[1] + e.pos;
}

}  // namespace blsw
