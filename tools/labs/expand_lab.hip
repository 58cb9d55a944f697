// EXPERIMENT HARNESS (not part of libblsw.so): variants of the bit -> Fp expansion kernel (k_sha_expand, the HBM-bound kernel
// of the path) timed ALONE against plain fill kernels of the same size, to see how far the store pattern is from the box's
// write rate.   hipcc -O3 --offload-arch=gfx950 -o /tmp/expand_lab tools/expand_lab.hip && /tmp/expand_lab [n_instances]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

static constexpr uint32_t SHA_BITS = 655107, N_WITNESS = 707427, OFF_EXPAND = 14619;
#define R1_LIMBS {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u}

// ---- V0: the shipped kernel (384 threads, 8 iterations, one 16-byte piece per thread and iteration)
template <int THREADS, int ITERS, int NT>
__global__ __launch_bounds__(THREADS) void k_expand(const uint32_t* __restrict__ bits, uint64_t sha_words, uint32_t sha_bits, uint32_t off_expand,
                                                    uint64_t* __restrict__ d_witness, uint64_t stride) {
    constexpr uint32_t R1[12] = R1_LIMBS;
    constexpr uint32_t EPI = THREADS / 3;
    const uint64_t inst = blockIdx.y;
    uint4* out = reinterpret_cast<uint4*>(d_witness + (inst * stride + off_expand) * 6);
    const uint32_t* b = bits + (inst >> 6) * sha_words * 64 + (inst & 63);
    const uint32_t P0 = (16u - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) & 15u)) & 15u;
    const uint32_t t = threadIdx.x, pt = P0 + t, c = pt % 3;
    const uint32_t e0 = blockIdx.x * (EPI * ITERS) + pt / 3;
    if (blockIdx.x == 0 && t < P0) {
        const uint32_t he = t / 3, hc = t % 3;
        uint32_t m = 0u - ((b[0] >> he) & 1u);
        out[(uint64_t)he * 3 + hc] = make_uint4(R1[4 * hc] & m, R1[4 * hc + 1] & m, R1[4 * hc + 2] & m, R1[4 * hc + 3] & m);
    }
    uint4 rc;
    rc.x = c == 0 ? R1[0] : (c == 1 ? R1[4] : R1[8]);
    rc.y = c == 0 ? R1[1] : (c == 1 ? R1[5] : R1[9]);
    rc.z = c == 0 ? R1[2] : (c == 1 ? R1[6] : R1[10]);
    rc.w = c == 0 ? R1[3] : (c == 1 ? R1[7] : R1[11]);
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
        uint32_t e = e0 + EPI * k;
        if (e < sha_bits) {
            uint32_t w = b[(uint64_t)(e >> 5) * 64];
            uint32_t m = 0u - ((w >> (e & 31)) & 1u);
            uint4 v = make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m);
            if (NT) {
                __builtin_nontemporal_store(v.x, &out[(uint64_t)e * 3 + c].x);
                __builtin_nontemporal_store(v.y, &out[(uint64_t)e * 3 + c].y);
                __builtin_nontemporal_store(v.z, &out[(uint64_t)e * 3 + c].z);
                __builtin_nontemporal_store(v.w, &out[(uint64_t)e * 3 + c].w);
            } else
                out[(uint64_t)e * 3 + c] = v;
        }
    }
}

// ---- V1: same geometry, no bit reads (how much do the reads / mask arithmetic cost?)
template <int THREADS, int ITERS>
__global__ __launch_bounds__(THREADS) void k_fill_same_geometry(uint32_t sha_bits, uint32_t off_expand, uint64_t* __restrict__ d_witness, uint64_t stride) {
    constexpr uint32_t EPI = THREADS / 3;
    const uint64_t inst = blockIdx.y;
    uint4* out = reinterpret_cast<uint4*>(d_witness + (inst * stride + off_expand) * 6);
    const uint32_t P0 = (16u - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) & 15u)) & 15u;
    const uint32_t t = threadIdx.x, pt = P0 + t, c = pt % 3;
    const uint32_t e0 = blockIdx.x * (EPI * ITERS) + pt / 3;
    uint4 v = make_uint4(t, c, e0, 7);
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
        uint32_t e = e0 + EPI * k;
        if (e < sha_bits) out[(uint64_t)e * 3 + c] = v;
    }
}

// ---- V2: linear sweep of the WHOLE tensor region by a 1-D grid (pieces numbered across instances): no 2-D grid, no per-instance tail
template <int THREADS, int ITERS>
__global__ __launch_bounds__(THREADS) void k_expand_linear(const uint32_t* __restrict__ bits, uint64_t sha_words, uint32_t sha_bits, uint32_t off_expand,
                                                           uint64_t* __restrict__ d_witness, uint64_t stride, uint32_t blocks_per_inst) {
    constexpr uint32_t R1[12] = R1_LIMBS;
    constexpr uint32_t EPI = THREADS / 3;
    const uint32_t inst = blockIdx.x / blocks_per_inst, bx = blockIdx.x - inst * blocks_per_inst;
    uint4* out = reinterpret_cast<uint4*>(d_witness + ((uint64_t)inst * stride + off_expand) * 6);
    const uint32_t* b = bits + (uint64_t)(inst >> 6) * sha_words * 64 + (inst & 63);
    const uint32_t P0 = (16u - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) & 15u)) & 15u;
    const uint32_t t = threadIdx.x, pt = P0 + t, c = pt % 3;
    const uint32_t e0 = bx * (EPI * ITERS) + pt / 3;
    if (bx == 0 && t < P0) {
        const uint32_t he = t / 3, hc = t % 3;
        uint32_t m = 0u - ((b[0] >> he) & 1u);
        out[(uint64_t)he * 3 + hc] = make_uint4(R1[4 * hc] & m, R1[4 * hc + 1] & m, R1[4 * hc + 2] & m, R1[4 * hc + 3] & m);
    }
    uint4 rc;
    rc.x = c == 0 ? R1[0] : (c == 1 ? R1[4] : R1[8]);
    rc.y = c == 0 ? R1[1] : (c == 1 ? R1[5] : R1[9]);
    rc.z = c == 0 ? R1[2] : (c == 1 ? R1[6] : R1[10]);
    rc.w = c == 0 ? R1[3] : (c == 1 ? R1[7] : R1[11]);
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
        uint32_t e = e0 + EPI * k;
        if (e < sha_bits) {
            uint32_t w = b[(uint64_t)(e >> 5) * 64];
            uint32_t m = 0u - ((w >> (e & 31)) & 1u);
            out[(uint64_t)e * 3 + c] = make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m);
        }
    }
}

// ---- V3: bit words loaded ONCE per block into LDS / registers up front: one coalesced read phase, then stores only
template <int THREADS, int ITERS>
__global__ __launch_bounds__(THREADS) void k_expand_prefetch(const uint32_t* __restrict__ bits, uint64_t sha_words, uint32_t sha_bits, uint32_t off_expand,
                                                             uint64_t* __restrict__ d_witness, uint64_t stride) {
    constexpr uint32_t R1[12] = R1_LIMBS;
    constexpr uint32_t EPI = THREADS / 3;
    constexpr uint32_t WORDS = (EPI * ITERS + 31) / 32 + 2;
    __shared__ uint32_t sw[WORDS];
    const uint64_t inst = blockIdx.y;
    uint4* out = reinterpret_cast<uint4*>(d_witness + (inst * stride + off_expand) * 6);
    const uint32_t* b = bits + (inst >> 6) * sha_words * 64 + (inst & 63);
    const uint32_t P0 = (16u - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) & 15u)) & 15u;
    const uint32_t t = threadIdx.x, pt = P0 + t, c = pt % 3;
    const uint32_t ebase = blockIdx.x * (EPI * ITERS);
    const uint32_t w0 = ebase >> 5;
    const uint32_t wmax = (sha_bits + 31) / 32;
    if (t < WORDS) sw[t] = (w0 + t < wmax + 1) ? b[(uint64_t)(w0 + t) * 64] : 0u;
    __syncthreads();
    const uint32_t e0 = ebase + pt / 3;
    if (blockIdx.x == 0 && t < P0) {
        const uint32_t he = t / 3, hc = t % 3;
        uint32_t m = 0u - ((sw[0] >> he) & 1u);
        out[(uint64_t)he * 3 + hc] = make_uint4(R1[4 * hc] & m, R1[4 * hc + 1] & m, R1[4 * hc + 2] & m, R1[4 * hc + 3] & m);
    }
    uint4 rc;
    rc.x = c == 0 ? R1[0] : (c == 1 ? R1[4] : R1[8]);
    rc.y = c == 0 ? R1[1] : (c == 1 ? R1[5] : R1[9]);
    rc.z = c == 0 ? R1[2] : (c == 1 ? R1[6] : R1[10]);
    rc.w = c == 0 ? R1[3] : (c == 1 ? R1[7] : R1[11]);
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
        uint32_t e = e0 + EPI * k;
        if (e < sha_bits) {
            uint32_t w = sw[(e >> 5) - w0];
            uint32_t m = 0u - ((w >> (e & 31)) & 1u);
            out[(uint64_t)e * 3 + c] = make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m);
        }
    }
}

// ---- plain fills of the same bytes: contiguous region, 16 B per thread per iteration
template <int ITERS>
__global__ __launch_bounds__(256) void k_fill(uint4* __restrict__ out, uint64_t n16) {
    uint64_t q = ((uint64_t)blockIdx.x * ITERS) * 256 + threadIdx.x;
    uint4 v = make_uint4(1, 2, 3, 4);
#pragma unroll
    for (int k = 0; k < ITERS; k++, q += 256)
        if (q < n16) out[q] = v;
}
__global__ __launch_bounds__(256) void k_fill_gridstride(uint4* __restrict__ out, uint64_t n16) {
    uint4 v = make_uint4(1, 2, 3, 4);
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < n16; q += (uint64_t)gridDim.x * 256) out[q] = v;
}

// fill where every WAVE writes ITERS consecutive KiB (its stores are sequential in memory), waves of a block adjacent
template <int THREADS, int ITERS>
__global__ __launch_bounds__(THREADS) void k_fill_wavecontig(uint4* __restrict__ out, uint64_t n16) {
    const uint64_t wave = (uint64_t)blockIdx.x * (THREADS / 64) + threadIdx.x / 64;
    uint64_t q = wave * ITERS * 64 + (threadIdx.x & 63);
    uint4 v = make_uint4(1, 2, 3, 4);
#pragma unroll
    for (int k = 0; k < ITERS; k++, q += 64)
        if (q < n16) out[q] = v;
}
template <int THREADS, int ITERS>
__global__ __launch_bounds__(THREADS) void k_fill_strided(uint4* __restrict__ out, uint64_t n16) {
    uint64_t q = ((uint64_t)blockIdx.x * ITERS) * THREADS + threadIdx.x;
    uint4 v = make_uint4(1, 2, 3, 4);
#pragma unroll
    for (int k = 0; k < ITERS; k++, q += THREADS)
        if (q < n16) out[q] = v;
}
// expansion, every wave writes ITERS consecutive KiB of its instance's segment (piece p = element p / 3, column p % 3)
template <int THREADS, int ITERS>
__global__ __launch_bounds__(THREADS) void k_expand_wavecontig(const uint32_t* __restrict__ bits, uint64_t sha_words, uint32_t sha_bits, uint32_t off_expand,
                                                               uint64_t* __restrict__ d_witness, uint64_t stride) {
    constexpr uint32_t R1[12] = R1_LIMBS;
    const uint64_t inst = blockIdx.y;
    uint4* out = reinterpret_cast<uint4*>(d_witness + (inst * stride + off_expand) * 6);
    const uint32_t* b = bits + (inst >> 6) * sha_words * 64 + (inst & 63);
    const uint32_t P0 = (16u - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) & 15u)) & 15u;
    const uint32_t n_pieces = sha_bits * 3;
    if (blockIdx.x == 0 && threadIdx.x < P0) {
        const uint32_t he = threadIdx.x / 3, hc = threadIdx.x % 3;
        uint32_t m = 0u - ((b[0] >> he) & 1u);
        out[(uint64_t)he * 3 + hc] = make_uint4(R1[4 * hc] & m, R1[4 * hc + 1] & m, R1[4 * hc + 2] & m, R1[4 * hc + 3] & m);
    }
    const uint32_t wave = blockIdx.x * (THREADS / 64) + threadIdx.x / 64;
    uint32_t p = P0 + wave * (ITERS * 64) + (threadIdx.x & 63);
    uint32_t c = p % 3, e = p / 3;
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
        if (p < n_pieces) {
            uint32_t w = b[(uint64_t)(e >> 5) * 64];
            uint32_t m = 0u - ((w >> (e & 31)) & 1u);
            uint4 v;
            v.x = (c == 0 ? R1[0] : (c == 1 ? R1[4] : R1[8])) & m;
            v.y = (c == 0 ? R1[1] : (c == 1 ? R1[5] : R1[9])) & m;
            v.z = (c == 0 ? R1[2] : (c == 1 ? R1[6] : R1[10])) & m;
            v.w = (c == 0 ? R1[3] : (c == 1 ? R1[7] : R1[11])) & m;
            out[p] = v;
        }
        // next piece of this lane: + 64 pieces = + 21 elements + 1 column
        p += 64;
        c += 1;
        e += 21;
        if (c == 3) {
            c = 0;
            e += 1;
        }
    }
}

// expansion, one store per thread, blocks aligned to ALIGN_PIECES * 16 bytes of the ADDRESS (not of the segment): block b
// writes the aligned chunk b of the instance's segment; chunk 0 also covers the unaligned head
template <int THREADS, int READ_BITS>
__global__ __launch_bounds__(THREADS) void k_expand_aligned(const uint32_t* __restrict__ bits, uint64_t sha_words, uint32_t sha_bits, uint32_t off_expand,
                                                            uint64_t* __restrict__ d_witness, uint64_t stride) {
    constexpr uint32_t R1[12] = R1_LIMBS;
    const uint64_t inst = blockIdx.y;
    uint4* out = reinterpret_cast<uint4*>(d_witness + (inst * stride + off_expand) * 6);
    const uint32_t* b = bits + (inst >> 6) * sha_words * 64 + (inst & 63);
    // piece index of the first THREADS*16-byte boundary at or after the segment start
    const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) % THREADS);
    const uint32_t P0 = (THREADS - mis) % THREADS;
    const uint32_t n_pieces = sha_bits * 3;
    // block 0: head pieces [0, P0); block b >= 1: pieces [P0 + (b-1)*THREADS, P0 + b*THREADS)
    uint32_t p;
    if (blockIdx.x == 0) {
        if (threadIdx.x >= P0) return;
        p = threadIdx.x;
    } else
        p = P0 + (blockIdx.x - 1) * THREADS + threadIdx.x;
    if (p >= n_pieces) return;
    const uint32_t e = p / 3, c = p - 3 * e;
    uint32_t m = 0xffffffffu;
    if (READ_BITS) {
        uint32_t w = b[(uint64_t)(e >> 5) * 64];
        m = 0u - ((w >> (e & 31)) & 1u);
    }
    uint4 v;
    v.x = (c == 0 ? R1[0] : (c == 1 ? R1[4] : R1[8])) & m;
    v.y = (c == 0 ? R1[1] : (c == 1 ? R1[5] : R1[9])) & m;
    v.z = (c == 0 ? R1[2] : (c == 1 ? R1[6] : R1[10])) & m;
    v.w = (c == 0 ? R1[3] : (c == 1 ? R1[7] : R1[11])) & m;
    out[p] = v;
}

// aligned chunks, several per workgroup, SEQUENTIAL IN TIME: the workgroup writes aligned chunk after aligned chunk; WAIT = 1 drains
// the store and re-synchronises the four waves between chunks (never more than one 4 KiB chunk of a workgroup in flight)
template <int ITERS, int WAIT>
__global__ __launch_bounds__(256) void k_expand_aligned_seq(const uint32_t* __restrict__ bits, uint64_t sha_words, uint32_t sha_bits, uint32_t off_expand,
                                                            uint64_t* __restrict__ d_witness, uint64_t stride) {
    constexpr uint32_t R1[12] = R1_LIMBS;
    const uint64_t inst = blockIdx.y;
    uint4* out = reinterpret_cast<uint4*>(d_witness + (inst * stride + off_expand) * 6);
    const uint32_t* b = bits + (inst >> 6) * sha_words * 64 + (inst & 63);
    const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) % 256);
    const uint32_t P0 = (256 - mis) % 256;
    const uint32_t n_pieces = sha_bits * 3;
    if (blockIdx.x == 0 && threadIdx.x < P0) {  // head pieces
        const uint32_t e = threadIdx.x / 3, c = threadIdx.x % 3;
        uint32_t m = 0u - ((b[(uint64_t)(e >> 5) * 64] >> (e & 31)) & 1u);
        out[threadIdx.x] = make_uint4(R1[4 * c] & m, R1[4 * c + 1] & m, R1[4 * c + 2] & m, R1[4 * c + 3] & m);
    }
    uint32_t p = P0 + blockIdx.x * (256 * ITERS) + threadIdx.x;
#pragma unroll 1
    for (int k = 0; k < ITERS; k++, p += 256) {
        if (p < n_pieces) {
            const uint32_t e = p / 3, c = p - 3 * e;
            const uint32_t w = b[(uint64_t)(e >> 5) * 64];
            const uint32_t m = 0u - ((w >> (e & 31)) & 1u);
            uint4 v;
            v.x = (c == 0 ? R1[0] : (c == 1 ? R1[4] : R1[8])) & m;
            v.y = (c == 0 ? R1[1] : (c == 1 ? R1[5] : R1[9])) & m;
            v.z = (c == 0 ? R1[2] : (c == 1 ? R1[6] : R1[10])) & m;
            v.w = (c == 0 ? R1[3] : (c == 1 ? R1[7] : R1[11])) & m;
            out[p] = v;
        }
        if (WAIT) {
            __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
            __syncthreads();
        }
    }
}

// background load shaped like the curve chains: one wave per workgroup, ~256 VGPRs (one such wave per SIMD), an Fp-product's worth of
// 64-bit multiply-adds per iteration, then one 48-byte witness per lane, each lane appending to its own stream
__global__ __launch_bounds__(64) void k_background(uint4* __restrict__ dst, uint32_t lane_pieces, uint32_t iters, const int* stop) {
    asm volatile("; keep a large register allocation" ::: "v250");
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u;
    uint64_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 1, a4 = x + 2, a5 = y + 2, a6 = x + 3, a7 = y + 3;
    uint4* p = dst + ((uint64_t)blockIdx.x * 64 + threadIdx.x) * lane_pieces;
    uint32_t pos = 0;
    for (uint32_t i = 0; i < iters; i++) {
        for (int r = 0; r < 36; r++) {
            a0 = (uint64_t)(uint32_t)a0 * x + a0;
            a1 = (uint64_t)(uint32_t)a1 * y + a1;
            a2 = (uint64_t)(uint32_t)a2 * x + a2;
            a3 = (uint64_t)(uint32_t)a3 * y + a3;
            a4 = (uint64_t)(uint32_t)a4 * x + a4;
            a5 = (uint64_t)(uint32_t)a5 * y + a5;
            a6 = (uint64_t)(uint32_t)a6 * x + a6;
            a7 = (uint64_t)(uint32_t)a7 * y + a7;
        }
        p[pos] = make_uint4((uint32_t)a0, (uint32_t)a1, (uint32_t)a2, (uint32_t)a3);
        p[pos + 1] = make_uint4((uint32_t)a4, (uint32_t)a5, (uint32_t)a6, (uint32_t)a7);
        p[pos + 2] = make_uint4((uint32_t)(a0 >> 32), (uint32_t)(a1 >> 32), (uint32_t)(a2 >> 32), (uint32_t)(a3 >> 32));
        pos += 3;
        if (pos + 3 > lane_pieces) pos = 0;
        if ((i & 255) == 255 && __builtin_nontemporal_load(stop)) break;
    }
}

template <class F>
static double time_ms(F launch, int reps = 8) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024;
    const uint64_t stride = N_WITNESS, sha_words = (SHA_BITS + 31) / 32 + 1;
    uint64_t* d_wit;
    uint32_t* d_bits;
    CK(hipMalloc(&d_wit, n * stride * 48));
    CK(hipMalloc(&d_bits, sha_words * ((n + 63) / 64 * 64) * 4));
    std::vector<uint32_t> hb(sha_words * ((n + 63) / 64 * 64));
    for (auto& w : hb) w = (uint32_t)rand() * 2654435761u;
    CK(hipMemcpy(d_bits, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    const double bytes = (double)n * SHA_BITS * 48;
    auto report = [&](const char* name, double ms) { printf("%-44s %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0); };
#define GRID(T, I) dim3((SHA_BITS + (T / 3) * I - 1) / ((T / 3) * I), (unsigned)n)
    report("V0 shipped: 384 thr x 8 it", time_ms([&] { hipLaunchKernelGGL((k_expand<384, 8, 0>), GRID(384, 8), dim3(384), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("V0 nontemporal", time_ms([&] { hipLaunchKernelGGL((k_expand<384, 8, 1>), GRID(384, 8), dim3(384), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("V0 192 thr x 8", time_ms([&] { hipLaunchKernelGGL((k_expand<192, 8, 0>), GRID(192, 8), dim3(192), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("V0 384 thr x 32", time_ms([&] { hipLaunchKernelGGL((k_expand<384, 32, 0>), GRID(384, 32), dim3(384), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("V0 768 thr x 4", time_ms([&] { hipLaunchKernelGGL((k_expand<768, 4, 0>), GRID(768, 4), dim3(768), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("V0 384 thr x 2", time_ms([&] { hipLaunchKernelGGL((k_expand<384, 2, 0>), GRID(384, 2), dim3(384), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("V1 same geometry, no bit reads", time_ms([&] { hipLaunchKernelGGL((k_fill_same_geometry<384, 8>), GRID(384, 8), dim3(384), 0, 0, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    {
        const uint32_t bpi = (SHA_BITS + 128 * 8 - 1) / (128 * 8);
        report("V2 1-D grid, instance-major", time_ms([&] { hipLaunchKernelGGL((k_expand_linear<384, 8>), dim3((unsigned)(bpi * n)), dim3(384), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride, bpi); }));
    }
    report("V3 bit words via LDS", time_ms([&] { hipLaunchKernelGGL((k_expand_prefetch<384, 8>), GRID(384, 8), dim3(384), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("V3 bit words via LDS, 384 x 32", time_ms([&] { hipLaunchKernelGGL((k_expand_prefetch<384, 32>), GRID(384, 32), dim3(384), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
#define GRIDW(T, I) dim3((SHA_BITS * 3 + 16 + (T) * (I) - 1) / ((T) * (I)), (unsigned)n)
    report("E 384 thr x 1 (6 KiB per block)", time_ms([&] { hipLaunchKernelGGL((k_expand<384, 1, 0>), GRID(384, 1), dim3(384), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E 192 thr x 1", time_ms([&] { hipLaunchKernelGGL((k_expand<192, 1, 0>), GRID(192, 1), dim3(192), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E 768 thr x 1", time_ms([&] { hipLaunchKernelGGL((k_expand<768, 1, 0>), GRID(768, 1), dim3(768), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E wave-contiguous 256 thr x 1", time_ms([&] { hipLaunchKernelGGL((k_expand_wavecontig<256, 1>), GRIDW(256, 1), dim3(256), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E wave-contiguous 256 thr x 4", time_ms([&] { hipLaunchKernelGGL((k_expand_wavecontig<256, 4>), GRIDW(256, 4), dim3(256), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E wave-contiguous 256 thr x 8", time_ms([&] { hipLaunchKernelGGL((k_expand_wavecontig<256, 8>), GRIDW(256, 8), dim3(256), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E wave-contiguous 64 thr x 8", time_ms([&] { hipLaunchKernelGGL((k_expand_wavecontig<64, 8>), GRIDW(64, 8), dim3(64), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E wave-contiguous 1024 thr x 1", time_ms([&] { hipLaunchKernelGGL((k_expand_wavecontig<1024, 1>), GRIDW(1024, 1), dim3(1024), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
#define GRIDA(T) dim3((SHA_BITS * 3 + (T) - 1) / (T) + 1, (unsigned)n)
    report("E aligned 256 thr (4 KiB blocks)", time_ms([&] { hipLaunchKernelGGL((k_expand_aligned<256, 1>), GRIDA(256), dim3(256), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E aligned 256 thr, no bit reads", time_ms([&] { hipLaunchKernelGGL((k_expand_aligned<256, 0>), GRIDA(256), dim3(256), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E aligned 512 thr (8 KiB blocks)", time_ms([&] { hipLaunchKernelGGL((k_expand_aligned<512, 1>), GRIDA(512), dim3(512), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E aligned 128 thr (2 KiB blocks)", time_ms([&] { hipLaunchKernelGGL((k_expand_aligned<128, 1>), GRIDA(128), dim3(128), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    report("E aligned 1024 thr (16 KiB blocks)", time_ms([&] { hipLaunchKernelGGL((k_expand_aligned<1024, 1>), GRIDA(1024), dim3(1024), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    const uint64_t n16 = (uint64_t)(bytes / 16);
    report("fill, contiguous, 256 thr x 8", time_ms([&] { hipLaunchKernelGGL((k_fill<8>), dim3((unsigned)((n16 + 2047) / 2048)), dim3(256), 0, 0, (uint4*)d_wit, n16); }));
    report("fill, contiguous, 256 thr x 1", time_ms([&] { hipLaunchKernelGGL((k_fill<1>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (uint4*)d_wit, n16); }));
    report("fill, grid-stride, 256 CUs x 8 blocks", time_ms([&] { hipLaunchKernelGGL(k_fill_gridstride, dim3(2048), dim3(256), 0, 0, (uint4*)d_wit, n16); }));
    report("fill, grid-stride, 256 CUs x 32 blocks", time_ms([&] { hipLaunchKernelGGL(k_fill_gridstride, dim3(8192), dim3(256), 0, 0, (uint4*)d_wit, n16); }));
    report("fill strided 256 x 2", time_ms([&] { hipLaunchKernelGGL((k_fill_strided<256, 2>), dim3((unsigned)((n16 + 511) / 512)), dim3(256), 0, 0, (uint4*)d_wit, n16); }));
    report("fill strided 256 x 4", time_ms([&] { hipLaunchKernelGGL((k_fill_strided<256, 4>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (uint4*)d_wit, n16); }));
    report("fill 64 thr x 1", time_ms([&] { hipLaunchKernelGGL((k_fill_strided<64, 1>), dim3((unsigned)((n16 + 63) / 64)), dim3(64), 0, 0, (uint4*)d_wit, n16); }));
    report("fill 1024 thr x 1", time_ms([&] { hipLaunchKernelGGL((k_fill_strided<1024, 1>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(1024), 0, 0, (uint4*)d_wit, n16); }));
    report("fill wave-contiguous 256 x 4", time_ms([&] { hipLaunchKernelGGL((k_fill_wavecontig<256, 4>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (uint4*)d_wit, n16); }));
    report("fill wave-contiguous 256 x 8", time_ms([&] { hipLaunchKernelGGL((k_fill_wavecontig<256, 8>), dim3((unsigned)((n16 + 2047) / 2048)), dim3(256), 0, 0, (uint4*)d_wit, n16); }));
    report("fill wave-contiguous 64 x 16", time_ms([&] { hipLaunchKernelGGL((k_fill_wavecontig<64, 16>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(64), 0, 0, (uint4*)d_wit, n16); }));
    report("fill 256 x 1, region offset by 1 KiB", time_ms([&] { hipLaunchKernelGGL((k_fill_strided<256, 1>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (uint4*)d_wit + 64, n16 - 64); }));
    report("fill 256 x 1, region offset by 48 B", time_ms([&] { hipLaunchKernelGGL((k_fill_strided<256, 1>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (uint4*)d_wit + 3, n16 - 64); }));
    report("fill 128 x 1", time_ms([&] { hipLaunchKernelGGL((k_fill_strided<128, 1>), dim3((unsigned)((n16 + 127) / 128)), dim3(128), 0, 0, (uint4*)d_wit, n16); }));
    report("fill 512 x 1", time_ms([&] { hipLaunchKernelGGL((k_fill_strided<512, 1>), dim3((unsigned)((n16 + 511) / 512)), dim3(512), 0, 0, (uint4*)d_wit, n16); }));
    report("fill 384 x 1", time_ms([&] { hipLaunchKernelGGL((k_fill_strided<384, 1>), dim3((unsigned)((n16 + 383) / 384)), dim3(384), 0, 0, (uint4*)d_wit, n16); }));
    report("hipMemsetAsync", time_ms([&] { CK(hipMemsetAsync(d_wit, 0x5a, (size_t)bytes, 0)); }));
    // ---- the same variants UNDER LOAD (only when a background wave count is given: the synthetic load turned out far heavier
    // than the real chains): background waves on a low-priority stream, the variant on a high-priority stream
    if (argc > 2) {
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        hipStream_t sb, sx;
        CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, lo));
        CK(hipStreamCreateWithPriority(&sx, hipStreamNonBlocking, hi));
        const uint32_t bg_waves = argc > 2 ? atoi(argv[2]) : 1536, lane_pieces = 4096;
        uint4* d_bg;
        int* d_stop;
        CK(hipMalloc(&d_bg, (size_t)bg_waves * 64 * lane_pieces * 16));
        CK(hipMalloc(&d_stop, 4));
        auto loaded = [&](const char* name, auto launch) {
            CK(hipMemset(d_stop, 0, 4));
            hipLaunchKernelGGL(k_background, dim3(bg_waves), dim3(64), 0, sb, d_bg, lane_pieces, 100000000u, d_stop);
            hipEvent_t a, b;
            CK(hipEventCreate(&a));
            CK(hipEventCreate(&b));
            for (int i = 0; i < 3; i++) launch(sx);
            CK(hipEventRecord(a, sx));
            const int reps = 10;
            for (int i = 0; i < reps; i++) launch(sx);
            CK(hipEventRecord(b, sx));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            int one = 1;
            CK(hipMemcpyAsync(d_stop, &one, 4, hipMemcpyHostToDevice, sx));
            CK(hipDeviceSynchronize());
            char buf[128];
            snprintf(buf, sizeof buf, "LOADED %s", name);
            report(buf, ms / reps);
        };
        loaded("V0 shipped 384 x 8", [&](hipStream_t st) { hipLaunchKernelGGL((k_expand<384, 8, 0>), GRID(384, 8), dim3(384), 0, st, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); });
        loaded("aligned 256 x 1", [&](hipStream_t st) { hipLaunchKernelGGL((k_expand_aligned<256, 1>), GRIDA(256), dim3(256), 0, st, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); });
        loaded("aligned 512 x 1", [&](hipStream_t st) { hipLaunchKernelGGL((k_expand_aligned<512, 1>), GRIDA(512), dim3(512), 0, st, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); });
#define GRIDS(I) dim3((SHA_BITS * 3 + 256 * (I) - 1) / (256 * (I)) + 1, (unsigned)n)
        loaded("aligned seq 256 x 4 wait", [&](hipStream_t st) { hipLaunchKernelGGL((k_expand_aligned_seq<4, 1>), GRIDS(4), dim3(256), 0, st, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); });
        loaded("aligned seq 256 x 16 wait", [&](hipStream_t st) { hipLaunchKernelGGL((k_expand_aligned_seq<16, 1>), GRIDS(16), dim3(256), 0, st, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); });
        loaded("aligned seq 256 x 4 nowait", [&](hipStream_t st) { hipLaunchKernelGGL((k_expand_aligned_seq<4, 0>), GRIDS(4), dim3(256), 0, st, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); });
        loaded("aligned seq 256 x 16 nowait", [&](hipStream_t st) { hipLaunchKernelGGL((k_expand_aligned_seq<16, 0>), GRIDS(16), dim3(256), 0, st, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); });
        loaded("fill 256 x 1", [&](hipStream_t st) { hipLaunchKernelGGL((k_fill_strided<256, 1>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (uint4*)d_wit, n16); });
        loaded("fill 256 x 8", [&](hipStream_t st) { hipLaunchKernelGGL((k_fill<8>), dim3((unsigned)((n16 + 2047) / 2048)), dim3(256), 0, st, (uint4*)d_wit, n16); });
        // alone, same variants
        report("ALONE aligned seq 256 x 4 wait", time_ms([&] { hipLaunchKernelGGL((k_expand_aligned_seq<4, 1>), GRIDS(4), dim3(256), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
        report("ALONE aligned seq 256 x 16 wait", time_ms([&] { hipLaunchKernelGGL((k_expand_aligned_seq<16, 1>), GRIDS(16), dim3(256), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
        report("ALONE aligned seq 256 x 4 nowait", time_ms([&] { hipLaunchKernelGGL((k_expand_aligned_seq<4, 0>), GRIDS(4), dim3(256), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
        report("ALONE aligned seq 256 x 16 nowait", time_ms([&] { hipLaunchKernelGGL((k_expand_aligned_seq<16, 0>), GRIDS(16), dim3(256), 0, 0, d_bits, sha_words, SHA_BITS, OFF_EXPAND, d_wit, stride); }));
    }
    return 0;
}
