// LAB PROGRAM: pacing of multi-store streaming waves (why does one store per wave reach 6.8 TB/s and eight only 5.8?)
//   hipcc -O3 --offload-arch=gfx950 -o build/fill_lab tools/fill_lab.hip && build/fill_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
// MODE 0 back to back; 1 s_sleep S between stores; 2 vmcnt(0) between stores; 3 vmcnt(0) + barrier; 4 one store then the
// rest after a vmcnt(0) (warm the path); LAYOUT 0: workgroup-contiguous (iteration k writes THREADS pieces at k*THREADS);
// LAYOUT 1: iteration k of workgroup g writes chunk k * gridDim + g (every iteration is one sequential sweep of the grid)
template <int THREADS, int ITERS, int MODE, int S, int LAYOUT>
__global__ __launch_bounds__(THREADS) void k_fill(uint4* __restrict__ dst, uint64_t n16) {
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
        // LAYOUT 2: XCD-aware — workgroup b runs on XCD b % 8 (round-robin dispatch) and writes only chunks == b (mod 8):
        // chunk (b % 8) + 8 * ((b / 8) * ITERS + k); LAYOUT 3: the same with the iterations of a workgroup one sweep apart
        // (chunk (b % 8) + 8 * (k * (gridDim.x / 8) + b / 8))
        uint64_t chunk;
        if (LAYOUT == 0) chunk = (uint64_t)blockIdx.x * ITERS + k;
        else if (LAYOUT == 1) chunk = (uint64_t)k * gridDim.x + blockIdx.x;
        else if (LAYOUT == 2) chunk = (blockIdx.x & 7) + 8ull * ((uint64_t)(blockIdx.x >> 3) * ITERS + k);
        else chunk = (blockIdx.x & 7) + 8ull * ((uint64_t)k * (gridDim.x >> 3) + (blockIdx.x >> 3));
        const uint64_t p = chunk * THREADS + threadIdx.x;
        if (p < n16) dst[p] = v;
        if (k + 1 < ITERS) {
            if (MODE == 1) __builtin_amdgcn_s_sleep(S);
            if (MODE == 2 || MODE == 3 || (MODE == 4 && k == 0)) __builtin_amdgcn_s_waitcnt(0);
            if (MODE == 3) __syncthreads();
        }
    }
}
// the expansion's address structure with constant data: instance y owns the 16-byte pieces [seg0(y), seg0(y) + n_pieces) of a
// vector `stride16` pieces long; XCD-aware chunks as in k_sha_expand_xcd (ITERS chunks per workgroup in SWEEPS groups)
template <int ITERS, int SWEEPS, int MATH>
__global__ __launch_bounds__(256) void k_fill_inst(uint4* __restrict__ base, uint64_t stride16, uint32_t off16, uint32_t n_pieces) {
    uint4* out = base + blockIdx.y * stride16 + off16;
    const uintptr_t addr16 = reinterpret_cast<uintptr_t>(out) >> 4;
    const uint32_t P0 = (256 - (uint32_t)(addr16 % 256)) % 256;
    if (blockIdx.x == 0 && threadIdx.x < P0) out[threadIdx.x] = make_uint4(1, 2, 3, 4);
    constexpr int J = ITERS / SWEEPS;
    const uint32_t abs0 = (uint32_t)((addr16 + P0) >> 8);
    const uint32_t r = (blockIdx.x - abs0) & 7, G = gridDim.x >> 3, g = blockIdx.x >> 3;
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
        const uint32_t q = (k / J) * (G * J) + g * J + (k % J);
        const uint32_t p = P0 + (r + 8 * q) * 256 + threadIdx.x;
        uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
        if (MATH) {  // the expansion's per-piece arithmetic without its loads
            const uint32_t e = p / 3, c = p - 3 * e, m = 0u - (((e * 2654435761u) >> (e & 31)) & 1u);
            v = make_uint4((c == 0 ? 0x11u : c == 1 ? 0x22u : 0x33u) & m, (c == 0 ? 0x44u : c == 1 ? 0x55u : 0x66u) & m, (c == 0 ? 0x77u : c == 1 ? 0x88u : 0x99u) & m, m);
        }
        if (p < n_pieces) out[p] = v;
    }
}
template <class F>
static void run(const char* name, double bytes, F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 6; i++) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms / 6, bytes / (ms / 6 * 1e-3) / 1e9);
    fflush(stdout);
}
#define RUN(T, I, M, S, L) run(#T " thr x " #I " mode " #M " sleep " #S " layout " #L, bytes, [&] { \
    hipLaunchKernelGGL((k_fill<T, I, M, S, L>), dim3((unsigned)(((n16 + (uint64_t)T * I - 1) / ((uint64_t)T * I) + 7) / 8 * 8)), dim3(T), 0, 0, d, n16); })
int main() {
    const double bytes = 1024.0 * 655107 * 48;
    const uint64_t n16 = (uint64_t)(bytes / 16);
    uint4* d;
    CK(hipMalloc(&d, (size_t)bytes + (1 << 20)));
    {
        const uint32_t sha_bits = 655107, n_witness = 707427, off = 14619, n_pieces = sha_bits * 3, chunks = (n_pieces + 255) / 256 + 1;
#define RUNI(I, S, M) run("instances: 256 thr x " #I " in " #S " sweeps, math " #M, bytes, [&] { \
    hipLaunchKernelGGL((k_fill_inst<I, S, M>), dim3(8 * ((chunks + 8 * I - 1) / (8 * I)), 1024), dim3(256), 0, 0, dw, (uint64_t)n_witness * 3, off * 3, n_pieces); })
        uint4* dw;
        CK(hipMalloc(&dw, 1024ull * n_witness * 48));
        RUNI(8, 8, 0);
        RUNI(8, 8, 1);
        RUNI(8, 1, 0);
        RUNI(8, 2, 0);
        RUNI(4, 4, 0);
        RUNI(2, 2, 0);
        RUNI(1, 1, 0);
        RUNI(1, 1, 1);
        RUNI(16, 16, 0);
        CK(hipFree(dw));
    }
    RUN(256, 1, 0, 0, 0);
    RUN(256, 2, 0, 0, 2);
    RUN(256, 4, 0, 0, 2);
    RUN(256, 8, 0, 0, 2);
    RUN(256, 16, 0, 0, 2);
    RUN(256, 2, 0, 0, 3);
    RUN(256, 8, 0, 0, 3);
    RUN(512, 8, 0, 0, 2);
    RUN(128, 8, 0, 0, 2);
    RUN(64, 8, 0, 0, 2);
    RUN(1024, 4, 0, 0, 2);
    RUN(256, 2, 0, 0, 0);
    RUN(256, 8, 0, 0, 0);
    RUN(256, 8, 1, 1, 0);
    RUN(256, 8, 1, 4, 0);
    RUN(256, 8, 1, 16, 0);
    RUN(256, 8, 1, 64, 0);
    RUN(256, 8, 2, 0, 0);
    RUN(256, 8, 3, 0, 0);
    RUN(256, 8, 4, 0, 0);
    RUN(256, 2, 2, 0, 0);
    RUN(256, 2, 0, 0, 1);
    RUN(256, 8, 0, 0, 1);
    RUN(256, 8, 2, 0, 1);
    RUN(256, 32, 0, 0, 1);
    RUN(256, 32, 2, 0, 1);
    RUN(64, 8, 0, 0, 0);
    RUN(64, 8, 2, 0, 0);
    RUN(64, 32, 2, 0, 1);
    RUN(1024, 1, 0, 0, 0);
    RUN(1024, 4, 0, 0, 1);
    RUN(1024, 4, 2, 0, 1);
    RUN(512, 8, 2, 0, 1);
    RUN(256, 1, 0, 0, 0);
    return 0;
}
