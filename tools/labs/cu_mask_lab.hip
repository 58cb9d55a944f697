// LAB PROGRAM (round 4): can a SUBSET of the compute units carry the expansion's write stream? CU-masked streams (hipExtStreamCreateWithCUMask):
// which CUs does a mask select, and what plain-fill rate do E CUs per XCD reach with one 16-byte store per thread (the geometry that is fastest
// alone and collapses under the chains) and with eight?   hipcc -O3 --offload-arch=gfx950 -o build/cu_mask_lab tools/cu_mask_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_probe(uint32_t* out, int spin) {
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;  // HW_REG_XCC_ID
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID (all 32 bits)
        out[blockIdx.x] = (xcc << 24) | (hw & 0xffffff);
    }
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(8);
}
template <int THREADS, int ITERS>
__global__ __launch_bounds__(THREADS) void k_fill(uint4* __restrict__ dst, uint64_t n16) {
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
        const uint64_t p = ((uint64_t)blockIdx.x * ITERS + k) * THREADS + threadIdx.x;
        if (p < n16) dst[p] = v;
    }
}
int main() {
    const double bytes = 1024.0 * 655107 * 48;
    const uint64_t n16 = (uint64_t)(bytes / 16);
    uint4* d;
    CK(hipMalloc(&d, (size_t)bytes + (1 << 20)));
    uint32_t* probe;
    const int NP = 16384;
    CK(hipMalloc(&probe, NP * 4));
    for (int E : {32, 24, 20, 16, 12, 8}) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 8 * E; i++) mask[i / 32] |= 1u << (i % 32);
        hipStream_t st;
        CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
        hipLaunchKernelGGL(k_probe, dim3(NP), dim3(64), 0, st, probe, 200);
        CK(hipStreamSynchronize(st));
        std::vector<uint32_t> h(NP);
        CK(hipMemcpy(h.data(), probe, NP * 4, hipMemcpyDeviceToHost));
        std::set<uint32_t> cus;
        int per_xcc[16] = {0};
        for (uint32_t v : h) {
            const uint32_t xcc = v >> 24, cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7;
            if (cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu).second) per_xcc[xcc]++;
        }
        printf("E = %2d CUs per XCD in the mask: %3zu distinct CUs seen, per XCD:", E, cus.size());
        for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
        printf("\n");
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        auto run = [&](const char* name, auto launch) {
            launch();
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(a, st));
            for (int i = 0; i < 5; i++) launch();
            CK(hipEventRecord(b, st));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            printf("    %-34s %7.3f ms  %7.1f GB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e9);
            fflush(stdout);
        };
        run("fill 256 thr x 1 (4 KiB aligned)", [&] { hipLaunchKernelGGL((k_fill<256, 1>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, d, n16); });
        run("fill 256 thr x 2", [&] { hipLaunchKernelGGL((k_fill<256, 2>), dim3((unsigned)((n16 + 511) / 512)), dim3(256), 0, st, d, n16); });
        run("fill 384 thr x 8", [&] { hipLaunchKernelGGL((k_fill<384, 8>), dim3((unsigned)((n16 + 3071) / 3072)), dim3(384), 0, st, d, n16); });
        CK(hipStreamDestroy(st));
    }
    return 0;
}
