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
        const uint64_t p = LAYOUT == 0 ? ((uint64_t)blockIdx.x * ITERS + k) * THREADS + threadIdx.x : ((uint64_t)k * gridDim.x + blockIdx.x) * THREADS + threadIdx.x;
        if (p < n16) dst[p] = v;
        if (k + 1 < ITERS) {
            if (MODE == 1) __builtin_amdgcn_s_sleep(S);
            if (MODE == 2 || MODE == 3 || (MODE == 4 && k == 0)) __builtin_amdgcn_s_waitcnt(0);
            if (MODE == 3) __syncthreads();
        }
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
    hipLaunchKernelGGL((k_fill<T, I, M, S, L>), dim3((unsigned)((n16 + (uint64_t)T * I - 1) / ((uint64_t)T * I))), dim3(T), 0, 0, d, n16); })
int main() {
    const double bytes = 1024.0 * 655107 * 48;
    const uint64_t n16 = (uint64_t)(bytes / 16);
    uint4* d;
    CK(hipMalloc(&d, (size_t)bytes + (1 << 20)));
    RUN(256, 1, 0, 0, 0);
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
