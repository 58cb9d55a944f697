// LAB PROGRAM (round 4): what does the HBM deliver to a write stream with reads mixed in? A fill of the expansion's 32.2 GB (384 threads x 8 stores per thread, the
// shipped geometry) on one stream; on a second stream a reader of R GB — sequential 16-byte loads, or 48-byte gathers 3 KiB apart (the placement's access shape) —
// with one load in flight per thread ("polite") or eight. Total bytes over the wall time of both, per read share.
//   hipcc -O3 --offload-arch=gfx950 -o build/mix_lab tools/mix_lab.hip && build/mix_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(384) void k_fill(uint4* __restrict__ dst, uint64_t n16) {
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint64_t p = ((uint64_t)blockIdx.x * 8 + k) * 384 + threadIdx.x;
        if (p < n16) dst[p] = v;
    }
}
// GATHER: piece q of the read = 16 bytes at (q / 3) * 3072 + (q % 3) * 16 (48-byte runs 3 KiB apart); DEPTH loads in flight per thread
template <int GATHER, int DEPTH>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ src, uint64_t n16, uint32_t* sink) {
    uint64_t q = ((uint64_t)blockIdx.x * 8) * 256 + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll 1
    for (int it = 0; it < 8; it += DEPTH) {
        uint4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            const uint64_t p = q + (uint64_t)(it + k) * 256;
            const uint64_t idx = GATHER ? (p / 3) * 192 + (p % 3) : p;
            v[k] = p < n16 ? src[idx] : make_uint4(0, 0, 0, 0);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < DEPTH; k++) acc += v[k].x ^ v[k].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const double wbytes = 1024.0 * 655107 * 48;
    const uint64_t n16 = (uint64_t)(wbytes / 16);
    uint4 *dw, *dr;
    uint32_t* sink;
    const uint64_t rcap16 = (uint64_t)(176e9 / 16);  // source buffer: 176 GB (a gathered read of R bytes spans 64 R)
    CK(hipMalloc(&dw, (size_t)wbytes + (1 << 20)));
    CK(hipMalloc(&dr, rcap16 * 16));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(dr, 1, rcap16 * 16));
    hipStream_t sw, sr;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sw, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&sr, hipStreamNonBlocking, hi));
    hipEvent_t a, b, c;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipEventCreate(&c));
    auto fill = [&] { hipLaunchKernelGGL(k_fill, dim3((unsigned)((n16 + 3071) / 3072)), dim3(384), 0, sw, dw, n16); };
    for (int warm = 0; warm < 2; warm++) fill();
    CK(hipDeviceSynchronize());
    printf("%-44s %8s %9s %9s %10s\n", "reader", "read GB", "fill ms", "both ms", "total TB/s");
    const double reads_gb[] = {0, 0.3, 0.6, 1.35, 2.7, 5.4, 10.8, 32.2};
    for (int mode = 0; mode < 5; mode++) {
        for (double rgb : reads_gb) {
            if (mode > 0 && rgb == 0) continue;
            if (mode == 0 && rgb != 0) continue;
            uint64_t r16 = (uint64_t)(rgb * 1e9 / 16);
            // a gather piece p reads 16 bytes at uint4 index (p / 3) * 192 + p % 3 < 64 p + 3: the source holds rcap16 pieces
            const uint64_t gcap16 = rcap16 / 64 - 1024;
            if ((mode == 2 || mode == 3) && r16 > gcap16) r16 = gcap16;
            const unsigned rgrid = (unsigned)((r16 + 2047) / 2048);
            auto read = [&] {
                if (!r16) return;
                if (mode == 1) hipLaunchKernelGGL((k_read<0, 1>), dim3(rgrid), dim3(256), 0, sr, dr, r16, sink);
                if (mode == 2) hipLaunchKernelGGL((k_read<1, 1>), dim3(rgrid), dim3(256), 0, sr, dr, r16, sink);
                if (mode == 3) hipLaunchKernelGGL((k_read<1, 8>), dim3(rgrid), dim3(256), 0, sr, dr, r16, sink);
                if (mode == 4) hipLaunchKernelGGL((k_read<0, 8>), dim3(rgrid), dim3(256), 0, sr, dr, r16, sink);
            };
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a, sw));
            CK(hipStreamWaitEvent(sr, a, 0));
            for (int i = 0; i < 4; i++) {
                fill();
                read();
            }
            CK(hipEventRecord(b, sw));
            CK(hipEventRecord(c, sr));
            CK(hipStreamWaitEvent(sw, c, 0));
            CK(hipEventRecord(c, sw));
            CK(hipEventSynchronize(c));
            float fms = 0, tms = 0;
            CK(hipEventElapsedTime(&fms, a, b));
            CK(hipEventElapsedTime(&tms, a, c));
            const char* names[] = {"fill alone", "sequential read, 1 load in flight per thread", "48-byte gathers 3 KiB apart, 1 in flight", "48-byte gathers 3 KiB apart, 8 in flight",
                                   "sequential read, 8 loads in flight per thread"};
            if ((mode == 2 || mode == 3) && rgb > 2.8) continue;  // the 176 GB source bounds a gathered read at 2.75 GB
            const double rb = (double)r16 * 16;
            printf("%-44s %8.2f %9.3f %9.3f %10.2f\n", names[mode], rb / 1e9, fms / 4, tms / 4, (wbytes + rb) / (tms / 4 * 1e-3) / 1e12);
            fflush(stdout);
        }
    }
    return 0;
}
