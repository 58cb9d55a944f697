// LAB PROGRAM (round 4): the product's expansion kernels (k_stream.hip is included) on CU-masked streams — E compute units per XCD — alone and beside a
// long-running arithmetic kernel on the other compute units (a stand-in for the chain kernels: no memory traffic, 1 wave per SIMD).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o build/cu_expand_lab tools/cu_expand_lab.hip && build/cu_expand_lab
#include "../bls-verify-gadget_amd/csrc/k_stream.hip"
#include <vector>
using namespace blsw;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(64) void k_hog(uint32_t iters, uint32_t* out) {
    uint64_t a = threadIdx.x + 1, b = blockIdx.x + 3;
    for (uint32_t i = 0; i < iters; i++) {
        a = a * 0x9E3779B97F4A7C15ull + b;
        b = b * 0xC2B2AE3D27D4EB4Full + a;
    }
    if (a == 42 && b == 7) out[0] = 1;
}
int main() {
    const uint64_t n = 1024;
    blsw_layout_t L;
    make_layout(32, &L);
    const uint64_t sha_words = align_up((L.sha_bits + 31) / 32 + 1, BLSW_BITS_CHUNK_WORDS);
    uint64_t* d_wit;
    uint32_t *d_bits, *d_out;
    CK(hipMalloc(&d_wit, n * (uint64_t)L.n_witness * 48));
    CK(hipMalloc(&d_bits, bits_tile_words(sha_words) * (n / 64) * 4));
    CK(hipMalloc(&d_out, 4));
    std::vector<uint32_t> hb(bits_tile_words(sha_words) * (n / 64));
    for (auto& w : hb) w = (uint32_t)rand() * 2654435761u;
    CK(hipMemcpy(d_bits, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    const double bytes = (double)n * L.sha_bits * 48;
    for (int E : {32, 16, 12, 8}) {
        uint32_t ms_[8] = {0}, mc_[8] = {0};
        for (int i = 0; i < 256; i++) ((i / 8 < E) ? ms_ : mc_)[i / 32] |= 1u << (i % 32);
        hipStream_t st, sc = nullptr;
        CK(hipExtStreamCreateWithCUMask(&st, 8, ms_));
        if (E < 32) CK(hipExtStreamCreateWithCUMask(&sc, 8, mc_));
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        for (int hog = 0; hog < (E < 32 ? 2 : 1); hog++) {
            for (uint32_t variant : {0u, 5u, 10u, 11u, 12u}) {
                ExpandArgs xa = {d_bits, sha_words, 0, L.sha_bits, L.off_expand, d_wit, L.n_witness, 1u, 0u, 0};
                launch_expand(variant, 0, 0, st, xa, (unsigned)n);
                CK(hipDeviceSynchronize());
                if (hog) hipLaunchKernelGGL(k_hog, dim3((32 - E) * 8 * 4), dim3(64), 0, sc, 6000000u, d_out);  // ~1 wave per SIMD of the other partition, tens of ms
                CK(hipEventRecord(a, st));
                for (int i = 0; i < 5; i++) launch_expand(variant, 0, 0, st, xa, (unsigned)n);
                CK(hipEventRecord(b, st));
                CK(hipEventSynchronize(b));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, a, b));
                printf("E = %2d CUs per XCD  %-28s expand_variant %2u  %7.3f ms  %7.1f GB/s\n", E, hog ? "beside an arithmetic kernel" : "alone", variant, ms / 5, bytes / (ms / 5 * 1e-3) / 1e9);
                fflush(stdout);
                CK(hipDeviceSynchronize());
            }
        }
    }
    return 0;
}
