// LAB PROGRAM: the product's own expansion kernels (k_stream.hip is included, launch_expand is called directly) timed in
// different launch contexts, to find what the pipeline does differently from tools/expand_lab.hip.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o build/expand_ctx_lab tools/expand_ctx_lab.hip && build/expand_ctx_lab
#include "../bls-verify-gadget_amd/csrc/k_stream.hip"
#include <vector>
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)
int main(int argc, char** argv) {
    const uint64_t n = 1024;
    blsw_layout_t L;
    make_layout(32, &L);
    const uint64_t sha_words = align_up((L.sha_bits + 31) / 32 + 1, BLSW_BITS_CHUNK_WORDS);
    uint64_t* d_wit[2];
    uint32_t* d_bits;
    for (int k = 0; k < 2; k++) CK(hipMalloc(&d_wit[k], n * (uint64_t)L.n_witness * 48));
    CK(hipMalloc(&d_bits, bits_tile_words(sha_words) * (n / 64) * 4));
    std::vector<uint32_t> hb(bits_tile_words(sha_words) * (n / 64));
    const double bytes = (double)n * L.sha_bits * 48;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s_nb, s_hi;
    CK(hipStreamCreateWithFlags(&s_nb, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi));
    hipEvent_t evs[64];
    for (auto& e : evs) CK(hipEventCreate(&e));
    auto run = [&](const char* name, uint32_t variant, hipStream_t st, bool events, bool ring, unsigned lds = 0) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        const int reps = 8;
        for (int i = -2; i < reps; i++) {
            if (i == 0) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(a, st));
            }
            ExpandArgs xa = {d_bits, sha_words, 0, L.sha_bits, L.off_expand, d_wit[ring ? (i & 1) : 0], L.n_witness, 1u, 0u, 0};
            if (events && i >= 0) CK(hipEventRecord(evs[2 * i], st));
            launch_expand(variant, 0, lds, st, xa, (unsigned)n);
            if (events && i >= 0) CK(hipEventRecord(evs[2 * i + 1], st));
        }
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("%-58s %7.3f ms  %7.1f GB/s\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
    };
    auto check = [&](uint32_t variant) {  // variant's output == variant 0's output, on a few instances (whole expand segment)
        ExpandArgs x0 = {d_bits, sha_words, 0, L.sha_bits, L.off_expand, d_wit[1], L.n_witness, 1u, 0u, 0};
        ExpandArgs x1 = x0;
        x1.d_witness = d_wit[0];
        CK(hipMemset(d_wit[0], 0xee, n * (uint64_t)L.n_witness * 48));
        launch_expand(0, 0, 0, 0, x0, (unsigned)n);
        launch_expand(variant, 0, 0, 0, x1, (unsigned)n);
        CK(hipDeviceSynchronize());
        const size_t seg = (size_t)L.sha_bits * 48, lead = 4096;
        std::vector<uint8_t> ha(seg + 2 * lead), hb2(seg + 2 * lead);
        bool ok = true;
        for (uint64_t i : {0ull, 1ull, 63ull, 64ull, 777ull, 1023ull}) {
            const size_t off = ((size_t)i * L.n_witness + L.off_expand) * 48 - lead;  // neighbours too: nothing may be written outside the segment
            CK(hipMemcpy(ha.data(), (uint8_t*)d_wit[1] + off, seg + 2 * lead, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hb2.data(), (uint8_t*)d_wit[0] + off, seg + 2 * lead, hipMemcpyDeviceToHost));
            if (memcmp(ha.data() + lead, hb2.data() + lead, seg)) ok = false;
            for (size_t k = 0; k < lead; k++) {
                if (i > 0 && i < 1023 && (hb2[k] != 0xee && hb2[k] != ha[k])) ok = false;
            }
        }
        printf("variant %u vs variant 0: %s\n", variant, ok ? "equal" : "DIFFERENT");
    };
    for (auto& w : hb) w = (uint32_t)rand() * 2654435761u;
    CK(hipMemcpy(d_bits, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    for (uint32_t v = 1; v <= 5; v++) check(v);
    for (int pat = 0; pat < 1; pat++) {
        for (auto& w : hb) w = pat == 0 ? (uint32_t)rand() * 2654435761u : (pat == 1 ? 0u : 0xffffffffu);
        CK(hipMemcpy(d_bits, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        printf("bits: %s\n", pat == 0 ? "random" : pat == 1 ? "all zero" : "all one");
        run("v1 null stream", 1, 0, false, false);
        run("v0 null stream", 0, 0, false, false);
        if (pat) continue;
        run("v1 non-blocking stream", 1, s_nb, false, false);
        run("v1 high-priority stream", 1, s_hi, false, false);
        run("v1 high-priority stream, events around each launch", 1, s_hi, true, false);
        run("v1 high-priority stream, events, ring of 2 outputs", 1, s_hi, true, true);
        run("v0 high-priority stream, events, ring of 2 outputs", 0, s_hi, true, true);
        for (uint32_t v = 2; v <= 5; v++) {
            char nm[64];
            snprintf(nm, sizeof nm, "v%u null stream", v);
            run(nm, v, 0, false, false);
        }
    }
    return 0;
}
