/* Torch-free benchmark of the grouped engine through the C ABI (include/blsw.h): the same loop as bench.py — K steps of n
 * instances into a ring of output tensors — from a plain C program with hipMalloc'ed buffers. Prints one JSON line.
 *
 *   make -C tools engine_bench && tools/engine_bench [steps 96] [warmup 32] [n 1024] [coalesce 10] [buffers 3] [outputs 2]
 *
 * Inputs: n distinct (sk, msg) pairs signed on the GPU by blsw_sign_batch, every 16th message tampered after signing. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <hip/hip_runtime_api.h>
#include "blsw.h"

#define CHECK(x)                                                              \
    do {                                                                      \
        int rc_ = (int)(x);                                                   \
        if (rc_) {                                                            \
            fprintf(stderr, "%s failed: %d (line %d)\n", #x, rc_, __LINE__); \
            return 10;                                                        \
        }                                                                     \
    } while (0)

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char** argv) {
    const unsigned steps = argc > 1 ? atoi(argv[1]) : 96, warmup = argc > 2 ? atoi(argv[2]) : 32;
    const uint64_t n = argc > 3 ? strtoull(argv[3], NULL, 10) : 1024;
    const unsigned coalesce = argc > 4 ? atoi(argv[4]) : 10, buffers = argc > 5 ? atoi(argv[5]) : 3, n_out = argc > 6 ? atoi(argv[6]) : 2;
    blsw_layout_t L;
    CHECK(blsw_layout(32, &L));
    /* inputs */
    uint8_t *h_sk = calloc(n, 32), *h_msg = calloc(n, 32);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (uint64_t i = 0; i < n; i++) {
        for (int k = 0; k < 31; k++) { /* sk < 2^248 < r */
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            h_sk[i * 32 + k] = (uint8_t)x | (k == 0);
        }
        for (int k = 0; k < 32; k++) {
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            h_msg[i * 32 + k] = (uint8_t)(x >> 8);
        }
    }
    uint8_t *d_sk, *d_msg;
    uint64_t *d_pk_xy, *d_sig_xy;
    int32_t* d_st;
    void* d_sws;
    uint64_t sws = 0;
    CHECK(blsw_hash_to_g2_workspace_bytes(n, 32, &sws));
    CHECK(hipMalloc((void**)&d_sk, n * 32) || hipMalloc((void**)&d_msg, n * 32) || hipMalloc((void**)&d_pk_xy, n * 96) || hipMalloc((void**)&d_sig_xy, n * 192) ||
          hipMalloc((void**)&d_st, n * 4) || hipMalloc(&d_sws, sws));
    CHECK(hipMemcpy(d_sk, h_sk, n * 32, hipMemcpyHostToDevice) || hipMemcpy(d_msg, h_msg, n * 32, hipMemcpyHostToDevice));
    CHECK(blsw_sign_batch(d_sk, d_msg, 32, n, NULL, d_sig_xy, NULL, d_pk_xy, d_st, d_sws, sws, NULL));
    CHECK(hipDeviceSynchronize());
    for (uint64_t i = 0; i < n; i += 16) h_msg[i * 32 + 31] ^= 1; /* tampered: expected result 0 */
    CHECK(hipMemcpy(d_msg, h_msg, n * 32, hipMemcpyHostToDevice));
    CHECK(hipFree(d_sws));
    /* engine + output ring */
    uint64_t ws = 0;
    void* d_ws;
    CHECK(blsw_engine_workspace_bytes(n, 32, coalesce, buffers, &ws));
    CHECK(hipMalloc(&d_ws, ws));
    blsw_engine_t* e = NULL;
    CHECK(blsw_engine_create(&e, n, 32, coalesce, buffers, d_ws, ws));
    uint64_t* d_wit[16];
    int32_t* d_res[16];
    if (n_out < 1 || n_out > 16) return 2;
    for (unsigned k = 0; k < n_out; k++) CHECK(hipMalloc((void**)&d_wit[k], n * (uint64_t)L.n_witness * 48) || hipMalloc((void**)&d_res[k], n * 4));
    for (unsigned k = 0; k < warmup; k++) CHECK(blsw_engine_submit(e, d_pk_xy, d_sig_xy, d_msg, d_wit[k % n_out], L.n_witness, d_res[k % n_out], NULL));
    CHECK(blsw_engine_flush(e, NULL));
    CHECK(hipDeviceSynchronize());
    uint32_t cnt = 0;
    float avg = 0;
    CHECK(blsw_engine_expand_stats(e, &cnt, &avg));
    const double t0 = now_s();
    for (unsigned k = 0; k < steps; k++)
        CHECK(blsw_engine_submit(e, d_pk_xy, d_sig_xy, d_msg, d_wit[(warmup + k) % n_out], L.n_witness, d_res[(warmup + k) % n_out], NULL));
    CHECK(blsw_engine_flush(e, NULL));
    CHECK(hipDeviceSynchronize());
    const double dt = now_s() - t0;
    CHECK(blsw_engine_expand_stats(e, &cnt, &avg));
    int32_t* h_res = malloc(n * 4);
    CHECK(hipMemcpy(h_res, d_res[0], n * 4, hipMemcpyDeviceToHost));
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; i++) bad += h_res[i] != (i % 16 != 0);
    CHECK(blsw_engine_destroy(e));
    printf("{\"tool\": \"engine_bench (C ABI, no torch)\", \"value\": %.1f, \"unit\": \"instances/s\", \"steps\": %u, \"warmup\": %u, \"n\": %llu, \"coalesce\": %u, "
           "\"buffers\": %u, \"outputs\": %u, \"ms_per_step\": %.4f, \"expand_launches\": %u, \"expand_avg_ms\": %.4f, \"wrong_results\": %llu}\n",
           n * steps / dt, steps, warmup, (unsigned long long)n, coalesce, buffers, n_out, dt / steps * 1e3, cnt, avg, (unsigned long long)bad);
    return bad ? 3 : 0;
}
