/* A C caller of the drop-in boundary (include/blsw.h), compiled with gcc against the header and linked to libblsw.so:
 * the role a Rust `extern "C"` shim plays in the reference crate. TEST PROGRAM.
 *
 *   caller layout                               host-only entry points (runs without a GPU): prints one line of key=value pairs
 *   caller verify <pk48 hex> <msg hex> <sig96 hex>   decode -> engine create/submit/flush -> result + witness digest (needs a GPU)
 *   caller stream <pk48 hex> <msg hex> <sig96 hex>   64 copies of the instance, 5 steps through a consumer-mode engine (groups of 2)
 *       with ONE witness tensor and ONE compact buffer as the whole output ring: steps alternate plain / compact submits, the
 *       consumer (digest kernel; expand_compact first for compact steps) releases each output before the next step may use it
 *   caller bytes <file>    every line "<pk48 hex> <msg32 hex> <sig96 hex>" is one instance of ONE batch through
 *       blsw_engine_submit_bytes (decode + status rule + gadget in one call): prints results=<0/1 per instance> statuses=<pk,sig;...>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "blsw.h"

static int unhex(const char* s, uint8_t* out, size_t n) {
    if (strlen(s) != 2 * n) return -1;
    for (size_t i = 0; i < n; i++) {
        unsigned v;
        if (sscanf(s + 2 * i, "%2x", &v) != 1) return -1;
        out[i] = (uint8_t)v;
    }
    return 0;
}
#define CHECK(x)                                                      \
    do {                                                              \
        int rc_ = (x);                                                \
        if (rc_) {                                                    \
            fprintf(stderr, "%s failed: %d (line %d)\n", #x, rc_, __LINE__); \
            return 10;                                                \
        }                                                             \
    } while (0)

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "layout")) {
        blsw_layout_t L, M;
        blsw_engine_options_t o;
        uint64_t ws = 0, ws_multi = 0;
        CHECK(blsw_layout(32, &L));
        CHECK(blsw_layout_multi(32, 128, &M));
        CHECK(blsw_engine_options_default(&o));
        CHECK(blsw_engine_workspace_bytes(1024, 32, 16, 3, &ws));
        CHECK(blsw_verify_multi_workspace_bytes(1, 32, 128, &ws_multi));
        blsw_matrices_info_t mi;
        CHECK(blsw_matrices_info(0, 0, 1, &mi)); /* host-only synthesis of the smallest shape (empty message) */
        if (mi.n_instance_vars != 1 || mi.n_constraints == 0 || mi.nnz[0] == 0) return 13;
        if (blsw_layout(32, NULL) != BLSW_ERR_ARG || blsw_engine_workspace_bytes(0, 32, 1, 1, &ws_multi) != BLSW_ERR_ARG) return 11;
        printf("version=%d sizeof_layout=%zu n_witness=%u sha_bits=%u off_expand=%u off_miller=%u multi_n_witness=%u multi_pairs=%u multi_stride_hash=%u "
               "workspace=%llu device=%d pairing_mode=%u prio_mode=%u matrices0_constraints=%llu matrices0_witness=%llu\n",
               blsw_version(), sizeof(blsw_layout_t), L.n_witness, L.sha_bits, L.off_expand, L.off_miller, M.n_witness, M.n_pairs, M.stride_hash,
               (unsigned long long)ws, o.device, o.pairing_mode, o.prio_mode, (unsigned long long)mi.n_constraints, (unsigned long long)mi.n_witness);
        return 0;
    }
    if (argc == 5 && !strcmp(argv[1], "verify")) {
        uint8_t pk[48], msg[32], sig[96];
        if (unhex(argv[2], pk, 48) || unhex(argv[3], msg, 32) || unhex(argv[4], sig, 96)) return 12;
        blsw_layout_t L;
        CHECK(blsw_layout(32, &L));
        uint8_t *d_pk, *d_sig, *d_msg;
        uint64_t *d_pk_xy, *d_sig_xy, *d_wit, *d_dig;
        int32_t *d_st, *d_res;
        void* d_ws;
        uint64_t ws = 0;
        CHECK(blsw_engine_workspace_bytes(1, 32, 1, 1, &ws));
        CHECK(hipMalloc((void**)&d_pk, 48) || hipMalloc((void**)&d_sig, 96) || hipMalloc((void**)&d_msg, 32) || hipMalloc((void**)&d_pk_xy, 96) ||
              hipMalloc((void**)&d_sig_xy, 192) || hipMalloc((void**)&d_st, 8) || hipMalloc((void**)&d_res, 4) || hipMalloc((void**)&d_dig, 16) ||
              hipMalloc((void**)&d_wit, (size_t)L.n_witness * 48) || hipMalloc(&d_ws, ws));
        CHECK(hipMemcpy(d_pk, pk, 48, hipMemcpyHostToDevice) || hipMemcpy(d_sig, sig, 96, hipMemcpyHostToDevice) || hipMemcpy(d_msg, msg, 32, hipMemcpyHostToDevice));
        CHECK(blsw_decode_batch(d_pk, d_sig, 1, d_pk_xy, d_sig_xy, d_st, NULL));
        blsw_engine_t* e = NULL;
        CHECK(blsw_engine_create(&e, 1, 32, 1, 1, d_ws, ws));
        CHECK(blsw_engine_submit(e, d_pk_xy, d_sig_xy, d_msg, d_wit, L.n_witness, d_res, NULL));
        uint64_t sub = 0, lau = 0;
        CHECK(blsw_engine_submitted(e, &sub) || blsw_engine_launched(e, &lau));
        CHECK(blsw_engine_flush(e, NULL));
        CHECK(blsw_engine_wait_step(e, 0, NULL));
        CHECK(blsw_witness_digest(d_wit, L.n_witness, 1, L.n_witness, d_dig, NULL));
        CHECK(hipDeviceSynchronize());
        int32_t st[2], res;
        uint64_t dig[2];
        CHECK(hipMemcpy(st, d_st, 8, hipMemcpyDeviceToHost) || hipMemcpy(&res, d_res, 4, hipMemcpyDeviceToHost) || hipMemcpy(dig, d_dig, 16, hipMemcpyDeviceToHost));
        CHECK(blsw_engine_destroy(e));
        printf("status_pk=%d status_sig=%d result=%d submitted=%llu digest0=%llu digest1=%llu n_witness=%u\n", st[0], st[1], res, (unsigned long long)sub,
               (unsigned long long)dig[0], (unsigned long long)dig[1], L.n_witness);
        return 0;
    }
    if (argc == 5 && !strcmp(argv[1], "stream")) {
        enum { N = 64, STEPS = 5 };
        uint8_t pk[48], msg[32], sig[96], *h_pk = malloc(N * 48), *h_msg = malloc(N * 32), *h_sig = malloc(N * 96);
        if (unhex(argv[2], pk, 48) || unhex(argv[3], msg, 32) || unhex(argv[4], sig, 96)) return 12;
        for (int i = 0; i < N; i++) memcpy(h_pk + 48 * i, pk, 48), memcpy(h_msg + 32 * i, msg, 32), memcpy(h_sig + 96 * i, sig, 96);
        blsw_layout_t L;
        CHECK(blsw_layout(32, &L));
        blsw_engine_options_t o;
        CHECK(blsw_engine_options_default(&o));
        o.consumer_mode = 1;
        uint8_t *d_pk, *d_sig, *d_msg;
        uint64_t *d_pk_xy, *d_sig_xy, *d_wit, *d_scratch, *d_dig;
        int32_t *d_st, *d_res;
        void *d_ws, *d_compact;
        uint64_t ws = 0, cb = 0;
        CHECK(blsw_engine_workspace_bytes_ex(N, 32, 2, 2, &o, &ws));
        CHECK(hipMalloc((void**)&d_pk, N * 48) || hipMalloc((void**)&d_sig, N * 96) || hipMalloc((void**)&d_msg, N * 32) || hipMalloc((void**)&d_pk_xy, N * 96) ||
              hipMalloc((void**)&d_sig_xy, N * 192) || hipMalloc((void**)&d_st, N * 8) || hipMalloc((void**)&d_res, STEPS * N * 4) ||
              hipMalloc((void**)&d_dig, STEPS * N * 16) || hipMalloc((void**)&d_wit, (size_t)N * L.n_witness * 48) ||
              hipMalloc((void**)&d_scratch, (size_t)N * L.n_witness * 48) || hipMalloc(&d_ws, ws));
        CHECK(hipMemcpy(d_pk, h_pk, N * 48, hipMemcpyHostToDevice) || hipMemcpy(d_sig, h_sig, N * 96, hipMemcpyHostToDevice) || hipMemcpy(d_msg, h_msg, N * 32, hipMemcpyHostToDevice));
        CHECK(blsw_decode_batch(d_pk, d_sig, N, d_pk_xy, d_sig_xy, d_st, NULL));
        blsw_engine_t* e = NULL;
        CHECK(blsw_engine_create_ex(&e, N, 32, 2, 2, &o, d_ws, ws));
        CHECK(blsw_engine_compact_bytes(e, &cb));
        CHECK(hipMalloc(&d_compact, cb));
        hipStream_t consumer;
        CHECK(hipStreamCreateWithFlags(&consumer, hipStreamNonBlocking));
        uint64_t next = 0, mat = 0;
        int busy = 0;
        for (int k = 0; k <= STEPS; k++) { /* k == STEPS: flush and drain the rest */
            for (;;) {
                int rc = k == STEPS ? blsw_engine_flush(e, NULL)
                         : (k & 1) ? blsw_engine_submit_compact(e, d_pk_xy, d_sig_xy, d_msg, d_compact, d_res + k * N, NULL)
                                   : blsw_engine_submit(e, d_pk_xy, d_sig_xy, d_msg, d_wit, L.n_witness, d_res + k * N, NULL);
                if (rc != BLSW_OK && rc != BLSW_ERR_BUSY) return 20 + rc;
                busy += rc == BLSW_ERR_BUSY;
                /* drain: every materialised step is consumed and its output released (which lets the next user of it go) */
                for (;;) {
                    CHECK(blsw_engine_materialised(e, &mat));
                    if (next >= mat) break;
                    CHECK(blsw_engine_wait_step(e, next, consumer));
                    const uint64_t* vec = d_wit;
                    if (next & 1) {
                        CHECK(blsw_engine_expand_compact(e, d_compact, d_scratch, L.n_witness, consumer));
                        vec = d_scratch;
                    }
                    CHECK(blsw_witness_digest(vec, L.n_witness, N, L.n_witness, d_dig + next * N * 2, consumer));
                    CHECK(blsw_engine_output_consumed(e, (next & 1) ? d_compact : (void*)d_wit, consumer));
                    next++;
                }
                if (rc == BLSW_OK) break;
            }
        }
        if (next != STEPS) return 14;
        CHECK(hipDeviceSynchronize());
        uint64_t* dig = malloc(STEPS * N * 16);
        int32_t* res = malloc(STEPS * N * 4);
        CHECK(hipMemcpy(dig, d_dig, STEPS * N * 16, hipMemcpyDeviceToHost) || hipMemcpy(res, d_res, STEPS * N * 4, hipMemcpyDeviceToHost));
        int same = 1, ok = 1;
        for (int i = 0; i < STEPS * N; i++) same &= dig[2 * i] == dig[0] && dig[2 * i + 1] == dig[1], ok &= res[i] == res[0];
        CHECK(blsw_engine_destroy(e));
        printf("steps=%d n=%d compact_bytes=%llu all_digests_equal=%d all_results_equal=%d result=%d busy_returns=%d digest0=%llu digest1=%llu\n", STEPS, N,
               (unsigned long long)cb, same, ok, res[0], busy, (unsigned long long)dig[0], (unsigned long long)dig[1]);
        return 0;
    }
    if (argc == 3 && !strcmp(argv[1], "bytes")) {
        FILE* f = fopen(argv[2], "r");
        if (!f) return 15;
        enum { MAXN = 256 };
        static uint8_t h_pk[MAXN * 48], h_msg[MAXN * 32], h_sig[MAXN * 96];
        static char a[128], b[128], c[256];
        int n = 0;
        while (n < MAXN && fscanf(f, "%127s %127s %255s", a, b, c) == 3) {
            if (unhex(a, h_pk + 48 * n, 48) || unhex(b, h_msg + 32 * n, 32) || unhex(c, h_sig + 96 * n, 96)) return 12;
            n++;
        }
        fclose(f);
        if (n == 0) return 16;
        blsw_layout_t L;
        CHECK(blsw_layout(32, &L));
        uint8_t *d_pk, *d_sig, *d_msg;
        uint64_t *d_pk_xy, *d_sig_xy;
        int32_t *d_st, *d_res;
        void* d_ws;
        uint64_t ws = 0;
        CHECK(blsw_engine_workspace_bytes((uint64_t)n, 32, 1, 1, &ws));
        CHECK(hipMalloc((void**)&d_pk, n * 48) || hipMalloc((void**)&d_sig, n * 96) || hipMalloc((void**)&d_msg, n * 32) || hipMalloc((void**)&d_pk_xy, n * 96) ||
              hipMalloc((void**)&d_sig_xy, n * 192) || hipMalloc((void**)&d_st, n * 8) || hipMalloc((void**)&d_res, n * 4) || hipMalloc(&d_ws, ws));
        CHECK(hipMemcpy(d_pk, h_pk, n * 48, hipMemcpyHostToDevice) || hipMemcpy(d_sig, h_sig, n * 96, hipMemcpyHostToDevice) || hipMemcpy(d_msg, h_msg, n * 32, hipMemcpyHostToDevice));
        blsw_engine_t* e = NULL;
        CHECK(blsw_engine_create(&e, (uint64_t)n, 32, 1, 1, d_ws, ws));
        CHECK(blsw_engine_submit_bytes(e, d_pk, d_sig, d_msg, d_pk_xy, d_sig_xy, d_st, NULL, 0, d_res, NULL));
        CHECK(blsw_engine_flush(e, NULL));
        CHECK(hipDeviceSynchronize());
        static int32_t st[2 * MAXN], res[MAXN];
        CHECK(hipMemcpy(st, d_st, n * 8, hipMemcpyDeviceToHost) || hipMemcpy(res, d_res, n * 4, hipMemcpyDeviceToHost));
        CHECK(blsw_engine_destroy(e));
        printf("n=%d results=", n);
        for (int i = 0; i < n; i++) printf("%d", res[i]);
        printf(" statuses=");
        for (int i = 0; i < n; i++) printf("%d,%d;", st[2 * i], st[2 * i + 1]);
        printf("\n");
        return 0;
    }
    fprintf(stderr, "usage: caller layout | caller verify <pk48> <msg32> <sig96> | caller stream <pk48> <msg32> <sig96> (hex) | caller bytes <file>\n");
    return 2;
}
