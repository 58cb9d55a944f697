// libblsw.so — HIP kernels (gfx950) and the C ABI of include/blsw.h.
// One BLS-verify instance per lane for the field/curve/pairing chains (integer VALU work, no MFMA),
// plus a streaming bit->Fp expansion kernel that writes the ~655k boolean witnesses of the in-circuit
// SHA-256 (93 % of the witness bytes) at HBM-write speed. See DESIGN.md for the data layout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "chains.cuh"
#include "layout.h"

using namespace blsw;

namespace {

// ---------------------------------------------------------------- workspace
// All per-instance scratch is stored element-major: element e of instance i lives at index e*n + i, so that the
// 64 lanes of a wave touch one contiguous 3 KiB window per element (48 B per lane).
struct Workspace {
    uint32_t* bits;  // [sha_words][n] u32 : SHA witness bitstream, word-major
    Fp* u;           // [4][n]   hash_to_field output u0.c0,u0.c1,u1.c0,u1.c1
    Fp* q;           // [12][n]  Q0 (x.c0,x.c1,y.c0,y.c1,z.c0,z.c1), Q1
    Fp* h;           // [6][n]   H(m) projective
    Fp* pkaff;       // [2][n]   prepare_g1(pk)
    Fp* coeff;       // [2][272][n]  line coefficients: 0 = H(m), 1 = sig
    uint64_t sha_words;
    uint64_t total_bytes;
};
inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
Workspace carve(void* base, uint64_t n, const blsw_layout_t& L) {
    Workspace w;
    w.sha_words = (L.sha_bits + 31) / 32 + 1;
    uint64_t off = 0;
    auto take = [&](uint64_t bytes) {
        uint64_t o = off;
        off = align_up(off + bytes, 256);
        return reinterpret_cast<char*>(base) + o;
    };
    w.bits = reinterpret_cast<uint32_t*>(take(w.sha_words * n * 4));
    w.u = reinterpret_cast<Fp*>(take(4 * n * sizeof(Fp)));
    w.q = reinterpret_cast<Fp*>(take(12 * n * sizeof(Fp)));
    w.h = reinterpret_cast<Fp*>(take(6 * n * sizeof(Fp)));
    w.pkaff = reinterpret_cast<Fp*>(take(2 * n * sizeof(Fp)));
    w.coeff = reinterpret_cast<Fp*>(take(2ull * 272 * n * sizeof(Fp)));
    w.total_bytes = off;
    return w;
}

__device__ __forceinline__ Fp ld_fp(const Fp* p) {
    const uint4* s = reinterpret_cast<const uint4*>(p);
    uint4 a = s[0], b = s[1], c = s[2];
    Fp r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    return r;
}
__device__ __forceinline__ void st_fp(Fp* p, const Fp& v) {
    uint4* d = reinterpret_cast<uint4*>(p);
    d[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    d[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    d[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
}
__device__ __forceinline__ Fp2 ld_fp2(const Fp* p, uint64_t n) { return {ld_fp(p), ld_fp(p + n)}; }
__device__ __forceinline__ uint32_t* wit_base(uint64_t* d_witness, uint64_t stride, uint64_t i) {
    return d_witness ? reinterpret_cast<uint32_t*>(d_witness + i * stride * 6) : nullptr;
}

// ---------------------------------------------------------------- kernels (one instance per lane)
__global__ __launch_bounds__(64) void k_sha(const uint8_t* __restrict__ msgs, uint32_t msg_len, uint64_t n, blsw_layout_t L, Workspace ws,
                                            uint64_t* d_witness, uint64_t stride, int want_bits, int write_u) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* msg = msgs + i * msg_len;
    // UInt8::new_witness_vec(msg): 8 booleans per byte, little-endian
    Emitter em = {wit_base(d_witness, stride, i), L.off_msg};
    for (uint32_t k = 0; k < msg_len; k++) {
        uint32_t b = msg[k];
        for (int j = 0; j < 8; j++) em.put_bool((b >> j) & 1);
    }
    BitSink s;
    s.init(want_bits ? ws.bits + i : nullptr, n);
    uint32_t uw[64];
    expand_message_w(s, msg, msg_len, false, uw);
    if (write_u)
        for (int j = 0; j < 4; j++) st_fp(ws.u + (uint64_t)j * n + i, hash_to_field_elem(uw + 16 * j));
}

// value-only expand_message + hash_to_field: hands u0, u1 to k_map without waiting for the witness-bit pass
__global__ __launch_bounds__(64) void k_sha_values(const uint8_t* __restrict__ msgs, uint32_t msg_len, uint64_t n, Workspace ws) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t uw[64];
    expand_message_values(msgs + i * msg_len, msg_len, uw);
    for (int j = 0; j < 4; j++) st_fp(ws.u + (uint64_t)j * n + i, hash_to_field_elem(uw + 16 * j));
}

// bitstream -> Fp elements: element e of the expand segment = bit ? R mod p : 0. One 16-byte chunk per thread
// per step, consecutive threads write consecutive 16 B: every store instruction covers 1 KiB contiguous per wave.
__global__ __launch_bounds__(256) void k_sha_expand(const uint32_t* __restrict__ bits, uint64_t n, uint32_t sha_bits, uint32_t off_expand,
                                                    uint64_t* __restrict__ d_witness, uint64_t stride) {
    constexpr uint32_t R1[12] = BLSW_R1_LIMBS;
    const uint64_t inst = blockIdx.y;
    const uint32_t nchunks = sha_bits * 3;
    uint4* out = reinterpret_cast<uint4*>(d_witness + (inst * stride + off_expand) * 6);
    const uint32_t* b = bits + inst;
    uint32_t q0 = blockIdx.x * (256 * 16) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t q = q0 + k * 256;
        if (q < nchunks) {
            uint32_t e = q / 3, c = q - e * 3;
            uint32_t w = b[(uint64_t)(e >> 5) * n];
            uint32_t m = 0u - ((w >> (e & 31)) & 1u);
            uint4 v;
            v.x = (c == 0 ? R1[0] : (c == 1 ? R1[4] : R1[8])) & m;
            v.y = (c == 0 ? R1[1] : (c == 1 ? R1[5] : R1[9])) & m;
            v.z = (c == 0 ? R1[2] : (c == 1 ? R1[6] : R1[10])) & m;
            v.w = (c == 0 ? R1[3] : (c == 1 ? R1[7] : R1[11])) & m;
            out[q] = v;
        }
    }
}

__global__ __launch_bounds__(64) void k_g1(const uint64_t* __restrict__ pk_xy, uint64_t n, blsw_layout_t L, Workspace ws, uint64_t* d_witness,
                                           uint64_t stride) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fp* p = reinterpret_cast<const Fp*>(pk_xy + i * 12);
    uint32_t* base = wit_base(d_witness, stride, i);
    G1ChainOut o = chain_g1_alloc({base, L.off_pk_alloc}, {base, L.off_pk_not_zero}, {base, L.off_prep_pk}, ld_fp(p), ld_fp(p + 1));
    st_fp(ws.pkaff + i, o.ax);
    st_fp(ws.pkaff + n + i, o.ay);
}

__global__ __launch_bounds__(64) void k_g2_alloc(const uint64_t* __restrict__ sig_xy, uint64_t n, blsw_layout_t L, uint64_t* d_witness, uint64_t stride) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fp* p = reinterpret_cast<const Fp*>(sig_xy + i * 24);
    Fp2 sx = {ld_fp(p), ld_fp(p + 1)}, sy = {ld_fp(p + 2), ld_fp(p + 3)};
    chain_g2_alloc({wit_base(d_witness, stride, i), L.off_sig_alloc}, sx, sy);
}

// lanes [0, n): u0 -> Q0 ; lanes [n, 2n): u1 -> Q1
__global__ __launch_bounds__(64) void k_map(uint64_t n, blsw_layout_t L, Workspace ws, uint64_t* d_witness, uint64_t stride) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n) return;
    uint32_t which = t >= n;
    uint64_t i = which ? t - n : t;
    Fp2 u = ld_fp2(ws.u + (uint64_t)(2 * which) * n + i, n);
    Proj<OpsFp2> q = chain_map_to_curve({wit_base(d_witness, stride, i), which ? L.off_map1 : L.off_map0}, u);
    Fp* o = ws.q + (uint64_t)(6 * which) * n + i;
    st_fp(o, q.x.c0);
    st_fp(o + n, q.x.c1);
    st_fp(o + 2 * n, q.y.c0);
    st_fp(o + 3 * n, q.y.c1);
    st_fp(o + 4 * n, q.z.c0);
    st_fp(o + 5 * n, q.z.c1);
}

__device__ __forceinline__ Proj<OpsFp2> ld_proj2(const Fp* p, uint64_t n) {
    Proj<OpsFp2> r;
    r.x = ld_fp2(p, n);
    r.y = ld_fp2(p + 2 * n, n);
    r.z = ld_fp2(p + 4 * n, n);
    return r;
}
__global__ __launch_bounds__(64) void k_cofactor(uint64_t n, blsw_layout_t L, Workspace ws, uint64_t* d_witness, uint64_t stride) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Proj<OpsFp2> q0 = ld_proj2(ws.q + i, n), q1 = ld_proj2(ws.q + 6 * n + i, n);
    uint32_t* base = wit_base(d_witness, stride, i);
    Proj<OpsFp2> h = chain_cofactor({base, L.off_add}, {base, L.off_cofactor}, q0, q1);
    Fp* o = ws.h + i;
    st_fp(o, h.x.c0);
    st_fp(o + n, h.x.c1);
    st_fp(o + 2 * n, h.y.c0);
    st_fp(o + 3 * n, h.y.c1);
    st_fp(o + 4 * n, h.z.c0);
    st_fp(o + 5 * n, h.z.c1);
}

// line coefficients, element-major: coefficient idx of instance i at p[idx * n]
struct CoeffStrided {
    Fp* p;
    uint64_t n;
    __device__ __forceinline__ void st(uint32_t idx, const Fp& v) const { st_fp(p + (uint64_t)idx * n, v); }
    __device__ __forceinline__ Fp ld(uint32_t idx) const { return ld_fp(p + (uint64_t)idx * n); }
};
// which = 0: prepare_g2(H(m)) ; which = 1: prepare_g2(sig). Lanes [0, n) take `which_first`, [n, 2n) the next one.
__global__ __launch_bounds__(64) void k_prepare(const uint64_t* __restrict__ sig_xy, uint64_t n, blsw_layout_t L, Workspace ws, uint64_t* d_witness,
                                                uint64_t stride, int which_first, int which_count) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)which_count * n) return;
    uint32_t which = which_first + (uint32_t)(t / n);
    uint64_t i = t % n;
    Proj<OpsFp2> q;
    if (which == 0) {
        q = ld_proj2(ws.h + i, n);
    } else {
        const Fp* p = reinterpret_cast<const Fp*>(sig_xy + i * 24);
        Fp2 sx = {ld_fp(p), ld_fp(p + 1)}, sy = {ld_fp(p + 2), ld_fp(p + 3)};
        bool inf = fp2_is_zero(sx) && fp2_is_zero(sy);
        q.x = inf ? fp2_zero() : sx;
        q.y = inf ? fp2_one() : sy;
        q.z = inf ? fp2_zero() : fp2_one();
    }
    CoeffStrided out = {ws.coeff + (uint64_t)which * 272 * n + i, n};
    chain_prepare_g2({wit_base(d_witness, stride, i), which == 0 ? L.off_prep_h : L.off_prep_sig}, q, out);
}

// Miller loop + final exponentiation + is_one
__global__ __launch_bounds__(64) void k_pairing(uint64_t n, blsw_layout_t L, Workspace ws, uint64_t* d_witness, uint64_t stride, int32_t* d_result) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* base = wit_base(d_witness, stride, i);
    Fp pkx = ld_fp(ws.pkaff + i), pky = ld_fp(ws.pkaff + n + i);
    CoeffStrided ch = {ws.coeff + i, n};
    CoeffStrided cs = {ws.coeff + 272ull * n + i, n};
    Fp12 f = chain_miller({base, L.off_miller}, pkx, pky, cs, ch);
    bool res = chain_final_exp_is_one({base, L.off_final_exp}, {base, L.off_is_one}, f);
    if (d_result) d_result[i] = res ? 1 : 0;
}

// H(m) projective -> affine (hash_to_g2 batch output)
__global__ __launch_bounds__(64) void k_h_to_affine(uint64_t n, Workspace ws, uint64_t* d_out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Proj<OpsFp2> h = ld_proj2(ws.h + i, n);
    Fp2 zi = fp2_inv(h.z);
    Fp2 x = fp2_mul(h.x, zi), y = fp2_mul(h.y, zi);
    Fp* o = reinterpret_cast<Fp*>(d_out + i * 24);
    st_fp(o, x.c0);
    st_fp(o + 1, x.c1);
    st_fp(o + 2, y.c0);
    st_fp(o + 3, y.c1);
}

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

inline int hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) {
        fprintf(stderr, "[blsw] %s: %s\n", what, hipGetErrorString(e));
        return BLSW_ERR_HIP;
    }
    return BLSW_OK;
}

}  // namespace

// Execution context: auxiliary streams and events so that the independent chains of one batch overlap.
//   main : sha_values -> map -> cofactor -> prepare(H) ............ -> pairing -> join
//   aux0 : g1_alloc, g2_alloc                                        (needs only pk / sig)
//   aux1 : prepare(sig)                                              (needs only sig)
//   aux2 : sha witness bits -> sha_expand (the HBM-bound stream of ~31 MB / instance)
struct blsw_ctx {
    hipStream_t aux[3];
    hipEvent_t ev_start, ev_aux[3];
    hipEvent_t ev_exp0, ev_exp1;  // around k_sha_expand, for the live roofline measurement
    int have_expand_timing;
};

extern "C" {

int blsw_version(void) { return 2; }

int blsw_ctx_create(blsw_ctx_t** out) {
    if (!out) return BLSW_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return BLSW_ERR_NO_DEVICE;
    blsw_ctx* c = new blsw_ctx();
    for (int i = 0; i < 3; i++) {
        if (hip_ok(hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking), "stream create")) return BLSW_ERR_HIP;
        if (hip_ok(hipEventCreateWithFlags(&c->ev_aux[i], hipEventDisableTiming), "event create")) return BLSW_ERR_HIP;
    }
    if (hip_ok(hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming), "event create")) return BLSW_ERR_HIP;
    if (hip_ok(hipEventCreate(&c->ev_exp0), "event create") || hip_ok(hipEventCreate(&c->ev_exp1), "event create")) return BLSW_ERR_HIP;
    c->have_expand_timing = 0;
    *out = c;
    return BLSW_OK;
}
int blsw_ctx_destroy(blsw_ctx_t* c) {
    if (!c) return BLSW_ERR_ARG;
    for (int i = 0; i < 3; i++) {
        hipStreamDestroy(c->aux[i]);
        hipEventDestroy(c->ev_aux[i]);
    }
    hipEventDestroy(c->ev_start);
    hipEventDestroy(c->ev_exp0);
    hipEventDestroy(c->ev_exp1);
    delete c;
    return BLSW_OK;
}
// duration of the last k_sha_expand launch issued through this context (blocks until it has finished)
int blsw_ctx_last_expand_ms(blsw_ctx_t* c, float* ms) {
    if (!c || !ms || !c->have_expand_timing) return BLSW_ERR_ARG;
    if (hip_ok(hipEventSynchronize(c->ev_exp1), "event sync")) return BLSW_ERR_HIP;
    return hip_ok(hipEventElapsedTime(ms, c->ev_exp0, c->ev_exp1), "event elapsed");
}

int blsw_layout(uint32_t msg_len, blsw_layout_t* out) {
    if (!out || msg_len > 65535) return BLSW_ERR_ARG;
    make_layout(msg_len, out);
    return BLSW_OK;
}

int blsw_workspace_bytes(uint64_t n, uint32_t msg_len, uint64_t* bytes) {
    if (!bytes || n == 0) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    Workspace w = carve(nullptr, n, L);
    *bytes = w.total_bytes;
    return BLSW_OK;
}

int blsw_witness_batch(blsw_ctx_t* c, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, uint32_t msg_len, uint64_t n,
                       uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, void* d_workspace, uint64_t workspace_bytes, void* stream_) {
    if (!c || !d_pk_xy || !d_sig_xy || (!d_msg && msg_len) || n == 0 || !d_workspace) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    if (d_witness && witness_stride < L.n_witness) return BLSW_ERR_ARG;
    Workspace ws = carve(d_workspace, n, L);
    if (ws.total_bytes > workspace_bytes) return BLSW_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64);
    // fork
    hipEventRecord(c->ev_start, st);
    for (int i = 0; i < 3; i++) hipStreamWaitEvent(c->aux[i], c->ev_start, 0);
    // aux0: group allocations
    hipLaunchKernelGGL(k_g1, dim3(g1), dim3(64), 0, c->aux[0], d_pk_xy, n, L, ws, d_witness, witness_stride);
    hipLaunchKernelGGL(k_g2_alloc, dim3(g1), dim3(64), 0, c->aux[0], d_sig_xy, n, L, d_witness, witness_stride);
    hipEventRecord(c->ev_aux[0], c->aux[0]);
    // aux1: prepare_g2(sig)
    hipLaunchKernelGGL(k_prepare, dim3(g1), dim3(64), 0, c->aux[1], d_sig_xy, n, L, ws, d_witness, witness_stride, 1, 1);
    hipEventRecord(c->ev_aux[1], c->aux[1]);
    // aux2: SHA-256 witness bits and their expansion (only when witnesses are requested)
    if (d_witness) {
        hipLaunchKernelGGL(k_sha, dim3(g1), dim3(64), 0, c->aux[2], d_msg, msg_len, n, L, ws, d_witness, witness_stride, 1, 0);
        dim3 grid((L.sha_bits * 3 + 4095) / 4096, (unsigned)n);
        hipEventRecord(c->ev_exp0, c->aux[2]);
        hipLaunchKernelGGL(k_sha_expand, grid, dim3(256), 0, c->aux[2], ws.bits, n, L.sha_bits, L.off_expand, d_witness, witness_stride);
        hipEventRecord(c->ev_exp1, c->aux[2]);
        c->have_expand_timing = 1;
    }
    hipEventRecord(c->ev_aux[2], c->aux[2]);
    // main: the hash-to-G2 critical path
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, d_msg, msg_len, n, ws);
    hipLaunchKernelGGL(k_map, dim3(g2), dim3(64), 0, st, n, L, ws, d_witness, witness_stride);
    hipLaunchKernelGGL(k_cofactor, dim3(g1), dim3(64), 0, st, n, L, ws, d_witness, witness_stride);
    hipLaunchKernelGGL(k_prepare, dim3(g1), dim3(64), 0, st, d_sig_xy, n, L, ws, d_witness, witness_stride, 0, 1);
    hipStreamWaitEvent(st, c->ev_aux[0], 0);
    hipStreamWaitEvent(st, c->ev_aux[1], 0);
    hipLaunchKernelGGL(k_pairing, dim3(g1), dim3(64), 0, st, n, L, ws, d_witness, witness_stride, d_result);
    hipStreamWaitEvent(st, c->ev_aux[2], 0);  // join
    return hip_ok(hipGetLastError(), "launch");
}

int blsw_hash_to_g2_batch(const uint8_t* d_msg, uint32_t msg_len, uint64_t n, uint64_t* d_out_affine, void* d_workspace, uint64_t workspace_bytes,
                          void* stream_) {
    if ((!d_msg && msg_len) || n == 0 || !d_workspace || !d_out_affine) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    Workspace ws = carve(d_workspace, n, L);
    if (ws.total_bytes > workspace_bytes) return BLSW_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64);
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, d_msg, msg_len, n, ws);
    hipLaunchKernelGGL(k_map, dim3(g2), dim3(64), 0, st, n, L, ws, (uint64_t*)nullptr, (uint64_t)0);
    hipLaunchKernelGGL(k_cofactor, dim3(g1), dim3(64), 0, st, n, L, ws, (uint64_t*)nullptr, (uint64_t)0);
    hipLaunchKernelGGL(k_h_to_affine, dim3(g1), dim3(64), 0, st, n, ws, d_out_affine);
    return hip_ok(hipGetLastError(), "launch");
}

// which = 0: v_mad_u64_u32 issue rate (result in multiply-adds/s); which = 1: fp_mul rate (Fp products/s). Synchronous.
int blsw_microbench(int which, uint32_t iters, uint32_t blocks, double* ops_per_s) {
    if (!ops_per_s || iters == 0 || blocks == 0) return BLSW_ERR_ARG;
    uint32_t* d = nullptr;
    if (hip_ok(hipMalloc(&d, 4), "malloc")) return BLSW_ERR_HIP;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int threads = which == 0 ? 256 : 64;
    for (int rep = 0; rep < 2; rep++) {  // first pass warms up
        hipEventRecord(e0, 0);
        if (which == 0)
            hipLaunchKernelGGL(k_bench_mad, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else
            hipLaunchKernelGGL(k_bench_fpmul, dim3(blocks), dim3(threads), 0, 0, iters, d);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double per_lane = which == 0 ? 8.0 * iters : 2.0 * iters;
    *ops_per_s = per_lane * blocks * threads / (ms * 1e-3);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d);
    return hip_ok(hipGetLastError(), "microbench");
}
}
