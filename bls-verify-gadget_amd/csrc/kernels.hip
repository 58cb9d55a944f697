// libblsw.so — HIP kernels (gfx950) and the C ABI of include/blsw.h.
// One BLS-verify instance per lane for the field/curve/pairing chains (integer VALU work, no MFMA),
// plus a streaming bit->Fp expansion kernel that writes the ~655k boolean witnesses of the in-circuit
// SHA-256 (93 % of the witness bytes) at HBM-write speed. See DESIGN.md for the data layout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "chains.cuh"
#include "decode.cuh"
#include "layout.h"
#include "team.cuh"

using namespace blsw;

namespace {

// ---------------------------------------------------------------- workspace
// All per-instance scratch is stored element-major: element e of instance I lives at index e*N + I, so that the
// 64 lanes of a wave touch one contiguous 3 KiB window per element (48 B per lane).
struct Workspace {
    uint32_t* bits;  // [N/64][sha_words][64] u32: SHA witness bitstream; a wave appends 256-byte rows to its own tile
    Fp* u;           // [4][N]   hash_to_field output u0.c0,u0.c1,u1.c0,u1.c1
    Fp* q;           // [12][N]  Q0 (x.c0,x.c1,y.c0,y.c1,z.c0,z.c1), Q1
    Fp* h;           // [6][N]   H(m) projective
    Fp* pkaff;       // [2][N]   prepare_g1(pk)
    Fp* coeff;       // [2][272][N]  line coefficients: 0 = H(m), 1 = sig
    Fp* staging;     // [N/64][split_row][64] field witnesses (engine mode), or nullptr (direct mode): each wave of 64
                     // instances owns one contiguous tile and appends 3 KiB rows to it (sequential HBM writes per wave)
    uint64_t staging_rows;
    Fp* pair;            // [N][pair_rows]: rows >= split_row of the staging coordinates (Miller loop, final exponentiation,
    uint32_t split_row;  // is_one), instance-major: the six-lane pairing kernel appends each instance's segment sequentially
    uint32_t pair_rows;  // (split_row = staging_rows, pair_rows = 0 when the single-lane pairing kernel is in use)
    uint64_t sha_words;
    uint64_t total_bytes;
};
inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
// pairing segment: six lanes per instance (default) or the single-lane chain (BLSW_PAIRING=lane, kept for A/B runs)
static bool pairing_team_mode() {
    static int mode = -1;
    if (mode < 0) {
        const char* s = getenv("BLSW_PAIRING");
        mode = (s && s[0] == 'l') ? 0 : 1;
    }
    return mode == 1;
}
inline blsw_layout_t staging_layout(const blsw_layout_t& L);
Workspace carve(void* base, uint64_t N, const blsw_layout_t& L, bool with_staging) {
    Workspace w;
    w.sha_words = (L.sha_bits + 31) / 32 + 1;
    uint64_t off = 0;
    auto take = [&](uint64_t bytes) {
        uint64_t o = off;
        off = align_up(off + bytes, 256);
        return reinterpret_cast<char*>(base) + o;
    };
    w.bits = reinterpret_cast<uint32_t*>(take(w.sha_words * align_up(N, 64) * 4));
    w.u = reinterpret_cast<Fp*>(take(4 * N * sizeof(Fp)));
    w.q = reinterpret_cast<Fp*>(take(12 * N * sizeof(Fp)));
    w.h = reinterpret_cast<Fp*>(take(6 * N * sizeof(Fp)));
    w.pkaff = reinterpret_cast<Fp*>(take(2 * N * sizeof(Fp)));
    w.coeff = reinterpret_cast<Fp*>(take(2ull * 272 * N * sizeof(Fp)));
    w.staging_rows = L.n_witness - L.sha_bits;
    w.split_row = pairing_team_mode() ? staging_layout(L).off_miller : (uint32_t)w.staging_rows;
    w.pair_rows = (uint32_t)w.staging_rows - w.split_row;
    w.staging = with_staging ? reinterpret_cast<Fp*>(take((uint64_t)w.split_row * align_up(N, 64) * sizeof(Fp))) : nullptr;
    w.pair = with_staging && w.pair_rows ? reinterpret_cast<Fp*>(take((uint64_t)w.pair_rows * N * sizeof(Fp))) : nullptr;
    w.total_bytes = off;
    return w;
}

// one submitted batch ("step"): where its inputs are and where its witness tensor / results go
struct StepDesc {
    const uint64_t* pk;
    const uint64_t* sig;
    const uint8_t* msg;
    uint64_t* out;        // [n][out_stride] field elements, or nullptr (results only)
    uint64_t out_stride;  // in field elements
    int32_t* result;
    // aggregate_verify only
    const uint64_t* keys;   // [n][n_keys][12]
    const uint8_t* bitmap;  // [n][n_keys]
    uint32_t* count;        // [n]
};
// a group of `steps` batches of n instances each, processed by one set of launches (N = steps * n lanes per chain)
struct Group {
    uint64_t N;
    uint32_t n;
    uint32_t msg_len;
    const StepDesc* desc;  // device array [steps]
    blsw_layout_t L;       // offsets in the witness vector
    blsw_layout_t LS;      // offsets in the staging rows (the vector with the SHA segment cut out)
    Workspace ws;
    int chain_prio;        // chain waves raise s_setprio
};
// G2 allocation on the six-lane machinery (BLSW_G2=team; needs the six-lane pairing mode): its segment is staged
// instance-major like the pairing rows, so it moves to the end of the staging coordinates
static bool g2_team_mode() {
    static int mode = -1;
    if (mode < 0) {
        const char* s = getenv("BLSW_G2");
        mode = (s && s[0] == 't' && pairing_team_mode()) ? 1 : 0;
    }
    return mode == 1;
}
inline blsw_layout_t staging_layout(const blsw_layout_t& L) {
    blsw_layout_t S = L;
    uint32_t* f = &S.off_msg;
    const uint32_t* g = &L.off_msg;
    for (int k = 0; k < 15; k++) f[k] = g[k] > L.off_expand ? g[k] - L.sha_bits : g[k];
    if (g2_team_mode()) {
        const uint32_t lo = L.off_sig_alloc, len = L.off_pk_not_zero - L.off_sig_alloc;
        for (int k = 0; k < 15; k++)
            if (f[k] > lo) f[k] -= len;
        S.off_sig_alloc = L.n_witness - L.sha_bits - len;  // last rows of the staging coordinates
    }
    return S;
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

// lane -> (step, instance-in-step)
struct LaneId {
    uint64_t I;
    uint32_t s, i;
};
__device__ __forceinline__ LaneId lane_id(const Group& g, uint64_t I) {
    LaneId r;
    r.I = I;
    r.s = (uint32_t)(I / g.n);
    r.i = (uint32_t)(I - (uint64_t)r.s * g.n);
    return r;
}
// witness cursor for a segment: staging row (engine mode), the instance's dense vector (direct mode), or value-only
__device__ __forceinline__ Emitter emitter(const Group& g, const LaneId& id, uint32_t off_full, uint32_t off_staging) {
    Emitter e;
    if (g.ws.staging) {
        if (off_staging >= g.ws.split_row) {  // instance-major rows of the pairing segments
            e.base = reinterpret_cast<uint32_t*>(g.ws.pair + id.I * g.ws.pair_rows);
            e.pos = off_staging - g.ws.split_row;
            e.stride = 12;
            return e;
        }
        e.base = reinterpret_cast<uint32_t*>(g.ws.staging + (id.I >> 6) * (uint64_t)g.ws.split_row * 64 + (id.I & 63));
        e.pos = off_staging;
        e.stride = 64 * 12;
        return e;
    }
    const StepDesc& d = g.desc[id.s];
    e.base = d.out ? reinterpret_cast<uint32_t*>(d.out + (uint64_t)id.i * d.out_stride * 6) : nullptr;
    e.pos = off_full;
    e.stride = 12;
    return e;
}
#define EMIT(g, id, field) emitter(g, id, (g).L.field, (g).LS.field)

// ---------------------------------------------------------------- kernels (one instance per lane)
// SHA-256 witness bits of expand_message (+ the message bits themselves)
__global__ __launch_bounds__(64) void k_sha(Group g, int want_bits, int write_u) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint8_t* msg = g.desc[id.s].msg + (uint64_t)id.i * g.msg_len;
    // UInt8::new_witness_vec(msg): 8 booleans per byte, little-endian
    Emitter em = EMIT(g, id, off_msg);
    for (uint32_t k = 0; k < g.msg_len; k++) {
        uint32_t b = msg[k];
        for (int j = 0; j < 8; j++) em.put_bool((b >> j) & 1);
    }
    BitSink s;
    s.init(want_bits ? g.ws.bits + (I >> 6) * g.ws.sha_words * 64 + (I & 63) : nullptr, 64);
    uint32_t uw[64];
    expand_message_w(s, msg, g.msg_len, false, uw);
    if (write_u)
        for (int j = 0; j < 4; j++) st_fp(g.ws.u + (uint64_t)j * g.N + I, hash_to_field_elem(uw + 16 * j));
}

// value-only expand_message + hash_to_field: hands u0, u1 to k_map without waiting for the witness-bit pass
__global__ __launch_bounds__(64) void k_sha_values(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    uint32_t uw[64];
    expand_message_values(g.desc[id.s].msg + (uint64_t)id.i * g.msg_len, g.msg_len, uw);
    for (int j = 0; j < 4; j++) st_fp(g.ws.u + (uint64_t)j * g.N + I, hash_to_field_elem(uw + 16 * j));
}

// bitstream -> Fp elements: element e of the expand segment = bit ? R mod p : 0. Thread t of a 384-thread block owns
// the 16-byte column c = t % 3 of elements e0 + 128k, so its three R limbs-of-four are loop invariants and one store
// instruction of a wave covers 1 KiB contiguous. ~10 VALU instructions per 16 bytes stored; streaming (nontemporal)
// stores: the tensor is not read again on the device. blockIdx.y = instance of the step.
#ifndef BLSW_EXPAND_ITERS
#define BLSW_EXPAND_ITERS 8
#endif
#ifndef BLSW_EXPAND_THREADS
#define BLSW_EXPAND_THREADS 384  // a multiple of 192: three 16-byte columns per element, whole waves
#endif
#define BLSW_EXPAND_EPI (BLSW_EXPAND_THREADS / 3)  // elements per block and iteration
#ifndef BLSW_EXPAND_UNROLL
#define BLSW_EXPAND_UNROLL 8
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(BLSW_EXPAND_THREADS) void k_sha_expand(const uint32_t* __restrict__ bits, uint64_t sha_words, uint64_t first, uint32_t sha_bits,
                                                    uint32_t off_expand, uint64_t* __restrict__ d_witness, uint64_t stride) {
    constexpr uint32_t R1[12] = BLSW_R1_LIMBS;
    const uint64_t inst = blockIdx.y;
    uint4* out = reinterpret_cast<uint4*>(d_witness + (inst * stride + off_expand) * 6);
    const uint64_t lane = first + inst;
    const uint32_t* b = bits + (lane >> 6) * sha_words * 64 + (lane & 63);
    // The segment is a stream of 16-byte pieces (piece p = element p / 3, column p % 3) that starts at an arbitrary multiple
    // of 16 bytes (instance vectors are 33 956 496 bytes apart). Pieces are assigned from the first 256-byte boundary on
    // (P0 < 16 head pieces are written by block 0 as well), so every 1 KiB wave store covers whole 128-byte lines.
    const uint32_t P0 = (16u - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) & 15u)) & 15u;
    const uint32_t t = threadIdx.x, pt = P0 + t, c = pt % 3;
    const uint32_t e0 = blockIdx.x * (BLSW_EXPAND_EPI * BLSW_EXPAND_ITERS) + pt / 3;
    if (blockIdx.x == 0 && t < P0) {  // head pieces
        const uint32_t he = t / 3, hc = t % 3;
        uint32_t w = b[0];
        uint32_t m = 0u - ((w >> he) & 1u);
        out[(uint64_t)he * 3 + hc] = make_uint4(R1[4 * hc] & m, R1[4 * hc + 1] & m, R1[4 * hc + 2] & m, R1[4 * hc + 3] & m);
    }
    uint4 rc;
    rc.x = c == 0 ? R1[0] : (c == 1 ? R1[4] : R1[8]);
    rc.y = c == 0 ? R1[1] : (c == 1 ? R1[5] : R1[9]);
    rc.z = c == 0 ? R1[2] : (c == 1 ? R1[6] : R1[10]);
    rc.w = c == 0 ? R1[3] : (c == 1 ? R1[7] : R1[11]);
#pragma unroll BLSW_EXPAND_UNROLL
    for (int k = 0; k < BLSW_EXPAND_ITERS; k++) {
        uint32_t e = e0 + BLSW_EXPAND_EPI * k;
        if (e < sha_bits) {
            uint32_t w = b[(uint64_t)(e >> 5) * 64];
            uint32_t m = 0u - ((w >> (e & 31)) & 1u);
            uint4 v = make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m);
            if (NT == 2) {
                uint4* dst = &out[(uint64_t)e * 3 + c];
                u32x4 vv = {v.x, v.y, v.z, v.w};
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(vv) : "memory");
            } else if (NT == 3) {
                uint4* dst = &out[(uint64_t)e * 3 + c];
                u32x4 vv = {v.x, v.y, v.z, v.w};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(vv) : "memory");
            } else if (NT == 1) {
                __builtin_nontemporal_store(v.x, &out[(uint64_t)e * 3 + c].x);
                __builtin_nontemporal_store(v.y, &out[(uint64_t)e * 3 + c].y);
                __builtin_nontemporal_store(v.z, &out[(uint64_t)e * 3 + c].z);
                __builtin_nontemporal_store(v.w, &out[(uint64_t)e * 3 + c].w);
            } else {
                out[(uint64_t)e * 3 + c] = v;
            }
        }
    }
}
// Engine mode: the field witnesses of one step are moved into place around the SHA segment. Rows below split_row are staged
// in 64-instance tiles ([tile][row][64]: 48-byte gathers), the pairing rows instance-major (contiguous copies).
// 16-byte chunk q of instance i covers elements [0, off_expand) and [off_expand + sha_bits, n_witness). Every block
// writes 32 KiB contiguous of ONE instance's vector (16 and 64 KiB measure the same). An LDS-transposed variant with contiguous reads and 384-byte
// writes was measured slower (3.8 ms vs 1.8 ms per 1024 instances).
#ifndef BLSW_PLACE_ITERS
#define BLSW_PLACE_ITERS 8
#endif
__global__ __launch_bounds__(256) void k_place_field(const Fp* __restrict__ staging, const Fp* __restrict__ pair, uint64_t first, uint32_t off_expand,
                                                     uint32_t sha_bits, uint32_t staging_rows, uint32_t split_row, uint64_t* __restrict__ d_witness,
                                                     uint64_t stride, uint32_t n_inst, uint32_t moved_lo, uint32_t moved_len, uint32_t moved_at) {
    // XCD-aware block order: workgroups go round-robin to the 8 XCDs (each with its own L2). The 64 instances of a tile read
    // neighbouring 48-byte pieces of the same staging lines, so all instances of one chunk of rows run back to back on ONE
    // XCD: linear id L -> xcd = L % 8, chunk = xcd + 8 * ((L / 8) / n_inst), instance = (L / 8) % n_inst.
    const uint32_t L = blockIdx.x, s_in_xcd = L >> 3;
    const uint32_t chunk = (L & 7) + 8 * (s_in_xcd / n_inst);
    const uint64_t inst = s_in_xcd % n_inst;
    const uint32_t nchunks = staging_rows * 3;
    if (chunk * (256u * BLSW_PLACE_ITERS) >= nchunks) return;
    const uint64_t lane = first + inst;
    const uint4* src = reinterpret_cast<const uint4*>(staging + (lane >> 6) * (uint64_t)split_row * 64 + (lane & 63));
    const uint4* src2 = reinterpret_cast<const uint4*>(pair + lane * (uint64_t)(staging_rows - split_row));
    uint4* out = reinterpret_cast<uint4*>(d_witness + inst * stride * 6);
    uint32_t q0 = chunk * (256 * BLSW_PLACE_ITERS) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < BLSW_PLACE_ITERS; k++) {
        uint32_t q = q0 + k * 256;
        if (q < nchunks) {
            uint32_t e = q / 3, c = q - e * 3;
            uint4 v = e < split_row ? src[(uint64_t)e * 64 * 3 + c] : src2[(uint64_t)(e - split_row) * 3 + c];
            // staging row -> witness index: the SHA segment is cut out; a segment staged at the end (moved_len rows that belong
            // at moved_lo, staged from row moved_at on) goes back to its place
            uint32_t t = e;
            if (moved_len) t = e >= moved_at ? moved_lo + (e - moved_at) : (e >= moved_lo ? e + moved_len : e);
            uint32_t dst_e = (moved_len && e >= moved_at) ? t : (t < off_expand ? t : t + sha_bits);
            out[(uint64_t)dst_e * 3 + c] = v;
        }
    }
}

__global__ __launch_bounds__(64) void k_g1(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].pk + (uint64_t)id.i * 12);
    G1ChainOut o = chain_g1_alloc(EMIT(g, id, off_pk_alloc), EMIT(g, id, off_pk_not_zero), EMIT(g, id, off_prep_pk), ld_fp(p), ld_fp(p + 1));
    st_fp(g.ws.pkaff + I, o.ax);
    st_fp(g.ws.pkaff + g.N + I, o.ay);
}

// aggregate_verify: lane t = k * N + I allocates key k of instance I (N * n_keys lanes), result to ws.keyproj
__global__ __launch_bounds__(64) void k_agg_keys(Group g, Fp* keyproj) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t N = g.N, nk = g.L.n_keys;
    if (t >= N * nk) return;
    uint32_t k = (uint32_t)(t / N);
    LaneId id = lane_id(g, t - (uint64_t)k * N);
    const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].keys + ((uint64_t)id.i * nk + k) * 12);
    Proj<OpsFp> r = chain_g1_alloc_only(emitter(g, id, g.L.off_keys + k * SEG_PK_ALLOC, g.LS.off_keys + k * SEG_PK_ALLOC), ld_fp(p), ld_fp(p + 1));
    Fp* o = keyproj + t;
    st_fp(o, r.x);
    st_fp(o + N * nk, r.y);
    st_fp(o + 2 * N * nk, r.z);
}
struct KeyProjSrc {
    const Fp* p;  // keyproj + I
    uint64_t N, total;
    __device__ __forceinline__ Proj<OpsFp> ld(uint32_t k) const {
        const Fp* q = p + (uint64_t)k * N;
        return {ld_fp(q), ld_fp(q + total), ld_fp(q + 2 * total)};
    }
};
// aggregate_verify: bitmap booleans, mapped_aggregate, then pk != 0 and prepare_g1 on the aggregated key
__global__ __launch_bounds__(64) void k_agg_sum(Group g, const Fp* keyproj) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint32_t nk = g.L.n_keys;
    const uint8_t* bm = g.desc[id.s].bitmap + (uint64_t)id.i * nk;
    Emitter eb = EMIT(g, id, off_bitmap);
    for (uint32_t k = 0; k < nk; k++) eb.put_bool(bm[k] != 0);  // Boolean::new_witness per key (constraints.rs:414-419)
    KeyProjSrc src = {keyproj + I, g.N, g.N * nk};
    uint32_t count = 0;
    Proj<OpsFp> pk = chain_mapped_aggregate(EMIT(g, id, off_count), EMIT(g, id, off_agg), src, bm, nk, &count);
    G1ChainOut o = chain_g1_post(EMIT(g, id, off_pk_not_zero), EMIT(g, id, off_prep_pk), pk);
    st_fp(g.ws.pkaff + I, o.ax);
    st_fp(g.ws.pkaff + g.N + I, o.ay);
    uint32_t* c = g.desc[id.s].count;
    if (c) c[id.i] = count;
}

__global__ __launch_bounds__(64) void k_g2_alloc(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].sig + (uint64_t)id.i * 24);
    Fp2 sx = {ld_fp(p), ld_fp(p + 1)}, sy = {ld_fp(p + 2), ld_fp(p + 3)};
    chain_g2_alloc(EMIT(g, id, off_sig_alloc), sx, sy);
}

// lanes [0, N): u0 -> Q0 ; lanes [N, 2N): u1 -> Q1
__global__ __launch_bounds__(64) void k_map(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * g.N) return;
    uint32_t which = t >= g.N;
    uint64_t I = which ? t - g.N : t;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Fp2 u = ld_fp2(g.ws.u + (uint64_t)(2 * which) * N + I, N);
    Proj<OpsFp2> q = chain_map_to_curve(which ? EMIT(g, id, off_map1) : EMIT(g, id, off_map0), u);
    Fp* o = g.ws.q + (uint64_t)(6 * which) * N + I;
    st_fp(o, q.x.c0);
    st_fp(o + N, q.x.c1);
    st_fp(o + 2 * N, q.y.c0);
    st_fp(o + 3 * N, q.y.c1);
    st_fp(o + 4 * N, q.z.c0);
    st_fp(o + 5 * N, q.z.c1);
}

__device__ __forceinline__ Proj<OpsFp2> ld_proj2(const Fp* p, uint64_t n) {
    Proj<OpsFp2> r;
    r.x = ld_fp2(p, n);
    r.y = ld_fp2(p + 2 * n, n);
    r.z = ld_fp2(p + 4 * n, n);
    return r;
}
__global__ __launch_bounds__(64) void k_cofactor(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Proj<OpsFp2> q0 = ld_proj2(g.ws.q + I, N), q1 = ld_proj2(g.ws.q + 6 * N + I, N);
    Proj<OpsFp2> h = chain_cofactor(EMIT(g, id, off_add), EMIT(g, id, off_cofactor), q0, q1);
    Fp* o = g.ws.h + I;
    st_fp(o, h.x.c0);
    st_fp(o + N, h.x.c1);
    st_fp(o + 2 * N, h.y.c0);
    st_fp(o + 3 * N, h.y.c1);
    st_fp(o + 4 * N, h.z.c0);
    st_fp(o + 5 * N, h.z.c1);
}

// line coefficients, element-major: coefficient idx of instance I at p[idx * N]
struct CoeffStrided {
    Fp* p;
    uint64_t n;
    __device__ __forceinline__ void st(uint32_t idx, const Fp& v) const { st_fp(p + (uint64_t)idx * n, v); }
    __device__ __forceinline__ Fp ld(uint32_t idx) const { return ld_fp(p + (uint64_t)idx * n); }
};
// which = 0: prepare_g2(H(m)) ; which = 1: prepare_g2(sig)
__global__ __launch_bounds__(64) void k_prepare(Group g, int which) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Proj<OpsFp2> q;
    if (which == 0) {
        q = ld_proj2(g.ws.h + I, N);
    } else {
        const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].sig + (uint64_t)id.i * 24);
        Fp2 sx = {ld_fp(p), ld_fp(p + 1)}, sy = {ld_fp(p + 2), ld_fp(p + 3)};
        bool inf = fp2_is_zero(sx) && fp2_is_zero(sy);
        q.x = inf ? fp2_zero() : sx;
        q.y = inf ? fp2_one() : sy;
        q.z = inf ? fp2_zero() : fp2_one();
    }
    CoeffStrided out = {g.ws.coeff + (uint64_t)which * 272 * N + I, N};
    chain_prepare_g2(which == 0 ? EMIT(g, id, off_prep_h) : EMIT(g, id, off_prep_sig), q, out);
}

// Miller loop + final exponentiation + is_one
__global__ __launch_bounds__(64) void k_pairing(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Fp pkx = ld_fp(g.ws.pkaff + I), pky = ld_fp(g.ws.pkaff + N + I);
    CoeffStrided ch = {g.ws.coeff + I, N};
    CoeffStrided cs = {g.ws.coeff + 272ull * N + I, N};
    Fp12 f = chain_miller(EMIT(g, id, off_miller), pkx, pky, cs, ch);
    bool res = chain_final_exp_is_one(EMIT(g, id, off_final_exp), EMIT(g, id, off_is_one), f);
    int32_t* r = g.desc[id.s].result;
    if (r) r[id.i] = res ? 1 : 0;
}

// Miller loop + final exponentiation + is_one, SIX LANES PER INSTANCE (team.cuh): ten instances per wave, every Fp12
// value distributed over the team's registers, operands and products exchanged through the team's 3.5 KB slot file in LDS
#define BLSW_TEAMS_PER_WAVE 10
__global__ __launch_bounds__(64) void k_pairing_team(Group g) {
    __shared__ Fp2 lds[BLSW_TEAMS_PER_WAVE * TS_NSLOTS];
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t team = threadIdx.x / 6, j = threadIdx.x % 6;
    const uint64_t I0 = (uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE + team;
    const bool active = team < BLSW_TEAMS_PER_WAVE && I0 < g.N;
    const uint64_t I = active ? I0 : 0, N = g.N;  // idle lanes only take part in the barriers
    LaneId id = lane_id(g, I);
    TeamLanes<CoeffStrided> t;
    t.slots = lds + (active ? team : 0) * TS_NSLOTS;
    t.j = j;
    t.active = active;
    t.coeff_h = {g.ws.coeff + I, N};
    t.coeff_sig = {g.ws.coeff + 272ull * N + I, N};
    t.e = EMIT(g, id, off_miller);
    if (!active) t.e.base = nullptr;
    t.set_consts(ld_fp(g.ws.pkaff + I), ld_fp(g.ws.pkaff + N + I));
    Fp2 f = team_miller(t);
    Emitter e_one = EMIT(g, id, off_is_one);
    if (!active) e_one.base = nullptr;
    bool res = team_final_exp_is_one(t, f, e_one);
    int32_t* r = g.desc[id.s].result;
    if (active && j == 0 && r) r[id.i] = res ? 1 : 0;
}
// G2 allocation, six lanes per instance: the (r - 1) * sig chain of the subgroup check runs on the team machinery (points on
// lanes 0..2), the allocation witnesses and the enforce_equal tail are single-lane work of lane 0
__global__ __launch_bounds__(64) void k_g2_alloc_team(Group g) {
    // the G2 op tables use the operand slots and 12 product slots only: 24 slots = 23 KB per wave, six waves per CU
    constexpr uint32_t G2_SLOTS = TS_P + 12;
    __shared__ Fp2 lds[BLSW_TEAMS_PER_WAVE * G2_SLOTS];
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);
    constexpr uint32_t RM1[8] = BLSW_RM1_WORDS;
    const uint32_t team = threadIdx.x / 6, j = threadIdx.x % 6;
    const uint64_t I0 = (uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE + team;
    const bool active = team < BLSW_TEAMS_PER_WAVE && I0 < g.N;
    const uint64_t I = active ? I0 : 0;
    LaneId id = lane_id(g, I);
    const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].sig + (uint64_t)id.i * 24);
    Fp2 sx = {ld_fp(p), ld_fp(p + 1)}, sy = {ld_fp(p + 2), ld_fp(p + 3)};
    const bool inf = fp2_is_zero(sx) && fp2_is_zero(sy);
    Proj<OpsFp2> ge = {inf ? fp2_zero() : sx, inf ? fp2_one() : sy, inf ? fp2_zero() : fp2_one()};
    TeamLanes<CoeffStrided> t;
    t.slots = lds + (active ? team : 0) * G2_SLOTS;
    t.j = j;
    t.active = active;
    t.coeff_h = {nullptr, 0};
    t.coeff_sig = {nullptr, 0};
    t.e = EMIT(g, id, off_sig_alloc);
    if (!active) t.e.base = nullptr;
    Fp2 mine = j == 0 ? ge.x : (j == 1 ? ge.y : (j == 2 ? ge.z : fp2_zero()));
    if (j < 3) {  // the six allocation witnesses: x.c0, x.c1, y.c0, y.c1, z.c0, z.c1
        Emitter w = t.e;
        w.pos += 2 * j;
        w.put(mine.c0);
        w.put(mine.c1);
    }
    t.e.pos += 6;
    (void)team_g2_mul_bits(t, mine, RM1, BLSW_RM1_NBITS);
    if (active && j == 0) chain_g2_alloc_tail(t.e, ge);
}
static void launch_pairing(const Group& g, hipStream_t st) {
    if (!pairing_team_mode())
        hipLaunchKernelGGL(k_pairing, dim3((unsigned)((g.N + 63) / 64)), dim3(64), 0, st, g);
    else
        hipLaunchKernelGGL(k_pairing_team, dim3((unsigned)((g.N + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, st, g);
}

// input decode: lanes [0, n) decompress pk (48 B), lanes [n, 2n) decompress sig (96 B); status[i][0] / status[i][1]
__global__ __launch_bounds__(64) void k_decode(const uint8_t* __restrict__ pk48, const uint8_t* __restrict__ sig96, uint64_t n, uint64_t* pk_xy,
                                               uint64_t* sig_xy, int32_t* status) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n) return;
    if (t < n) {
        Fp x, y;
        int st = g1_decode(pk48 + t * 48, x, y);
        Fp* o = reinterpret_cast<Fp*>(pk_xy + t * 12);
        st_fp(o, x);
        st_fp(o + 1, y);
        status[2 * t] = st;
    } else {
        uint64_t i = t - n;
        Fp2 x, y;
        int st = g2_decode(sig96 + i * 96, x, y);
        Fp* o = reinterpret_cast<Fp*>(sig_xy + i * 24);
        st_fp(o, x.c0);
        st_fp(o + 1, x.c1);
        st_fp(o + 2, y.c0);
        st_fp(o + 3, y.c1);
        status[2 * i + 1] = st;
    }
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

// native signer (bls.rs:411-425, 183-195): lanes [0, n) sig_i = sk_i * H(msg_i) (H projective in ws.h), lanes [n, 2n)
// pk_i = sk_i * g1. Outputs (each optional): compressed bytes and affine Montgomery limbs; status[i] (SIGN_*)
__global__ __launch_bounds__(64) void k_sign(uint64_t n, Workspace ws, const uint8_t* __restrict__ sk32, uint8_t* sig96, uint64_t* sig_xy, uint8_t* pk48,
                                             uint64_t* pk_xy, int32_t* status) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n) return;
    const uint64_t i = t < n ? t : t - n;
    uint32_t k[8];
    int st = sk_from_le32(sk32 + i * 32, k);
    if (t < n) {
        Fp2 x = fp2_zero(), y = fp2_zero();
        bool inf = true;
        if (st == SIGN_OK) {
            Proj<OpsFp2> h = ld_proj2(ws.h + i, n);
            if (!fp2_is_zero(h.z)) {
                Fp2 zi = fp2_inv(h.z);
                Fp2 hx = fp2_mul(h.x, zi), hy = fp2_mul(h.y, zi);
                inf = !g2_mul_affine(hx, hy, k, x, y);
            }
        }
        if (sig_xy) {
            Fp* o = reinterpret_cast<Fp*>(sig_xy + i * 24);
            st_fp(o, x.c0);
            st_fp(o + 1, x.c1);
            st_fp(o + 2, y.c0);
            st_fp(o + 3, y.c1);
        }
        if (sig96) g2_encode(x, y, inf, sig96 + i * 96);
        status[i] = st;
    } else {
        Fp x = fp_zero(), y = fp_zero();
        bool inf = true;
        if (st == SIGN_OK) inf = !g1_mul_affine(K_G1_GEN_X(), fp_neg(K_G1_GEN_NEG_Y()), k, x, y);
        if (pk_xy) {
            Fp* o = reinterpret_cast<Fp*>(pk_xy + i * 12);
            st_fp(o, x);
            st_fp(o + 1, y);
        }
        if (pk48) g1_encode(x, y, inf, pk48 + i * 48);
    }
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

__global__ __launch_bounds__(64) void k_bench_fpinv(uint32_t iters, uint32_t* out) {
    Fp a = fp_one();
    a.l[0] ^= threadIdx.x * 2654435761u + 1;
    a.l[5] ^= blockIdx.x + 1;
    for (uint32_t i = 0; i < iters; i++) {
        a = fp_inv(a);
        a.l[0] ^= i + 1;  // stays < p: only the low limb changes
        a.l[11] &= 0x0fffffffu;
    }
    if (a.l[0] == 0x12345678u && a.l[3] == 0x9abcdef0u) out[0] = a.l[1];
}
__global__ __launch_bounds__(64) void k_bench_fp2mulw(uint32_t iters, uint32_t* out) {
    Fp2 a = fp2_one(), b = fp2_one();
    a.c0.l[0] ^= threadIdx.x + 1;
    b.c1.l[1] ^= blockIdx.x + 1;
    Emitter e = {nullptr, 0};
    for (uint32_t i = 0; i < iters; i++) {
        a = fp2_mul_w(e, a, b);
        b = fp2_sqr_w(e, b);
    }
    if (a.c0.l[0] == 0x12345678u && b.c0.l[3] == 0x9abcdef0u) out[0] = a.c1.l[1] + e.pos;
}

inline int hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) {
        fprintf(stderr, "[blsw] %s: %s\n", what, hipGetErrorString(e));
        return BLSW_ERR_HIP;
    }
    return BLSW_OK;
}

}  // namespace

// Execution engine. Batches ("steps") are SUBMITTED with their input/output pointers and processed in GROUPS of up
// to max_steps batches by one set of launches (N = steps * n lanes per chain kernel), which is what fills the chip:
// one batch of 1024 instances is only 16 waves per chain. Per group:
//   main : sha_values -> map -> cofactor -> prepare(H) ............ -> pairing
//   aux0 : g1_alloc, g2_alloc                                        (needs only pk / sig)
//   aux1 : prepare(sig), sha witness bits                            (need only sig / msg)
// Field witnesses go to an element-major staging area (coalesced stores); then, per step and in submission order,
// the `place` stream writes the step's complete witness tensor: k_sha_expand (bit -> Fp, ~31 MB per instance, the
// HBM-bound kernel) and k_place_field (staging -> its place around the SHA segment). Two group buffers ping-pong,
// so the next group's chains overlap the previous group's placement.
#define BLSW_MAX_BUFFERS 32
#define BLSW_MAX_TIMED 1024
struct GroupBuf {
    void* base;
    Workspace ws;
    StepDesc* h_desc;  // pinned host
    StepDesc* d_desc;
    hipStream_t st[3];  // main, aux0, aux1
    hipEvent_t ev_start, ev_aux[3], ev_chains, ev_done;
    bool used;
};
struct blsw_engine {
    uint64_t n;
    uint32_t msg_len, max_steps;
    blsw_layout_t L, LS;
    GroupBuf buf[BLSW_MAX_BUFFERS];
    int nbuf;
    int cur;
    uint32_t pending;
    hipStream_t place;
    hipEvent_t ev_in;
    // HIP event pairs around every k_sha_expand launch since the last stats reset (live roofline measurement)
    hipEvent_t* ev_exp;  // 2 * BLSW_MAX_TIMED events
    uint32_t n_timed;
    bool staged;  // false: direct mode (max_steps == 1, no staging; witnesses written in place by the chains)
};

static unsigned place_lds_bytes() {
    static int v = -1;
    if (v < 0) {
        const char* s = getenv("BLSW_PLACE_LDS");
        v = s ? atoi(s) : 0;  // optional occupancy limiter for the placement kernel (bytes of dynamic LDS per workgroup)
    }
    return (unsigned)v;
}
static int prio_mode() {  // 0: chains high / placement low; 1: placement high / chains low (default); 2: all equal
    static int v = -1;
    if (v < 0) {
        const char* s = getenv("BLSW_PRIO_MODE");
        v = s ? atoi(s) : 1;
    }
    return v;
}
static int place_nt() {
    static int v = -1;
    if (v < 0) {
        // 0 plain stores (default), 1 nontemporal, 2 sc1, 3 sc0 sc1. Nontemporal stores were the better choice while the chain
        // kernels kept 10 KB stacks in L2 (+10 %); with the stack traffic cut, plain stores win by 8-10 %
        const char* s = getenv("BLSW_EXPAND_NT");
        v = s ? atoi(s) : 0;
    }
    return v;
}
static int launch_group(blsw_engine* e, hipStream_t user_stream) {
    GroupBuf& b = e->buf[e->cur];
    const uint32_t steps = e->pending;
    if (steps == 0) return BLSW_OK;
    Group g;
    g.N = (uint64_t)steps * e->n;
    g.n = (uint32_t)e->n;
    g.msg_len = e->msg_len;
    g.desc = b.d_desc;
    g.L = e->L;
    g.LS = e->LS;
    g.ws = carve(b.base, g.N, e->L, e->staged);
    g.chain_prio = prio_mode() == 0;
    const unsigned g1 = (unsigned)((g.N + 63) / 64), g2 = (unsigned)((2 * g.N + 63) / 64);
    hipStream_t st = b.st[0];
    // inputs are ready once the submitting stream reaches this point
    hipEventRecord(e->ev_in, user_stream);
    hipStreamWaitEvent(st, e->ev_in, 0);
    hipMemcpyAsync(b.d_desc, b.h_desc, sizeof(StepDesc) * steps, hipMemcpyHostToDevice, st);
    hipEventRecord(b.ev_start, st);
    for (int i = 0; i < 2; i++) hipStreamWaitEvent(b.st[1 + i], b.ev_start, 0);
    bool any_out = false;
    for (uint32_t s = 0; s < steps; s++) any_out = any_out || b.h_desc[s].out != nullptr;
    // aux0: group allocations
    hipLaunchKernelGGL(k_g1, dim3(g1), dim3(64), 0, b.st[1], g);
    if (g2_team_mode())
        hipLaunchKernelGGL(k_g2_alloc_team, dim3((unsigned)((g.N + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, b.st[1], g);
    else
        hipLaunchKernelGGL(k_g2_alloc, dim3(g1), dim3(64), 0, b.st[1], g);
    hipEventRecord(b.ev_aux[0], b.st[1]);
    // aux1: prepare_g2(sig), then the SHA-256 witness bits
    hipLaunchKernelGGL(k_prepare, dim3(g1), dim3(64), 0, b.st[2], g, 1);
    hipEventRecord(b.ev_aux[1], b.st[2]);
    if (any_out) hipLaunchKernelGGL(k_sha, dim3(g1), dim3(64), 0, b.st[2], g, 1, 0);
    hipEventRecord(b.ev_aux[2], b.st[2]);
    // main: the hash-to-G2 critical path, then the pairing
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_map, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_prepare, dim3(g1), dim3(64), 0, st, g, 0);
    hipStreamWaitEvent(st, b.ev_aux[0], 0);
    hipStreamWaitEvent(st, b.ev_aux[1], 0);
    launch_pairing(g, st);
    hipStreamWaitEvent(st, b.ev_aux[2], 0);
    hipEventRecord(b.ev_chains, st);
    // placement, per step, in submission order
    hipStreamWaitEvent(e->place, b.ev_chains, 0);
    for (uint32_t s = 0; s < steps && any_out; s++) {
        const StepDesc& d = b.h_desc[s];
        if (!d.out) continue;
        dim3 grid((e->L.sha_bits + BLSW_EXPAND_EPI * BLSW_EXPAND_ITERS - 1) / (BLSW_EXPAND_EPI * BLSW_EXPAND_ITERS), (unsigned)e->n);
        const bool timed = e->n_timed < BLSW_MAX_TIMED;
        if (timed) hipEventRecord(e->ev_exp[2 * e->n_timed], e->place);
#define BLSW_LAUNCH_EXPAND(MODE)                                                                                                                   \
    hipLaunchKernelGGL(k_sha_expand<MODE>, grid, dim3(BLSW_EXPAND_THREADS), place_lds_bytes(), e->place, g.ws.bits, g.ws.sha_words, (uint64_t)s * e->n, e->L.sha_bits, \
                       e->L.off_expand, d.out, d.out_stride)
        switch (place_nt()) {
            case 0: BLSW_LAUNCH_EXPAND(0); break;
            case 2: BLSW_LAUNCH_EXPAND(2); break;
            case 3: BLSW_LAUNCH_EXPAND(3); break;
            default: BLSW_LAUNCH_EXPAND(1); break;
        }
        if (timed) {
            hipEventRecord(e->ev_exp[2 * e->n_timed + 1], e->place);
            e->n_timed++;
        }
        if (e->staged) {
            const uint32_t rows = e->L.n_witness - e->L.sha_bits;
            const unsigned chunks = (rows * 3 + 256 * BLSW_PLACE_ITERS - 1) / (256 * BLSW_PLACE_ITERS);
            dim3 grid2(8 * ((chunks + 7) / 8) * (unsigned)e->n);
            hipLaunchKernelGGL(k_place_field, grid2, dim3(256), 0, e->place, g.ws.staging, g.ws.pair, (uint64_t)s * e->n, e->L.off_expand, e->L.sha_bits, rows,
                               g.ws.split_row, d.out, d.out_stride, (uint32_t)e->n, e->L.off_sig_alloc,
                               g2_team_mode() ? e->L.off_pk_not_zero - e->L.off_sig_alloc : 0u, e->LS.off_sig_alloc);
        }
    }
    hipEventRecord(b.ev_done, e->place);
    b.used = true;
    e->pending = 0;
    e->cur = (e->cur + 1) % e->nbuf;
    return hip_ok(hipGetLastError(), "launch");
}

extern "C" {

int blsw_version(void) { return 4; }

int blsw_layout(uint32_t msg_len, blsw_layout_t* out) {
    if (!out || msg_len > 65535) return BLSW_ERR_ARG;
    make_layout(msg_len, out);
    return BLSW_OK;
}

int blsw_engine_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, uint64_t* bytes) {
    if (!bytes || n == 0 || max_steps == 0 || n_buffers == 0 || n_buffers > BLSW_MAX_BUFFERS) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    Workspace w = carve(nullptr, n * max_steps, L, max_steps > 1 || n_buffers > 1);
    *bytes = (uint64_t)n_buffers * align_up(w.total_bytes, 4096);
    return BLSW_OK;
}

int blsw_engine_create(blsw_engine_t** out, uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, void* d_workspace,
                       uint64_t workspace_bytes) {
    if (!out || n == 0 || max_steps == 0 || !d_workspace || n_buffers == 0 || n_buffers > BLSW_MAX_BUFFERS) return BLSW_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return BLSW_ERR_NO_DEVICE;
    uint64_t need = 0;
    blsw_engine_workspace_bytes(n, msg_len, max_steps, n_buffers, &need);
    if (workspace_bytes < need) return BLSW_ERR_WORKSPACE;
    blsw_engine* e = new blsw_engine();
    e->n = n;
    e->msg_len = msg_len;
    e->max_steps = max_steps;
    e->staged = max_steps > 1 || n_buffers > 1;
    make_layout(msg_len, &e->L);
    e->LS = staging_layout(e->L);
    e->cur = 0;
    e->pending = 0;
    e->n_timed = 0;
    e->ev_exp = new hipEvent_t[2 * BLSW_MAX_TIMED];
    for (int i = 0; i < 2 * BLSW_MAX_TIMED; i++) hipEventCreate(&e->ev_exp[i]);
    e->nbuf = (int)n_buffers;
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // (least, greatest): numerically lower = higher priority
    for (int k = 0; k < e->nbuf; k++) {
        GroupBuf& b = e->buf[k];
        b.base = reinterpret_cast<char*>(d_workspace) + (uint64_t)k * (need / e->nbuf);
        b.used = false;
        if (hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b.h_desc), sizeof(StepDesc) * max_steps, hipHostMallocDefault), "host alloc")) return BLSW_ERR_HIP;
        if (hip_ok(hipMalloc(reinterpret_cast<void**>(&b.d_desc), sizeof(StepDesc) * max_steps), "desc alloc")) return BLSW_ERR_HIP;
        for (int i = 0; i < 3; i++)
            if (hip_ok(hipStreamCreateWithPriority(&b.st[i], hipStreamNonBlocking, prio_mode() == 1 ? prio_lo : prio_hi), "stream create")) return BLSW_ERR_HIP;
        for (int i = 0; i < 3; i++) hipEventCreateWithFlags(&b.ev_aux[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&b.ev_start, hipEventDisableTiming);
        hipEventCreateWithFlags(&b.ev_chains, hipEventDisableTiming);
        hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming);
    }
    if (hip_ok(hipStreamCreateWithPriority(&e->place, hipStreamNonBlocking, prio_mode() == 0 ? prio_lo : prio_hi), "stream create")) return BLSW_ERR_HIP;
    hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming);
    *out = e;
    return hip_ok(hipGetLastError(), "engine create");
}

int blsw_engine_destroy(blsw_engine_t* e) {
    if (!e) return BLSW_ERR_ARG;
    hipDeviceSynchronize();
    for (int k = 0; k < e->nbuf; k++) {
        GroupBuf& b = e->buf[k];
        hipHostFree(b.h_desc);
        hipFree(b.d_desc);
        for (int i = 0; i < 3; i++) hipStreamDestroy(b.st[i]);
        for (int i = 0; i < 3; i++) hipEventDestroy(b.ev_aux[i]);
        hipEventDestroy(b.ev_start);
        hipEventDestroy(b.ev_chains);
        hipEventDestroy(b.ev_done);
    }
    hipStreamDestroy(e->place);
    for (int i = 0; i < 2 * BLSW_MAX_TIMED; i++) hipEventDestroy(e->ev_exp[i]);
    delete[] e->ev_exp;
    hipEventDestroy(e->ev_in);
    delete e;
    return BLSW_OK;
}

int blsw_engine_submit(blsw_engine_t* e, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, uint64_t* d_witness,
                       uint64_t witness_stride, int32_t* d_result, void* stream_) {
    if (!e || !d_pk_xy || !d_sig_xy || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    if (d_witness && witness_stride < e->L.n_witness) return BLSW_ERR_ARG;
    GroupBuf& b = e->buf[e->cur];
    if (e->pending == 0 && b.used) {
        // the buffer's previous group must have been fully placed before its staging is overwritten
        if (hip_ok(hipEventSynchronize(b.ev_done), "event sync")) return BLSW_ERR_HIP;
        b.used = false;
    }
    StepDesc& d = b.h_desc[e->pending];
    d.pk = d_pk_xy;
    d.sig = d_sig_xy;
    d.msg = d_msg;
    d.out = d_witness;
    d.out_stride = witness_stride;
    d.result = d_result;
    d.keys = nullptr;
    d.bitmap = nullptr;
    d.count = nullptr;
    e->pending++;
    if (e->pending == e->max_steps) return launch_group(e, reinterpret_cast<hipStream_t>(stream_));
    return BLSW_OK;
}

// launches whatever is pending and makes `stream` wait for every group issued so far
int blsw_engine_flush(blsw_engine_t* e, void* stream_) {
    if (!e) return BLSW_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    int rc = launch_group(e, st);
    if (rc) return rc;
    for (int k = 0; k < e->nbuf; k++)
        if (e->buf[k].used) hipStreamWaitEvent(st, e->buf[k].ev_done, 0);
    return hip_ok(hipGetLastError(), "flush");
}

// Average duration (ms) of the k_sha_expand launches issued since the last call (HIP events recorded on the stream the
// kernel ran on); blocks until they have finished, then resets the statistics. count may be 0.
int blsw_engine_expand_stats(blsw_engine_t* e, uint32_t* count, float* avg_ms) {
    if (!e || !count || !avg_ms) return BLSW_ERR_ARG;
    double sum = 0;
    for (uint32_t i = 0; i < e->n_timed; i++) {
        if (hip_ok(hipEventSynchronize(e->ev_exp[2 * i + 1]), "event sync")) return BLSW_ERR_HIP;
        float ms = 0;
        if (hip_ok(hipEventElapsedTime(&ms, e->ev_exp[2 * i], e->ev_exp[2 * i + 1]), "event elapsed")) return BLSW_ERR_HIP;
        sum += ms;
    }
    *count = e->n_timed;
    *avg_ms = e->n_timed ? (float)(sum / e->n_timed) : 0.f;
    e->n_timed = 0;
    return BLSW_OK;
}

int blsw_hash_to_g2_batch(const uint8_t* d_msg, uint32_t msg_len, uint64_t n, uint64_t* d_out_affine, void* d_workspace, uint64_t workspace_bytes,
                          void* stream_) {
    if ((!d_msg && msg_len) || n == 0 || !d_workspace || !d_out_affine) return BLSW_ERR_ARG;
    Group g;
    make_layout(msg_len, &g.L);
    g.LS = staging_layout(g.L);
    // the step descriptor lives at the head of the workspace
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(d_workspace);
    g.ws = carve(reinterpret_cast<char*>(d_workspace) + 256, n, g.L, false);
    if (g.ws.total_bytes + 256 > workspace_bytes) return BLSW_ERR_WORKSPACE;
    g.N = n;
    g.n = (uint32_t)n;
    g.msg_len = msg_len;
    g.desc = d_desc;
    g.chain_prio = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    StepDesc h = {nullptr, nullptr, d_msg, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    hipMemcpyAsync(d_desc, &h, sizeof(h), hipMemcpyHostToDevice, st);
    hipStreamSynchronize(st);  // `h` is a stack object
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64);
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_map, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_h_to_affine, dim3(g1), dim3(64), 0, st, n, g.ws, d_out_affine);
    return hip_ok(hipGetLastError(), "launch");
}
// BLS::sign + PublicKey::from(&sk) for a batch (bls.rs:411-425, 183-195). Workspace: blsw_hash_to_g2_workspace_bytes.
int blsw_sign_batch(const uint8_t* d_sk32_le, const uint8_t* d_msg, uint32_t msg_len, uint64_t n, uint8_t* d_sig96, uint64_t* d_sig_xy, uint8_t* d_pk48,
                    uint64_t* d_pk_xy, int32_t* d_status, void* d_workspace, uint64_t workspace_bytes, void* stream_) {
    if (!d_sk32_le || (!d_msg && msg_len) || n == 0 || !d_workspace || !d_status) return BLSW_ERR_ARG;
    Group g;
    make_layout(msg_len, &g.L);
    g.LS = staging_layout(g.L);
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(d_workspace);
    g.ws = carve(reinterpret_cast<char*>(d_workspace) + 256, n, g.L, false);
    if (g.ws.total_bytes + 256 > workspace_bytes) return BLSW_ERR_WORKSPACE;
    g.N = n;
    g.n = (uint32_t)n;
    g.msg_len = msg_len;
    g.desc = d_desc;
    g.chain_prio = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    StepDesc h = {nullptr, nullptr, d_msg, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    hipMemcpyAsync(d_desc, &h, sizeof(h), hipMemcpyHostToDevice, st);
    hipStreamSynchronize(st);  // `h` is a stack object
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64);
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_map, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_sign, dim3(g2), dim3(64), 0, st, n, g.ws, d_sk32_le, d_sig96, d_sig_xy, d_pk48, d_pk_xy, d_status);
    return hip_ok(hipGetLastError(), "launch");
}
int blsw_layout_aggregate(uint32_t msg_len, uint32_t n_keys, blsw_layout_t* out) {
    if (!out || msg_len > 65535) return BLSW_ERR_ARG;
    make_layout(msg_len, out, n_keys);
    return BLSW_OK;
}
static uint64_t agg_workspace(uint64_t n, const blsw_layout_t& L, uint64_t* off_desc, uint64_t* off_keyproj, uint64_t* off_ws) {
    uint64_t o = 0;
    *off_desc = o;
    o = align_up(o + sizeof(StepDesc), 256);
    *off_keyproj = o;
    o = align_up(o + 3ull * n * L.n_keys * sizeof(Fp), 256);
    *off_ws = o;
    return o + carve(nullptr, n, L, false).total_bytes;
}
int blsw_aggregate_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t n_keys, uint64_t* bytes) {
    if (!bytes || n == 0 || n_keys == 0) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L, n_keys);
    uint64_t a, b, c;
    *bytes = agg_workspace(n, L, &a, &b, &c);
    return BLSW_OK;
}
int blsw_aggregate_verify_batch(const uint64_t* d_pks_xy, const uint8_t* d_bitmap, uint32_t n_keys, const uint64_t* d_sig_xy, const uint8_t* d_msg,
                                uint32_t msg_len, uint64_t n, uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, uint32_t* d_count,
                                void* d_workspace, uint64_t workspace_bytes, void* stream_) {
    if (!d_pks_xy || !d_bitmap || n_keys == 0 || !d_sig_xy || (!d_msg && msg_len) || n == 0 || !d_workspace) return BLSW_ERR_ARG;
    Group g;
    make_layout(msg_len, &g.L, n_keys);
    g.LS = g.L;
    if (d_witness && witness_stride < g.L.n_witness) return BLSW_ERR_ARG;
    uint64_t off_desc, off_keyproj, off_ws;
    if (agg_workspace(n, g.L, &off_desc, &off_keyproj, &off_ws) > workspace_bytes) return BLSW_ERR_WORKSPACE;
    char* base = reinterpret_cast<char*>(d_workspace);
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(base + off_desc);
    Fp* keyproj = reinterpret_cast<Fp*>(base + off_keyproj);
    g.ws = carve(base + off_ws, n, g.L, false);
    g.N = n;
    g.n = (uint32_t)n;
    g.msg_len = msg_len;
    g.desc = d_desc;
    g.chain_prio = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    StepDesc h = {nullptr, d_sig_xy, d_msg, d_witness, witness_stride, d_result, d_pks_xy, d_bitmap, d_count};
    hipMemcpyAsync(d_desc, &h, sizeof(h), hipMemcpyHostToDevice, st);
    hipStreamSynchronize(st);  // `h` is a stack object
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64), gk = (unsigned)((n * n_keys + 63) / 64);
    hipLaunchKernelGGL(k_agg_keys, dim3(gk), dim3(64), 0, st, g, keyproj);
    hipLaunchKernelGGL(k_agg_sum, dim3(g1), dim3(64), 0, st, g, (const Fp*)keyproj);
    hipLaunchKernelGGL(k_g2_alloc, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_prepare, dim3(g1), dim3(64), 0, st, g, 1);
    hipLaunchKernelGGL(k_sha, dim3(g1), dim3(64), 0, st, g, d_witness ? 1 : 0, 1);
    if (d_witness) {
        dim3 grid((g.L.sha_bits + BLSW_EXPAND_EPI * BLSW_EXPAND_ITERS - 1) / (BLSW_EXPAND_EPI * BLSW_EXPAND_ITERS), (unsigned)n);
        hipLaunchKernelGGL(k_sha_expand<0>, grid, dim3(BLSW_EXPAND_THREADS), 0, st, g.ws.bits, g.ws.sha_words, (uint64_t)0, g.L.sha_bits, g.L.off_expand, d_witness, witness_stride);
    }
    hipLaunchKernelGGL(k_map, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_prepare, dim3(g1), dim3(64), 0, st, g, 0);
    launch_pairing(g, st);
    return hip_ok(hipGetLastError(), "launch");
}
int blsw_decode_batch(const uint8_t* d_pk48, const uint8_t* d_sig96, uint64_t n, uint64_t* d_pk_xy, uint64_t* d_sig_xy, int32_t* d_status, void* stream_) {
    if (!d_pk48 || !d_sig96 || !d_pk_xy || !d_sig_xy || !d_status || n == 0) return BLSW_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_decode, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st, d_pk48, d_sig96, n, d_pk_xy, d_sig_xy, d_status);
    return hip_ok(hipGetLastError(), "launch");
}
int blsw_hash_to_g2_workspace_bytes(uint64_t n, uint32_t msg_len, uint64_t* bytes) {
    if (!bytes || n == 0) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    *bytes = carve(nullptr, n, L, false).total_bytes + 256;
    return BLSW_OK;
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
    const double per_iter[4] = {8.0, 2.0, 1.0, 5.0};  // MADs, fp products, fp inversions, fp products (one Fp2 mul + one Fp2 sqr)
    if (which < 0 || which > 3) return BLSW_ERR_ARG;
    for (int rep = 0; rep < 2; rep++) {  // first pass warms up
        hipEventRecord(e0, 0);
        if (which == 0)
            hipLaunchKernelGGL(k_bench_mad, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else if (which == 1)
            hipLaunchKernelGGL(k_bench_fpmul, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else if (which == 2)
            hipLaunchKernelGGL(k_bench_fpinv, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else
            hipLaunchKernelGGL(k_bench_fp2mulw, dim3(blocks), dim3(threads), 0, 0, iters, d);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double per_lane = per_iter[which] * iters;
    *ops_per_s = per_lane * blocks * threads / (ms * 1e-3);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d);
    return hip_ok(hipGetLastError(), "microbench");
}
}
