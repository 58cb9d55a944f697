// libblsw.so — HIP kernels (gfx950) and the C ABI of include/blsw.h.
// One BLS-verify instance per lane for the field/curve/pairing chains (integer VALU work, no MFMA),
// plus a streaming bit->Fp expansion kernel that writes the ~655k boolean witnesses of the in-circuit
// SHA-256 (93 % of the witness bytes) at HBM-write speed. See DESIGN.md for the data layout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <deque>
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
    uint32_t* bits;  // [N/64][sha_words/16][64][16] u32: SHA witness bitstream in 64-instance tiles; a wave appends 4 KiB rows
                     // (one 64-byte run of 16 words per instance) to its own tile; sha_words is a multiple of 16
    Fp* u;           // [4][N]   hash_to_field output u0.c0,u0.c1,u1.c0,u1.c1
    Fp* q;           // [12][N]  Q0 (x.c0,x.c1,y.c0,y.c1,z.c0,z.c1), Q1
    Fp* h;           // [6][N]   H(m) projective
    Fp* pkaff;       // [2][N]   prepare_g1(pk)
    Fp* coeff_h;     // [272][N]      line coefficients of prepare_g2(H(m))
    Fp* coeff_sig;   // [272][n_sig]  line coefficients of prepare_g2(sig)
    uint64_t n_sig;  // = N for the single-key circuit; = instances (not pairs) for the N+1-pair product
    Fp* keyproj;     // [3][N * n_keys] allocated keys of the aggregate_verify circuit (projective), else nullptr
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
BLSW_HD inline uint64_t bits_tile_words(uint64_t sha_words) { return sha_words * 64; }  // u32 per 64-instance tile
// kernel variants, fixed per engine at creation (blsw_engine_options_t)
struct Modes {
    bool pairing_team;  // pairing segment: six lanes per instance (default) or the single-lane chain (kept for A/B runs)
    bool g2_team;       // G2 allocation on the six-lane machinery: its segment is staged instance-major like the pairing rows,
                        // so it moves to the end of the staging coordinates
};
constexpr Modes DEFAULT_MODES = {true, false};
inline blsw_layout_t staging_layout(const blsw_layout_t& L, const Modes& m) {
    blsw_layout_t S = L;
    uint32_t* f = &S.off_msg;
    const uint32_t* g = &L.off_msg;
    for (int k = 0; k < 15; k++) f[k] = g[k] > L.off_expand ? g[k] - L.sha_bits : g[k];
    if (m.g2_team) {
        const uint32_t lo = L.off_sig_alloc, len = L.off_pk_not_zero - L.off_sig_alloc;
        for (int k = 0; k < 15; k++)
            if (f[k] > lo) f[k] -= len;
        S.off_sig_alloc = L.n_witness - L.sha_bits - len;  // last rows of the staging coordinates
    }
    return S;
}
// N lanes of per-(pk, msg) work, n_sig lanes of per-signature work (n_sig = N except for the N+1-pair product)
Workspace carve(void* base, uint64_t N, const blsw_layout_t& L, bool with_staging, const Modes& m, uint64_t n_sig = 0) {
    Workspace w;
    if (n_sig == 0) n_sig = N;
    w.sha_words = align_up((L.sha_bits + 31) / 32 + 1, BLSW_BITS_CHUNK_WORDS);
    uint64_t off = 0;
    auto take = [&](uint64_t bytes) {
        uint64_t o = off;
        off = align_up(off + bytes, 256);
        return reinterpret_cast<char*>(base) + o;
    };
    w.bits = reinterpret_cast<uint32_t*>(take(bits_tile_words(w.sha_words) * (align_up(N, 64) / 64) * 4));
    w.u = reinterpret_cast<Fp*>(take(4 * N * sizeof(Fp)));
    w.q = reinterpret_cast<Fp*>(take(12 * N * sizeof(Fp)));
    w.h = reinterpret_cast<Fp*>(take(6 * N * sizeof(Fp)));
    w.pkaff = reinterpret_cast<Fp*>(take(2 * N * sizeof(Fp)));
    w.coeff_h = reinterpret_cast<Fp*>(take(272ull * N * sizeof(Fp)));
    w.coeff_sig = reinterpret_cast<Fp*>(take(272ull * n_sig * sizeof(Fp)));
    w.n_sig = n_sig;
    w.keyproj = L.n_keys ? reinterpret_cast<Fp*>(take(3ull * N * L.n_keys * sizeof(Fp))) : nullptr;
    w.staging_rows = L.n_witness - L.sha_bits;
    w.split_row = m.pairing_team ? staging_layout(L, m).off_miller : (uint32_t)w.staging_rows;
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
    // host side only: the step leaves the engine in its compact wire form (blsw_engine_submit_compact) instead of as witness vectors
    void* compact;
};
// Compact wire form of a step of n instances (n a multiple of 64): the step's slices of the group workspace, back to back —
// [n/64][sha_words/16][64][16] u32 bit words | [n/64][split_row][64] Fp tile-major rows | [n][pair_rows] Fp instance-major rows
struct CompactForm {
    uint64_t bits_bytes, staging_bytes, pair_bytes, off_staging, off_pair, total;
};
static CompactForm compact_form(uint64_t n, const Workspace& w) {
    CompactForm c;
    c.bits_bytes = bits_tile_words(w.sha_words) * (n / 64) * 4;
    c.staging_bytes = (uint64_t)w.split_row * n * sizeof(Fp);
    c.pair_bytes = (uint64_t)w.pair_rows * n * sizeof(Fp);
    c.off_staging = align_up(c.bits_bytes, 256);
    c.off_pair = align_up(c.off_staging + c.staging_bytes, 256);
    c.total = align_up(c.off_pair + c.pair_bytes, 256);
    return c;
}
// a group of `steps` batches of n instances each, processed by one set of launches (N = steps * n * K lanes per chain;
// K = (pk, msg) pairs per instance: 1 except for the N+1-pair product)
struct Group {
    uint64_t N;
    uint32_t n;   // instances per step
    uint32_t K;   // pairs per instance
    uint32_t msg_len;
    const StepDesc* desc;  // device array [steps]
    blsw_layout_t L;       // offsets in the witness vector
    blsw_layout_t LS;      // offsets in the staging rows (the vector with the SHA segment cut out)
    Workspace ws;
    int chain_prio;        // chain waves raise s_setprio
};

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

// lane -> (step, instance-in-step, pair). Inputs of per-pair work are indexed by f (flat [n][K]), outputs by (i, j).
struct LaneId {
    uint64_t I;
    uint32_t s, i, j, f;
};
__device__ __forceinline__ LaneId lane_id(const Group& g, uint64_t I) {
    LaneId r;
    r.I = I;
    const uint32_t nk = g.n * g.K;
    r.s = (uint32_t)(I / nk);
    r.f = (uint32_t)(I - (uint64_t)r.s * nk);
    r.i = g.K == 1 ? r.f : r.f / g.K;
    r.j = g.K == 1 ? 0u : r.f - r.i * g.K;
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
// segment that exists once per pair: pair j's copy starts j * stride further (staged groups always have K = 1)
#define EMITJ(g, id, field, stride) emitter(g, id, (g).L.field + (id).j * (g).L.stride, (g).LS.field + (id).j * (g).L.stride)

// ---------------------------------------------------------------- kernels (one instance per lane)
// SHA-256 witness bits of expand_message (+ the message bits themselves)
__global__ __launch_bounds__(64) void k_sha(Group g, int want_bits, int write_u) {
    __shared__ uint32_t sha_lds[BLSW_BITS_CHUNK_WORDS * 64];  // the wave's word buffer of the bit sink
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint8_t* msg = g.desc[id.s].msg + (uint64_t)id.f * g.msg_len;
    // UInt8::new_witness_vec(msg): 8 booleans per byte, little-endian
    Emitter em = EMITJ(g, id, off_msg, stride_msg);
    for (uint32_t k = 0; k < g.msg_len; k++) {
        uint32_t b = msg[k];
        for (int j = 0; j < 8; j++) em.put_bool((b >> j) & 1);
    }
    uint32_t uw[64];
    if (want_bits) {
        BitSink s;
        s.init_device(sha_lds + threadIdx.x, reinterpret_cast<uint4*>(g.ws.bits + (I >> 6) * bits_tile_words(g.ws.sha_words) + (I & 63) * BLSW_BITS_CHUNK_WORDS));
        expand_message_w(s, msg, g.msg_len, false, uw);
    } else {
        expand_message_values(msg, g.msg_len, uw);  // the device sink always stores: no bits wanted = the value-only SHA
    }
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
    expand_message_values(g.desc[id.s].msg + (uint64_t)id.f * g.msg_len, g.msg_len, uw);
    for (int j = 0; j < 4; j++) st_fp(g.ws.u + (uint64_t)j * g.N + I, hash_to_field_elem(uw + 16 * j));
}

// bitstream -> Fp elements: element e of the expand segment = bit ? R mod p : 0. The segment is a stream of 16-byte pieces
// (piece p = element p / 3, column p % 3) that starts at an arbitrary multiple of 16 bytes (instance vectors are 33 956 496
// bytes apart). blockIdx.y = flat (instance, pair) index of the step. Three store geometries (blsw_engine_options_t::
// expand_variant; measured alone in tools/expand_lab.hip -> profiles/r02_expand_lab.txt, and in the pipeline by bench.py):
//   0  384-thread workgroups, 8 pieces per thread 6 KiB apart, pieces counted from the first 256-byte boundary
//   1  256-thread workgroups, ONE piece per thread, every workgroup writes one 4 KiB-aligned 4 KiB chunk of the address space
//   2  768-thread workgroups, 8 pieces per thread 12 KiB apart: every iteration writes three 4 KiB-aligned chunks; column and
//      bit position of a thread are loop invariants (768 = 3 * 256 pieces = 256 elements = 8 bit words per iteration)
//   3  as 2 with 4 pieces per thread;  4  as 2 with 16;  5  384 threads x 16 pieces, 4 KiB-aligned start
struct ExpandArgs {
    const uint32_t* bits;
    uint64_t sha_words, first;
    uint32_t sha_bits, off_expand;
    uint64_t* d_witness;
    uint64_t stride;
    uint32_t K, stride_hash;  // K = 1 for the single-key circuit
    int prio;                 // raise the wave priority (s_setprio 3): wins VALU issue arbitration against the chain waves
    int canonical;            // elements as canonical integers (true = 1) instead of Montgomery form (true = R mod p)
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ void expand_store(uint4* dst, const uint4& v) {
    if (NT == 2) {
        u32x4 vv = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(vv) : "memory");
    } else if (NT == 3) {
        u32x4 vv = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(vv) : "memory");
    } else if (NT == 1) {
        __builtin_nontemporal_store(v.x, &dst->x);
        __builtin_nontemporal_store(v.y, &dst->y);
        __builtin_nontemporal_store(v.z, &dst->z);
        __builtin_nontemporal_store(v.w, &dst->w);
    } else {
        *dst = v;
    }
}
__device__ __forceinline__ uint4 expand_column(uint32_t c, int canonical) {
    if (canonical) return make_uint4(c == 0 ? 1u : 0u, 0u, 0u, 0u);
    constexpr uint32_t R1[12] = BLSW_R1_LIMBS;
    uint4 rc;
    rc.x = c == 0 ? R1[0] : (c == 1 ? R1[4] : R1[8]);
    rc.y = c == 0 ? R1[1] : (c == 1 ? R1[5] : R1[9]);
    rc.z = c == 0 ? R1[2] : (c == 1 ? R1[6] : R1[10]);
    rc.w = c == 0 ? R1[3] : (c == 1 ? R1[7] : R1[11]);
    return rc;
}
// instance / pair of this workgroup: where its segment starts and where its bit words are
__device__ __forceinline__ void expand_locate(const ExpandArgs& a, uint4*& out, const uint32_t*& b) {
    const uint64_t inst = a.K == 1 ? blockIdx.y : blockIdx.y / a.K;
    const uint32_t pair = a.K == 1 ? 0u : blockIdx.y - (uint32_t)inst * a.K;
    out = reinterpret_cast<uint4*>(a.d_witness + (inst * a.stride + a.off_expand + (uint64_t)pair * a.stride_hash) * 6);
    const uint64_t lane = a.first + blockIdx.y;
    b = a.bits + (lane >> 6) * bits_tile_words(a.sha_words) + (lane & 63) * BLSW_BITS_CHUNK_WORDS;
}
// word w of the instance whose stream starts at b (64-byte runs of 16 words, 64 instances interleaved per chunk)
__device__ __forceinline__ uint32_t expand_word(const uint32_t* b, uint32_t w) {
    #ifdef BLSW_DEBUG_EXPAND_NOREAD  // timing experiment
    return 0x55555555u + w;
#endif
    return b[(uint64_t)(w / BLSW_BITS_CHUNK_WORDS) * (64 * BLSW_BITS_CHUNK_WORDS) + (w % BLSW_BITS_CHUNK_WORDS)];
}
// pieces [0, P0) in front of the first boundary: written by workgroup 0 of every variant
template <int NT>
__device__ __forceinline__ void expand_head(uint4* out, const uint32_t* b, uint32_t P0, uint32_t n_pieces, int canonical) {
    if (blockIdx.x == 0 && threadIdx.x < P0 && threadIdx.x < n_pieces) {
        const uint32_t e = threadIdx.x / 3, c = threadIdx.x - 3 * e;
        const uint32_t m = 0u - ((expand_word(b, e >> 5) >> (e & 31)) & 1u);
        const uint4 rc = expand_column(c, canonical);
        expand_store<NT>(&out[threadIdx.x], make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m));
    }
}
// variant 0 (and, with THREADS = 768 and 4 KiB alignment, variants 2 / 3): THREADS is a multiple of 192, so a thread's column is a
// loop invariant; a workgroup writes THREADS * ITERS consecutive pieces, THREADS of them per iteration
template <int THREADS, int ITERS, int ALIGN_PIECES, int NT>
__global__ __launch_bounds__(THREADS) void k_sha_expand(ExpandArgs a) {
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    uint4* out;
    const uint32_t* b;
    expand_locate(a, out, b);
    const uint32_t n_pieces = a.sha_bits * 3;
    const uint32_t P0 = (ALIGN_PIECES - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) % ALIGN_PIECES)) % ALIGN_PIECES;
    expand_head<NT>(out, b, P0, n_pieces, a.canonical);
    const uint32_t pt = P0 + threadIdx.x, c = pt % 3;
    const uint32_t e0 = blockIdx.x * ((THREADS / 3) * ITERS) + pt / 3;
    const uint4 rc = expand_column(c, a.canonical);
    // THREADS / 3 is a multiple of 32: the bit position of a thread is a loop invariant too, its word advances by THREADS / 96
    static_assert((THREADS / 3) % 32 == 0, "bit position must be loop invariant");
    const uint32_t sh = e0 & 31, w0 = e0 >> 5;
    uint4* dst = out + (uint64_t)e0 * 3 + c;
    if (blockIdx.x * ((THREADS / 3) * ITERS) + (P0 + THREADS - 1) / 3 + (THREADS / 3) * (ITERS - 1) < a.sha_bits) {
        // whole workgroup in range (all but the last one or two of an instance): all bit words first, then the stores back to
        // back — no bounds checks, no wait between a store and the next load
#ifdef BLSW_DEBUG_EXPAND_CONST  // timing experiment: the stores without the bit logic (wrong witnesses)
#pragma unroll
        for (int k = 0; k < ITERS; k++) expand_store<NT>(dst + (uint64_t)k * THREADS, rc);
        return;
#endif
        uint32_t w[ITERS];
#pragma unroll
        for (int k = 0; k < ITERS; k++) w[k] = expand_word(b, w0 + k * (THREADS / 96));
#pragma unroll
        for (int k = 0; k < ITERS; k++) {
            const uint32_t m = 0u - ((w[k] >> sh) & 1u);
            expand_store<NT>(dst + (uint64_t)k * THREADS, make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m));
        }
        return;
    }
#pragma unroll 1
    for (int k = 0; k < ITERS; k++) {
        const uint32_t e = e0 + (THREADS / 3) * k;
        if (e < a.sha_bits) {
            const uint32_t m = 0u - ((expand_word(b, w0 + k * (THREADS / 96)) >> sh) & 1u);
            expand_store<NT>(dst + (uint64_t)k * THREADS, make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m));
        }
    }
}
// variant 1: one piece per thread, workgroup b >= 1 writes exactly one 4 KiB-aligned chunk
template <int NT>
__global__ __launch_bounds__(256) void k_sha_expand_chunk(ExpandArgs a) {
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    uint4* out;
    const uint32_t* b;
    expand_locate(a, out, b);
    const uint32_t n_pieces = a.sha_bits * 3;
    const uint32_t P0 = (256 - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) % 256)) % 256;
    expand_head<NT>(out, b, P0, n_pieces, a.canonical);
    const uint32_t p = P0 + blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pieces) return;
    const uint32_t e = p / 3, c = p - 3 * e;
    const uint32_t m = 0u - ((expand_word(b, e >> 5) >> (e & 31)) & 1u);
    const uint4 rc = expand_column(c, a.canonical);
    expand_store<NT>(&out[p], make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m));
}
// variant: low byte 0..3 = geometry, bit 8 = raised wave priority; store: 0 plain, 1 nontemporal, 2 sc1, 3 sc0 sc1 (variant 0 only)
static void launch_expand(uint32_t variant, uint32_t store, unsigned lds, hipStream_t st, ExpandArgs a, unsigned n_y) {
    a.prio = (variant >> 8) & 1;
    const uint32_t n_pieces = a.sha_bits * 3;
    auto grid = [&](uint32_t per_wg) { return dim3((n_pieces + per_wg - 1) / per_wg + 1, n_y); };
    switch (variant & 0xff) {
        case 1: hipLaunchKernelGGL(k_sha_expand_chunk<0>, grid(256), dim3(256), lds, st, a); break;
        case 2: hipLaunchKernelGGL((k_sha_expand<768, 8, 256, 0>), grid(768 * 8), dim3(768), lds, st, a); break;
        case 3: hipLaunchKernelGGL((k_sha_expand<768, 4, 256, 0>), grid(768 * 4), dim3(768), lds, st, a); break;
        case 4: hipLaunchKernelGGL((k_sha_expand<768, 16, 256, 0>), grid(768 * 16), dim3(768), lds, st, a); break;
        case 5: hipLaunchKernelGGL((k_sha_expand<384, 16, 256, 0>), grid(384 * 16), dim3(384), lds, st, a); break;
        default:
            switch (store) {
                case 1: hipLaunchKernelGGL((k_sha_expand<384, 8, 16, 1>), grid(384 * 8), dim3(384), lds, st, a); break;
                case 2: hipLaunchKernelGGL((k_sha_expand<384, 8, 16, 2>), grid(384 * 8), dim3(384), lds, st, a); break;
                case 3: hipLaunchKernelGGL((k_sha_expand<384, 8, 16, 3>), grid(384 * 8), dim3(384), lds, st, a); break;
                default: hipLaunchKernelGGL((k_sha_expand<384, 8, 16, 0>), grid(384 * 8), dim3(384), lds, st, a); break;
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
    const Fp* p = reinterpret_cast<const Fp*>(g.desc[id.s].pk + (uint64_t)id.f * 12);
    G1ChainOut o = chain_g1_alloc(EMITJ(g, id, off_pk_alloc, stride_pk_alloc), EMITJ(g, id, off_pk_not_zero, stride_pk_not_zero),
                                  EMITJ(g, id, off_prep_pk, stride_prep_pk), ld_fp(p), ld_fp(p + 1));
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
    Proj<OpsFp2> q = chain_map_to_curve(which ? EMITJ(g, id, off_map1, stride_hash) : EMITJ(g, id, off_map0, stride_hash), u);
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
    Proj<OpsFp2> h = chain_cofactor(EMITJ(g, id, off_add, stride_hash), EMITJ(g, id, off_cofactor, stride_hash), q0, q1);
    Fp* o = g.ws.h + I;
    st_fp(o, h.x.c0);
    st_fp(o + N, h.x.c1);
    st_fp(o + 2 * N, h.y.c0);
    st_fp(o + 3 * N, h.y.c1);
    st_fp(o + 4 * N, h.z.c0);
    st_fp(o + 5 * N, h.z.c1);
}

// Value-only entry points (hash_to_g2 batch, signer): the same group element without the circuit's witness structure — the
// in-circuit clear_cofactor2 is an AFFINE double-and-add with one slope inversion per step (939 Fp2 inversions, App. A.5);
// here it is a Jacobian ladder over the 636 bits of h_eff with two inversions in all. Output as k_cofactor's: homogeneous (x, y, z).
// The ladder is written out with inlined Fp2 operations (only the Fp product and the inversion are calls), so that the kernel
// fits two waves per SIMD: the shared jac2_dbl / jac2_add_mixed are separate functions that take 248 VGPRs + 32 AGPRs each.
namespace {
__device__ __forceinline__ Fp2 v_sqr(const Fp2& a) {
    Fp v = fp_mul(a.c0, a.c1);
    Fp t = fp_mul(fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1));
    return {t, fp_dbl(v)};
}
__device__ __forceinline__ Jac2 v_dbl(const Jac2& p) {  // dbl-2009-l, a = 0
    Fp2 A = v_sqr(p.x), B = v_sqr(p.y), C = v_sqr(B);
    Fp2 D = fp2_dbl(fp2_sub(fp2_sub(v_sqr(fp2_add(p.x, B)), A), C));
    Fp2 E = fp2_add(fp2_dbl(A), A);
    Fp2 x3 = fp2_sub(v_sqr(E), fp2_dbl(D));
    Fp2 y3 = fp2_sub(fp2_mul_inl(E, fp2_sub(D, x3)), fp2_dbl(fp2_dbl(fp2_dbl(C))));
    Fp2 z3 = fp2_dbl(fp2_mul_inl(p.y, p.z));
    return {x3, y3, z3};
}
__device__ __forceinline__ Jac1v v1_dbl(const Jac1v& p) {
    Fp A = fp_sqr(p.x), B = fp_sqr(p.y), C = fp_sqr(B);
    Fp D = fp_dbl(fp_sub(fp_sub(fp_sqr(fp_add(p.x, B)), A), C));
    Fp E = fp_add(fp_dbl(A), A);
    Fp x3 = fp_sub(fp_sqr(E), fp_dbl(D));
    Fp y3 = fp_sub(fp_mul(E, fp_sub(D, x3)), fp_dbl(fp_dbl(fp_dbl(C))));
    Fp z3 = fp_dbl(fp_mul(p.y, p.z));
    return {x3, y3, z3};
}
__device__ __forceinline__ Jac1v v1_add_mixed(const Jac1v& p, const Fp& qx, const Fp& qy) {
    if (fp_is_zero(p.z)) return {qx, qy, fp_one()};
    Fp z1z1 = fp_sqr(p.z);
    Fp u2 = fp_mul(qx, z1z1);
    Fp s2 = fp_mul(fp_mul(qy, p.z), z1z1);
    Fp h = fp_sub(u2, p.x);
    Fp rr = fp_dbl(fp_sub(s2, p.y));
    if (fp_is_zero(h)) {
        if (fp_is_zero(rr)) return v1_dbl(p);
        return {fp_one(), fp_one(), fp_zero()};
    }
    Fp hh = fp_sqr(h);
    Fp i = fp_dbl(fp_dbl(hh));
    Fp j = fp_mul(h, i);
    Fp v = fp_mul(p.x, i);
    Fp x3 = fp_sub(fp_sub(fp_sqr(rr), j), fp_dbl(v));
    Fp y3 = fp_sub(fp_mul(rr, fp_sub(v, x3)), fp_dbl(fp_mul(p.y, j)));
    Fp z3 = fp_sub(fp_sub(fp_sqr(fp_add(p.z, h)), z1z1), hh);
    return {x3, y3, z3};
}
__device__ __forceinline__ Jac2 v_add_mixed(const Jac2& p, const Fp2& qx, const Fp2& qy) {  // madd-2007-bl; p = 0, p = +-q handled
    if (fp2_is_zero(p.z)) return {qx, qy, fp2_one()};
    Fp2 z1z1 = v_sqr(p.z);
    Fp2 u2 = fp2_mul_inl(qx, z1z1);
    Fp2 s2 = fp2_mul_inl(fp2_mul_inl(qy, p.z), z1z1);
    Fp2 h = fp2_sub(u2, p.x);
    Fp2 rr = fp2_dbl(fp2_sub(s2, p.y));
    if (fp2_is_zero(h)) {
        if (fp2_is_zero(rr)) return v_dbl(p);
        return {fp2_one(), fp2_one(), fp2_zero()};
    }
    Fp2 hh = v_sqr(h);
    Fp2 i = fp2_dbl(fp2_dbl(hh));
    Fp2 j = fp2_mul_inl(h, i);
    Fp2 v = fp2_mul_inl(p.x, i);
    Fp2 x3 = fp2_sub(fp2_sub(v_sqr(rr), j), fp2_dbl(v));
    Fp2 y3 = fp2_sub(fp2_mul_inl(rr, fp2_sub(v, x3)), fp2_dbl(fp2_mul_inl(p.y, j)));
    Fp2 z3 = fp2_sub(fp2_sub(v_sqr(fp2_add(p.z, h)), z1z1), hh);
    return {x3, y3, z3};
}
}  // namespace
// map_to_curve_9mod16 + isogeny_map (hasher.rs:352-502, 294-348) for the value-only entries: the statements of
// chain_map_to_curve without the witness cursor, on the inlined Fp2 operations above (two waves per SIMD). Same field
// operations in the same order, so the result is the same element bit for bit.
namespace {
__device__ __forceinline__ bool v_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
__device__ __forceinline__ Fp2 v_sel(bool c, const Fp2& a, const Fp2& b) { return c ? a : b; }
__device__ __forceinline__ bool v_sgn0(const Fp2& v) {  // hasher.rs:520-530
    const Fp c0 = fp_to_canonical(v.c0), c1 = fp_to_canonical(v.c1);
    return (c0.l[0] & 1) || (fp_is_zero(v.c0) && (c1.l[0] & 1));
}
__device__ __forceinline__ Fp2 v_poly(const Fp2* k, int n, const Fp2& x) {  // sum k[i] x^i, powers as DensePolynomialVar::evaluate builds them
    Fp2 result = k[0], cp = x;
    for (int i = 1; i < n; i++) {
        result = fp2_add(result, fp2_mul_inl(cp, k[i]));
        if (i + 1 < n) cp = fp2_mul_inl(cp, x);
    }
    return result;
}
__device__ __forceinline__ Proj<OpsFp2> v_map_to_curve(const Fp2& u) {
    constexpr uint32_t C1[24] = BLSW_SSWU_C1_WORDS;
    const Fp2 Z = K_SSWU_Z(), A = K_SSWU_A(), B = K_SSWU_B(), C2 = K_SSWU_C2(), C3 = K_SSWU_C3(), C4 = K_SSWU_C4(), C5 = K_SSWU_C5();
    Fp2 tv1 = v_sqr(u);
    Fp2 tv3 = fp2_mul_inl(Z, tv1);
    Fp2 tv5 = v_sqr(tv3);
    Fp2 xd = fp2_add(tv5, tv3);
    Fp2 x1n = fp2_mul_inl(fp2_add(xd, fp2_one()), B);
    xd = fp2_mul_inl(K_SSWU_NEG_A(), xd);
    xd = v_sel(fp2_is_zero(xd), K_SSWU_ZA(), xd);
    Fp2 tv2 = v_sqr(xd);
    Fp2 gxd = fp2_mul_inl(tv2, xd);
    tv2 = fp2_mul_inl(A, tv2);
    Fp2 gx1 = fp2_add(v_sqr(x1n), tv2);
    gx1 = fp2_mul_inl(gx1, x1n);
    gx1 = fp2_add(gx1, fp2_mul_inl(B, gxd));
    Fp2 tv4 = v_sqr(gxd);
    tv2 = fp2_mul_inl(tv4, gxd);
    tv4 = v_sqr(tv4);
    tv2 = fp2_mul_inl(tv2, tv4);
    tv2 = fp2_mul_inl(tv2, gx1);
    tv4 = v_sqr(tv4);
    tv4 = fp2_mul_inl(tv2, tv4);
    Fp2 y = tv4;  // y = tv4 ^ c1 (bits 759, 758 are zero, bit 757 is the leading one)
#pragma unroll 1
    for (int i = BLSW_SSWU_C1_NBITS - 4; i >= 0; i--) {
        y = v_sqr(y);
        if (bit_of(C1, i)) y = fp2_mul_inl(y, tv4);
    }
    y = fp2_mul_inl(y, tv2);
    tv4 = fp2_mul_inl(y, C2);
    y = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx1), tv4, y);
    tv4 = fp2_mul_inl(y, C3);
    y = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx1), tv4, y);
    tv4 = fp2_mul_inl(tv4, C2);
    y = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx1), tv4, y);
    Fp2 gx2 = fp2_mul_inl(fp2_mul_inl(gx1, tv5), tv3);
    tv5 = fp2_mul_inl(fp2_mul_inl(y, tv1), u);
    tv1 = fp2_mul_inl(tv5, C4);
    tv4 = fp2_mul_inl(tv1, C2);
    tv1 = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx2), tv4, tv1);
    tv4 = fp2_mul_inl(tv5, C5);
    tv1 = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx2), tv4, tv1);
    tv4 = fp2_mul_inl(tv4, C2);
    tv1 = v_sel(v_eq(fp2_mul_inl(v_sqr(tv4), gxd), gx2), tv4, tv1);
    const bool e8 = v_eq(fp2_mul_inl(v_sqr(y), gxd), gx1);
    y = v_sel(e8, y, tv1);
    const Fp2 xn = v_sel(e8, x1n, fp2_mul_inl(tv3, x1n));
    const bool e9 = !(v_sgn0(u) ^ v_sgn0(y));
    y = v_sel(e9, y, fp2_neg(y));
    // to_projective_short (hasher.rs:551-559), to_affine_unchecked (:569-583), isogeny_map (:294-348)
    const Fp2 xd3 = fp2_mul_inl(v_sqr(xd), xd);
    const Fp2 jx = fp2_mul_inl(xn, xd), jy = fp2_mul_inl(y, xd3);
    const bool is_infinity = fp2_is_zero(xd);
    const Fp2 zi = fp2_inv_inl(xd), zi2 = v_sqr(zi);
    const Fp2 ax = fp2_mul_inl(jx, zi2), ay = fp2_mul_inl(jy, fp2_mul_inl(zi2, zi));
    const Fp2 kxd[3] = {K_ISO_XDEN0(), K_ISO_XDEN1(), K_ISO_XDEN2()};
    const Fp2 kyd[4] = {K_ISO_YDEN0(), K_ISO_YDEN1(), K_ISO_YDEN2(), K_ISO_YDEN3()};
    const Fp2 kxn[4] = {K_ISO_XNUM0(), K_ISO_XNUM1(), K_ISO_XNUM2(), K_ISO_XNUM3()};
    const Fp2 kyn[4] = {K_ISO_YNUM0(), K_ISO_YNUM1(), K_ISO_YNUM2(), K_ISO_YNUM3()};
    const Fp2 x_den_inv = fp2_inv_inl(v_poly(kxd, 3, ax));
    const Fp2 y_den_inv = fp2_inv_inl(v_poly(kyd, 4, ax));
    const Fp2 img_x = fp2_mul_inl(v_poly(kxn, 4, ax), x_den_inv);
    const Fp2 img_y = fp2_mul_inl(fp2_mul_inl(v_poly(kyn, 4, ax), ay), y_den_inv);
    Proj<OpsFp2> q;
    q.x = v_sel(is_infinity, fp2_zero(), img_x);
    q.y = v_sel(is_infinity, fp2_zero(), img_y);
    q.z = is_infinity ? fp2_zero() : fp2_one();
    return q;
}
}  // namespace
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_map_values(Group g) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * g.N) return;
    const uint32_t which = t >= g.N;
    const uint64_t I = which ? t - g.N : t, N = g.N;
    const Proj<OpsFp2> q = v_map_to_curve(ld_fp2(g.ws.u + (uint64_t)(2 * which) * N + I, N));
    Fp* o = g.ws.q + (uint64_t)(6 * which) * N + I;
    st_fp(o, q.x.c0);
    st_fp(o + N, q.x.c1);
    st_fp(o + 2 * N, q.y.c0);
    st_fp(o + 3 * N, q.y.c1);
    st_fp(o + 4 * N, q.z.c0);
    st_fp(o + 5 * N, q.z.c1);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_cofactor_values(Group g) {
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    constexpr uint32_t HE[20] = BLSW_H_EFF_WORDS;
    const uint64_t N = g.N;
    // Q0, Q1 leave the isogeny as (x, y, 1) or (0, 0, 0) (hasher.rs:339-345): affine points or the identity
    Jac2 r;
    {
        const Proj<OpsFp2> q0 = ld_proj2(g.ws.q + I, N);
        r = {q0.x, q0.y, q0.z};
        if (fp2_is_zero(q0.z)) r = {fp2_one(), fp2_one(), fp2_zero()};
    }
    {
        const Proj<OpsFp2> q1 = ld_proj2(g.ws.q + 6 * N + I, N);
        if (!fp2_is_zero(q1.z)) r = v_add_mixed(r, q1.x, q1.y);  // hasher.rs:656 (handles Q0 = +-Q1 and Q0 = 0)
    }
    Proj<OpsFp2> h = {fp2_zero(), fp2_one(), fp2_zero()};
    if (!fp2_is_zero(r.z)) {
        const Fp2 zi = fp2_inv_inl(r.z), zi2 = v_sqr(zi);
        const Fp2 ax = fp2_mul_inl(r.x, zi2), ay = fp2_mul_inl(r.y, fp2_mul_inl(zi2, zi));
        Jac2 acc = {ax, ay, fp2_one()};
#pragma unroll 1
        for (int i = BLSW_H_EFF_NBITS - 2; i >= 0; i--) {
            acc = v_dbl(acc);
            if (bit_of(HE, i)) acc = v_add_mixed(acc, ax, ay);
        }
        if (!fp2_is_zero(acc.z)) {  // (X / Z^2, Y / Z^3) as homogeneous (X Z, Y, Z^3)
            h.x = fp2_mul_inl(acc.x, acc.z);
            h.y = acc.y;
            h.z = fp2_mul_inl(v_sqr(acc.z), acc.z);
        }
    }
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
    CoeffStrided out = which == 0 ? CoeffStrided{g.ws.coeff_h + I, N} : CoeffStrided{g.ws.coeff_sig + I, g.ws.n_sig};
    chain_prepare_g2(which == 0 ? EMITJ(g, id, off_prep_h, stride_prep_h) : EMIT(g, id, off_prep_sig), q, out);
}

// Miller loop + final exponentiation + is_one
__global__ __launch_bounds__(64) void k_pairing(Group g) {
    if (g.chain_prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win VALU issue arbitration against the streaming placement waves
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    LaneId id = lane_id(g, I);
    const uint64_t N = g.N;
    Fp pkx = ld_fp(g.ws.pkaff + I), pky = ld_fp(g.ws.pkaff + N + I);
    CoeffStrided ch = {g.ws.coeff_h + I, N};
    CoeffStrided cs = {g.ws.coeff_sig + I, g.ws.n_sig};
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
    if ((uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE >= g.N) return;  // a wave without instances (the scratch pre-warm launch)
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
    t.coeff_h = {g.ws.coeff_h + I, N};
    t.coeff_sig = {g.ws.coeff_sig + I, g.ws.n_sig};
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
// N+1-pair product (blsw_verify_multi_batch): one team per instance, K pairs per instance. `gs` is the per-signature view
// (N = instances), the per-pair values (prepare_g1(pk_j), line coefficients of H(m_j)) live at flat index I * K + j of the
// per-pair launch of n_h = N * K lanes.
struct TeamLanesMulti : TeamLanes<CoeffStrided> {
    const Fp* coeff_h_all;
    const Fp* pkaff;
    uint64_t n_h, flat0;
    BLSW_TEAM_DEV void load_coeff_sig(uint32_t k) {
        if (active) team_load_coeff_sig_lane(j, slots, coeff_sig, k);
        team_sync();
    }
    BLSW_TEAM_DEV void load_pair(uint32_t jp, uint32_t k) {
        if (active) {
            const uint64_t t = flat0 + jp;
            Fp px = fp_zero(), py = fp_zero();
            if (j == 5) px = ld_fp(pkaff + t);
            if (j == 4) py = ld_fp(pkaff + n_h + t);
            team_load_pair_lane(j, slots, CoeffStrided{const_cast<Fp*>(coeff_h_all) + t, n_h}, k, px, py);
        }
        team_sync();
    }
};
__global__ __launch_bounds__(64) void k_pairing_team_multi(Group gs, uint32_t K, uint64_t n_h) {
    __shared__ Fp2 lds[BLSW_TEAMS_PER_WAVE * TS_NSLOTS];
    const uint32_t team = threadIdx.x / 6, j = threadIdx.x % 6;
    const uint64_t I0 = (uint64_t)blockIdx.x * BLSW_TEAMS_PER_WAVE + team;
    const bool active = team < BLSW_TEAMS_PER_WAVE && I0 < gs.N;
    const uint64_t I = active ? I0 : 0;
    LaneId id = lane_id(gs, I);
    TeamLanesMulti t;
    t.slots = lds + (active ? team : 0) * TS_NSLOTS;
    t.j = j;
    t.active = active;
    t.coeff_h = {nullptr, 0};
    t.coeff_sig = {gs.ws.coeff_sig + I, gs.ws.n_sig};
    t.coeff_h_all = gs.ws.coeff_h;
    t.pkaff = gs.ws.pkaff;
    t.n_h = n_h;
    t.flat0 = I * K;
    t.e = EMIT(gs, id, off_miller);
    if (!active) t.e.base = nullptr;
    if (active && j == 0) team_st(t.slots, TS_XYC, {K_G1_GEN_NEG_Y(), fp_zero()});
    team_sync();
    Fp2 f = team_miller_multi(t, K);
    Emitter e_one = EMIT(gs, id, off_is_one);
    if (!active) e_one.base = nullptr;
    bool res = team_final_exp_is_one(t, f, e_one);
    int32_t* r = gs.desc[id.s].result;
    if (active && j == 0 && r) r[id.i] = res ? 1 : 0;
}
static void launch_pairing(const Group& g, const Modes& m, hipStream_t st) {
    if (!m.pairing_team)
        hipLaunchKernelGGL(k_pairing, dim3((unsigned)((g.N + 63) / 64)), dim3(64), 0, st, g);
    else
        hipLaunchKernelGGL(k_pairing_team, dim3((unsigned)((g.N + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, st, g);
}

// Digest of witness vectors (blsw_witness_digest): d[c] = sum_k mix64(w_k + (k + 1) * C_c) over the instance's u64 words.
// grid (chunks, n); 256 threads, each 16 bytes per iteration; block partial sums -> two atomics per block.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}
#define BLSW_DIGEST_ITERS 16
__global__ __launch_bounds__(256) void k_digest(const uint64_t* __restrict__ w, uint64_t stride, uint64_t n_words, uint64_t* __restrict__ digest) {
    const uint64_t inst = blockIdx.y;
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(w + inst * stride * 6);
    const uint64_t n_pairs = n_words / 2;  // n_witness * 6 is even
    uint64_t q = ((uint64_t)blockIdx.x * BLSW_DIGEST_ITERS) * 256 + threadIdx.x;
    uint64_t d0 = 0, d1 = 0;
#pragma unroll 4
    for (int it = 0; it < BLSW_DIGEST_ITERS; it++, q += 256) {
        if (q < n_pairs) {
            ulonglong2 v = src[q];
            const uint64_t k = 2 * q + 1;  // (index of v.x) + 1
            d0 += mix64(v.x + k * 0x9E3779B97F4A7C15ull) + mix64(v.y + (k + 1) * 0x9E3779B97F4A7C15ull);
            d1 += mix64(v.x + k * 0xC2B2AE3D27D4EB4Full) + mix64(v.y + (k + 1) * 0xC2B2AE3D27D4EB4Full);
        }
    }
    // wave reduction, then one atomic pair per wave
    for (int off = 32; off > 0; off >>= 1) {
        d0 += __shfl_down(d0, off, 64);
        d1 += __shfl_down(d1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(digest + inst * 2), (unsigned long long)d0);
        atomicAdd(reinterpret_cast<unsigned long long*>(digest + inst * 2 + 1), (unsigned long long)d1);
    }
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
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sign(uint64_t n, Workspace ws, const uint8_t* __restrict__ sk32, uint8_t* sig96, uint64_t* sig_xy, uint8_t* pk48,
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
            if (!fp2_is_zero(h.z)) {  // sig = sk * H(m): inlined Jacobian ladder (two waves per SIMD, as k_cofactor_values)
                const Fp2 zi = fp2_inv_inl(h.z);
                const Fp2 hx = fp2_mul_inl(h.x, zi), hy = fp2_mul_inl(h.y, zi);
                Jac2 acc = {fp2_one(), fp2_one(), fp2_zero()};
#pragma unroll 1
                for (int b = 254; b >= 0; b--) {
                    acc = v_dbl(acc);
                    if ((k[b >> 5] >> (b & 31)) & 1) acc = v_add_mixed(acc, hx, hy);
                }
                if (!fp2_is_zero(acc.z)) {
                    const Fp2 ai = fp2_inv_inl(acc.z), ai2 = v_sqr(ai);
                    x = fp2_mul_inl(acc.x, ai2);
                    y = fp2_mul_inl(acc.y, fp2_mul_inl(ai2, ai));
                    inf = false;
                }
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
        if (st == SIGN_OK) {  // pk = sk * g1, the same ladder over Fp
            const Fp gx = K_G1_GEN_X(), gy = fp_neg(K_G1_GEN_NEG_Y());
            Jac1v acc = {fp_one(), fp_one(), fp_zero()};
#pragma unroll 1
            for (int b = 254; b >= 0; b--) {
                acc = v1_dbl(acc);
                if ((k[b >> 5] >> (b & 31)) & 1) acc = v1_add_mixed(acc, gx, gy);
            }
            if (!fp_is_zero(acc.z)) {
                const Fp ai = fp_inv(acc.z), ai2 = fp_sqr(ai);
                x = fp_mul(acc.x, ai2);
                y = fp_mul(acc.y, fp_mul(ai2, ai));
                inf = false;
            }
        }
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

__global__ __launch_bounds__(64) void k_bench_fpmul32(uint32_t iters, uint32_t* out) {  // the 12 x 32-bit CIOS product (cross-check of fp_mul)
    Fp a = fp_one(), b = fp_one();
    a.l[0] ^= threadIdx.x + 1;
    b.l[1] ^= blockIdx.x + 1;
    for (uint32_t i = 0; i < iters; i++) {
        a = fp_mul32(a, b);
        b = fp_mul32(b, a);
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

// RAII: every ABI entry point of an engine runs on the engine's device and restores the caller's
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) hipSetDevice(prev);
    }
};

}  // namespace

// Execution engine. Batches ("steps") are SUBMITTED with their input/output pointers and processed in GROUPS of up
// to max_steps batches by one set of launches (N = steps * n lanes per chain kernel), which is what fills the chip:
// one batch of 1024 instances is only 16 waves per chain. Streams (the runtime backs the streams of ONE priority level
// with four hardware queues; streams that share a queue serialise, so the engine uses three levels and few streams):
//   engine-wide, high priority:  sha    : SHA witness bits of every group, in order (needs only msg)  -> ev_sha per group
//                                expand : k_sha_expand per step (bit -> Fp, 31 MB of the 34 MB per instance, the HBM-bound
//                                         kernel), as soon as the group's SHA bits exist: it never waits for the curve /
//                                         pairing chains, so a short job does not pay their latency before its HBM stream
//                                place  : k_place_field per step (staging -> its place around the SHA segment), once the
//                                         group's chains are done and the step's expansion has finished
//   per group buffer:   normal   main : sha_values -> map -> cofactor -> prepare(H) .......... -> pairing   -> ev_chains
//                       low      aux  : prepare(sig), g1_alloc, g2_alloc  (need only pk / sig)   -> ev_aux
// Field witnesses go to a staging area (coalesced stores). n_buffers group buffers rotate, so the next groups' chains
// overlap the previous groups' placement.
// Materialisation (expand + place of a step into its output) is a queue of jobs in submission order (pump): a free-running
// engine issues a group's jobs when the group is launched; in consumer mode (options.consumer_mode) a job waits until the
// consumer has released its output's previous user, so the 34 MB vectors exist only between expansion and consumption and
// the output ring can be smaller than a group. A step can also leave in compact form (its slices of the staging, copied).
#define BLSW_MAX_BUFFERS 32
#define BLSW_DEFAULT_EXPAND_VARIANT 0  // 384 x 8: the geometry that stays fast beside every chain build (profiles/r02_ab_fpmul_expand.txt)
#define BLSW_MAX_TIMED 1024
#define BLSW_MAX_CONSUMED 64
struct GroupBuf {
    void* base = nullptr;
    StepDesc* h_desc = nullptr;  // pinned host
    StepDesc* d_desc = nullptr;
    hipStream_t st[2] = {nullptr, nullptr};  // main, aux
    hipEvent_t ev_start = nullptr, ev_aux = nullptr, ev_sha = nullptr, ev_chains = nullptr, ev_done = nullptr;
    hipEvent_t* ev_in = nullptr;    // [max_steps] inputs of step s valid (recorded on the submitting stream)
    hipEvent_t* ev_x = nullptr;     // [max_steps] expansion of step s issued and finished
    hipEvent_t* ev_step = nullptr;  // [max_steps] step s complete (witness tensor + results)
    uint64_t first_seq = 0;
    uint32_t steps = 0;
    bool used = false;
    Workspace ws;            // of the group launched last from this buffer
    uint32_t jobs_left = 0;  // its steps whose expansion / placement has not been issued yet
};
struct Job {  // materialisation of one step: bit expansion + field placement into its output (or its compact form)
    int buf;
    uint32_t s;
};
struct blsw_engine {
    uint64_t n = 0;
    uint32_t msg_len = 0, max_steps = 0;
    blsw_layout_t L, LS;
    Modes modes = DEFAULT_MODES;
    blsw_engine_options_t opt;
    int device = -1;
    GroupBuf buf[BLSW_MAX_BUFFERS];
    int nbuf = 0;
    int cur = 0;
    uint32_t pending = 0;
    uint64_t submitted = 0, launched = 0, materialised = 0;
    std::deque<Job> jobs;  // steps whose chains are issued, in submission order, waiting for their output to be free (consumer mode)
    hipStream_t sha = nullptr, expand = nullptr, place = nullptr;
    // HIP event pairs around every k_sha_expand launch since the last stats reset (live roofline measurement)
    hipEvent_t* ev_exp = nullptr;  // 2 * BLSW_MAX_TIMED events
    uint32_t n_timed = 0;
    // consumer releases: output tensor pointer -> event after which it may be overwritten
    const void* consumed_ptr[BLSW_MAX_CONSUMED];
    hipEvent_t consumed_ev[BLSW_MAX_CONSUMED];
    bool consumed_live[BLSW_MAX_CONSUMED];  // a release has been recorded and not yet waited for
    bool held[BLSW_MAX_CONSUMED];           // consumer mode: a step was materialised into this output and it has not been released
    bool staged = false;  // false: direct mode (max_steps == 1, no staging; witnesses written in place by the chains)
};

static void engine_free(blsw_engine* e) {
    if (!e) return;
    for (int k = 0; k < BLSW_MAX_BUFFERS; k++) {
        GroupBuf& b = e->buf[k];
        if (b.h_desc) hipHostFree(b.h_desc);
        if (b.d_desc) hipFree(b.d_desc);
        for (int i = 0; i < 2; i++)
            if (b.st[i]) hipStreamDestroy(b.st[i]);
        hipEvent_t single[] = {b.ev_start, b.ev_aux, b.ev_sha, b.ev_chains, b.ev_done};
        for (hipEvent_t ev : single)
            if (ev) hipEventDestroy(ev);
        hipEvent_t* arrays[] = {b.ev_in, b.ev_x, b.ev_step};
        for (hipEvent_t* arr : arrays) {
            if (!arr) continue;
            for (uint32_t s = 0; s < e->max_steps; s++)
                if (arr[s]) hipEventDestroy(arr[s]);
            delete[] arr;
        }
    }
    if (e->sha) hipStreamDestroy(e->sha);
    if (e->expand) hipStreamDestroy(e->expand);
    if (e->place) hipStreamDestroy(e->place);
    if (e->ev_exp) {
        for (int i = 0; i < 2 * BLSW_MAX_TIMED; i++)
            if (e->ev_exp[i]) hipEventDestroy(e->ev_exp[i]);
        delete[] e->ev_exp;
    }
    for (int i = 0; i < BLSW_MAX_CONSUMED; i++)
        if (e->consumed_ev[i]) hipEventDestroy(e->consumed_ev[i]);
    delete e;
}

// options.output_form = 1: the field witnesses of a step, in place, from Montgomery form to canonical integers (what
// CanonicalSerialize writes for an Fq: 48 bytes little-endian); the SHA segment is written in that form by the expansion itself
__global__ __launch_bounds__(256) void k_canonical_rows(uint64_t* __restrict__ d_witness, uint64_t stride, uint32_t off_expand, uint32_t sha_bits, uint32_t rows) {
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows) return;
    Fp* p = reinterpret_cast<Fp*>(d_witness + ((uint64_t)blockIdx.y * stride + (idx < off_expand ? idx : idx + sha_bits)) * 6);
    st_fp(p, fp_to_canonical(ld_fp(p)));
}
// field witnesses of one step: staged rows (lanes first .. first + n of the tiles at `staging` / the rows at `pair`) -> their
// places around the SHA segment of the step's witness vectors
static void launch_place(blsw_engine* e, hipStream_t st, const Fp* staging, const Fp* pair, uint32_t split_row, uint64_t first, uint64_t* out, uint64_t out_stride) {
    const uint32_t rows = e->L.n_witness - e->L.sha_bits;
    const unsigned chunks = (rows * 3 + 256 * BLSW_PLACE_ITERS - 1) / (256 * BLSW_PLACE_ITERS);
    dim3 grid2(8 * ((chunks + 7) / 8) * (unsigned)e->n);
    hipLaunchKernelGGL(k_place_field, grid2, dim3(256), 0, st, staging, pair, first, e->L.off_expand, e->L.sha_bits, rows, split_row, out, out_stride, (uint32_t)e->n,
                       e->L.off_sig_alloc, e->modes.g2_team ? e->L.off_pk_not_zero - e->L.off_sig_alloc : 0u, e->LS.off_sig_alloc);
}
static void launch_canonical(blsw_engine* e, hipStream_t st, uint64_t* out, uint64_t out_stride) {
    const uint32_t rows = e->L.n_witness - e->L.sha_bits;
    hipLaunchKernelGGL(k_canonical_rows, dim3((rows + 255) / 256, (unsigned)e->n), dim3(256), 0, st, out, out_stride, e->L.off_expand, e->L.sha_bits, rows);
}

#ifdef BLSW_DEBUG_KNOBS  // timing experiments only (wrong witnesses): BLSW_DEBUG_SKIP bit 0 chains, bit 1 placement, bit 2 expansion
static const uint32_t dbg_skip = getenv("BLSW_DEBUG_SKIP") ? (uint32_t)atoi(getenv("BLSW_DEBUG_SKIP")) : 0u;
#else
constexpr uint32_t dbg_skip = 0;
#endif

static int consumed_slot(blsw_engine* e, const void* ptr) {
    for (int c = 0; c < BLSW_MAX_CONSUMED; c++)
        if (e->consumed_ptr[c] == ptr && (e->consumed_live[c] || e->held[c])) return c;
    return -1;
}
// a consumer's release of an output (blsw_engine_output_consumed): the stream that is about to overwrite it waits for it
static void wait_released(blsw_engine* e, hipStream_t stream, const void* ptr) {
    const int c = consumed_slot(e, ptr);
    if (c >= 0 && e->consumed_live[c]) {
        hipStreamWaitEvent(stream, e->consumed_ev[c], 0);
        e->consumed_live[c] = false;
    }
}
// Issues the expansion (expansion stream: needs the group's SHA bits) and the field placement (placement stream: needs the
// group's chains) of step s of buffer k. A step is complete after both.
static void materialise(blsw_engine* e, int k, uint32_t s) {
    GroupBuf& b = e->buf[k];
    const StepDesc& d = b.h_desc[s];
    const Workspace& ws = b.ws;
    const CompactForm cf = compact_form(e->n, ws);
    hipStreamWaitEvent(e->expand, b.ev_sha, 0);
    if (d.compact) {  // the step's bit words leave as they are
        wait_released(e, e->expand, d.compact);
        hipMemcpyAsync(d.compact, ws.bits + (uint64_t)s * (e->n / 64) * bits_tile_words(ws.sha_words), cf.bits_bytes, hipMemcpyDeviceToDevice, e->expand);
    }
    if (d.out) {
        wait_released(e, e->expand, d.out);
        const bool timed = e->n_timed < BLSW_MAX_TIMED;
        if (timed) hipEventRecord(e->ev_exp[2 * e->n_timed], e->expand);
        ExpandArgs xa = {ws.bits, ws.sha_words, (uint64_t)s * e->n, e->L.sha_bits, e->L.off_expand, d.out, d.out_stride, 1u, 0u, 0, (int)e->opt.output_form};
        if (!(dbg_skip & 4)) launch_expand(e->opt.expand_variant, e->opt.expand_store, e->opt.place_lds, e->expand, xa, (unsigned)e->n);
        if (timed) {
            hipEventRecord(e->ev_exp[2 * e->n_timed + 1], e->expand);
            e->n_timed++;
        }
    }
    hipEventRecord(b.ev_x[s], e->expand);
    hipStreamWaitEvent(e->place, b.ev_chains, 0);
    hipStreamWaitEvent(e->place, b.ev_x[s], 0);
    if (d.compact) {  // and so do its staged field witnesses
        char* dst = reinterpret_cast<char*>(d.compact);
        hipMemcpyAsync(dst + cf.off_staging, ws.staging + (uint64_t)s * (e->n / 64) * ws.split_row * 64, cf.staging_bytes, hipMemcpyDeviceToDevice, e->place);
        if (cf.pair_bytes) hipMemcpyAsync(dst + cf.off_pair, ws.pair + (uint64_t)s * e->n * ws.pair_rows, cf.pair_bytes, hipMemcpyDeviceToDevice, e->place);
    }
    if (d.out && e->staged && !(dbg_skip & 2)) launch_place(e, e->place, ws.staging, ws.pair, ws.split_row, (uint64_t)s * e->n, d.out, d.out_stride);
    if (d.out && e->opt.output_form) launch_canonical(e, e->place, d.out, d.out_stride);  // direct mode: the chains wrote the rows in place
    hipEventRecord(b.ev_step[s], e->place);
}
// Materialises queued steps in submission order. Free-running engines (consumer_mode 0) issue every step as soon as its
// group is launched; in consumer mode a step whose output still holds an unreleased earlier step stops the queue until
// blsw_engine_output_consumed names that output.
static int pump(blsw_engine* e) {
    while (!e->jobs.empty()) {
        const Job j = e->jobs.front();
        GroupBuf& b = e->buf[j.buf];
        const StepDesc& d = b.h_desc[j.s];
        const void* ptr = d.out ? static_cast<const void*>(d.out) : d.compact;
        const bool track = e->opt.consumer_mode && e->staged && ptr;
        if (track) {
            const int c = consumed_slot(e, ptr);
            if (c >= 0 && e->held[c]) break;
        }
        materialise(e, j.buf, j.s);
        if (track) {
            int c = consumed_slot(e, ptr);
            for (int i = 0; i < BLSW_MAX_CONSUMED && c < 0; i++)
                if (!e->consumed_live[i] && !e->held[i]) c = i;
            if (c < 0) return BLSW_ERR_ARG;  // more than BLSW_MAX_CONSUMED outputs in use
            e->consumed_ptr[c] = ptr;
            e->held[c] = true;
        }
        e->jobs.pop_front();
        e->materialised++;
        if (--b.jobs_left == 0) hipEventRecord(b.ev_done, e->place);
    }
    return hip_ok(hipGetLastError(), "materialise");
}

static int launch_group(blsw_engine* e) {
    GroupBuf& b = e->buf[e->cur];
    const uint32_t steps = e->pending;
    if (steps == 0) return BLSW_OK;
    Group g;
    g.N = (uint64_t)steps * e->n;
    g.n = (uint32_t)e->n;
    g.K = 1;
    g.msg_len = e->msg_len;
    g.desc = b.d_desc;
    g.L = e->L;
    g.LS = e->LS;
    g.ws = carve(b.base, g.N, e->L, e->staged, e->modes);
    g.chain_prio = e->opt.prio_mode == 0;
    const unsigned g1 = (unsigned)((g.N + 63) / 64), g2 = (unsigned)((2 * g.N + 63) / 64);
    const unsigned gt = (unsigned)((g.N + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE);
    hipStream_t st = b.st[0];
    // inputs of every step are ready once its submitting stream reached the point of the submit
    for (uint32_t s = 0; s < steps; s++) hipStreamWaitEvent(st, b.ev_in[s], 0);
    bool any_out = false;
    for (uint32_t s = 0; s < steps; s++) any_out = any_out || b.h_desc[s].out != nullptr || b.h_desc[s].compact != nullptr;
    // direct mode: the chains themselves write into the output tensors, so they wait for the consumer's release
    if (!e->staged)
        for (uint32_t s = 0; s < steps; s++)
            if (b.h_desc[s].out) wait_released(e, st, b.h_desc[s].out);
    hipMemcpyAsync(b.d_desc, b.h_desc, sizeof(StepDesc) * steps, hipMemcpyHostToDevice, st);
    hipEventRecord(b.ev_start, st);
    hipStreamWaitEvent(b.st[1], b.ev_start, 0);
    // sha: the witness bits of the in-circuit SHA-256 (first: the expansion stream is waiting for them)
    hipStreamWaitEvent(e->sha, b.ev_start, 0);
    if (any_out) hipLaunchKernelGGL(k_sha, dim3(g1), dim3(64), 0, e->sha, g, 1, 0);
    hipEventRecord(b.ev_sha, e->sha);
    // main, first part: the hash-to-G2 critical path
    if (!(dbg_skip & 1)) {
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_map, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_prepare, dim3(g1), dim3(64), 0, st, g, 0);
    // aux: prepare_g2(sig) and the group allocations (53 ms alone beside the 86 ms of the main stream's first part)
    hipLaunchKernelGGL(k_prepare, dim3(g1), dim3(64), 0, b.st[1], g, 1);
    if (e->L.n_keys) {  // aggregate_verify: one lane per (instance, key) allocates, then mapped_aggregate + pk != 0 + prepare_g1 per instance
        hipLaunchKernelGGL(k_agg_keys, dim3((unsigned)((g.N * e->L.n_keys + 63) / 64)), dim3(64), 0, b.st[1], g, g.ws.keyproj);
        hipLaunchKernelGGL(k_agg_sum, dim3(g1), dim3(64), 0, b.st[1], g, (const Fp*)g.ws.keyproj);
    } else
        hipLaunchKernelGGL(k_g1, dim3(g1), dim3(64), 0, b.st[1], g);
    if (e->modes.g2_team)
        hipLaunchKernelGGL(k_g2_alloc_team, dim3(gt), dim3(64), 0, b.st[1], g);
    else
        hipLaunchKernelGGL(k_g2_alloc, dim3(g1), dim3(64), 0, b.st[1], g);
    }
    hipEventRecord(b.ev_aux, b.st[1]);
    // main, second part: the pairing
    hipStreamWaitEvent(st, b.ev_aux, 0);
    if (!(dbg_skip & 1)) launch_pairing(g, e->modes, st);
    hipStreamWaitEvent(st, b.ev_sha, 0);
    hipEventRecord(b.ev_chains, st);
    // expansion + placement of the group's steps: queued, issued in submission order (at once unless a consumer holds an output)
    b.ws = g.ws;
    b.jobs_left = steps;
    for (uint32_t s = 0; s < steps; s++) e->jobs.push_back({e->cur, s});
    b.used = true;
    b.first_seq = e->launched;
    b.steps = steps;
    e->launched += steps;
    e->pending = 0;
    e->cur = (e->cur + 1) % e->nbuf;
    if (hip_ok(hipGetLastError(), "launch")) return BLSW_ERR_HIP;
    return pump(e);
}

static uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* s = getenv(name);
    return s && *s ? (uint32_t)strtoul(s, nullptr, 10) : dflt;
}

extern "C" {

int blsw_version(void) { return BLSW_ABI_VERSION; }

int blsw_layout(uint32_t msg_len, blsw_layout_t* out) {
    if (!out || msg_len > 65535) return BLSW_ERR_ARG;
    make_layout(msg_len, out);
    return BLSW_OK;
}

int blsw_engine_options_default(blsw_engine_options_t* o) {
    if (!o) return BLSW_ERR_ARG;
    const char* p = getenv("BLSW_PAIRING");
    const char* g2 = getenv("BLSW_G2");
    o->device = -1;
    o->n_keys = 0;
    o->pairing_mode = (p && p[0] == 'l') ? 1u : 0u;
    o->g2_mode = (g2 && g2[0] == 't' && o->pairing_mode == 0) ? 1u : 0u;
    o->expand_variant = env_u32("BLSW_EXPAND_VARIANT", BLSW_DEFAULT_EXPAND_VARIANT);
    o->expand_store = env_u32("BLSW_EXPAND_NT", 0);  // plain stores: nontemporal ones cost 8-10 % since the chains' stack traffic was cut
    o->prio_mode = env_u32("BLSW_PRIO_MODE", 1);
    o->place_lds = env_u32("BLSW_PLACE_LDS", 0);
    o->consumer_mode = 0;
    o->output_form = 0;
    return BLSW_OK;
}

int blsw_engine_workspace_bytes_ex(uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, const blsw_engine_options_t* options, uint64_t* bytes) {
    if (!bytes || n == 0 || max_steps == 0 || n_buffers == 0 || n_buffers > BLSW_MAX_BUFFERS || msg_len > 65535 || !options || options->n_keys > 65535)
        return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L, options->n_keys);
    const bool staged = max_steps > 1 || n_buffers > 1;
    // the same workspace serves every kernel variant: the largest carve of the three mode combinations
    uint64_t need = 0;
    const Modes all[3] = {{true, false}, {true, true}, {false, false}};
    for (const Modes& m : all) {
        uint64_t t = carve(nullptr, n * max_steps, L, staged, m).total_bytes;
        need = t > need ? t : need;
    }
    *bytes = (uint64_t)n_buffers * align_up(need, 4096);
    return BLSW_OK;
}
int blsw_engine_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, uint64_t* bytes) {
    blsw_engine_options_t o;
    blsw_engine_options_default(&o);
    return blsw_engine_workspace_bytes_ex(n, msg_len, max_steps, n_buffers, &o, bytes);
}

int blsw_engine_create_ex(blsw_engine_t** out, uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, const blsw_engine_options_t* options,
                          void* d_workspace, uint64_t workspace_bytes) {
    if (!out || n == 0 || n > 0x7fffffffu || max_steps == 0 || !d_workspace || n_buffers == 0 || n_buffers > BLSW_MAX_BUFFERS || !options || msg_len > 65535)
        return BLSW_ERR_ARG;
    if (options->pairing_mode > 1 || options->g2_mode > 1 || (options->g2_mode == 1 && options->pairing_mode != 0) || options->expand_store > 3 ||
        options->prio_mode > 2 || options->output_form > 1 || (options->expand_variant & 0xff) > 5 || (options->expand_variant >> 9) || options->n_keys > 65535 ||
        (options->n_keys && options->g2_mode))
        return BLSW_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return BLSW_ERR_NO_DEVICE;
    int dev = options->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return BLSW_ERR_NO_DEVICE;
    if (dev >= ndev) return BLSW_ERR_ARG;
    DeviceGuard guard(dev);
    uint64_t need = 0;
    if (blsw_engine_workspace_bytes_ex(n, msg_len, max_steps, n_buffers, options, &need)) return BLSW_ERR_ARG;
    if (workspace_bytes < need) return BLSW_ERR_WORKSPACE;
    // Scratch guard. ROCr backs a queue's scratch for full-device occupancy: stack bytes per lane x 64 lanes x wave slots
    // (CUs x 32), per queue that runs the kernel. The single-lane pairing kernel (9.7 KB of stack) on four or more group
    // buffers made the runtime abort with HSA_STATUS_ERROR_OUT_OF_RESOURCES; refuse instead.
    if (options->pairing_mode == 1) {
        hipFuncAttributes fa;
        int cus = 256;
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        uint64_t stack = 10240;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_pairing)) == hipSuccess && fa.localSizeBytes) stack = fa.localSizeBytes;
        const uint64_t projected = stack * 64ull * (uint64_t)cus * 32ull * n_buffers;
        if (projected > (16ull << 30)) {
            fprintf(stderr, "[blsw] pairing_mode 1 with %u group buffers needs about %.1f GB of per-queue scratch: refused (use pairing_mode 0 or n_buffers <= 3)\n",
                    n_buffers, projected / 1e9);
            return BLSW_ERR_SCRATCH;
        }
    }
    blsw_engine* e = new blsw_engine();
    e->n = n;
    e->msg_len = msg_len;
    e->max_steps = max_steps;
    e->opt = *options;
    e->opt.device = dev;
    e->device = dev;
    e->modes = {options->pairing_mode == 0, options->g2_mode == 1};
    e->staged = max_steps > 1 || n_buffers > 1;
    make_layout(msg_len, &e->L, options->n_keys);
    e->LS = staging_layout(e->L, e->modes);
    for (int i = 0; i < BLSW_MAX_CONSUMED; i++) {
        e->consumed_ptr[i] = nullptr;
        e->consumed_ev[i] = nullptr;
        e->consumed_live[i] = false;
        e->held[i] = false;
    }
    e->nbuf = (int)n_buffers;
    int rc = BLSW_OK;
    auto chk = [&](hipError_t err, const char* what) {
        if (rc == BLSW_OK && hip_ok(err, what)) rc = BLSW_ERR_HIP;
        return rc == BLSW_OK;
    };
    e->ev_exp = new hipEvent_t[2 * BLSW_MAX_TIMED]();
    for (int i = 0; i < 2 * BLSW_MAX_TIMED && rc == BLSW_OK; i++) chk(hipEventCreate(&e->ev_exp[i]), "event create");
    int prio_lo = 0, prio_hi = 0;
    chk(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi), "priority range");  // (least, greatest): numerically lower = higher priority
    // three priority levels = three pools of hardware queues: [placement, main chains, aux chains] from high to low
    // (prio_mode 1, default), [main, aux, placement] (prio_mode 0) or everything at the middle level (prio_mode 2)
    const int prio_mid = (prio_hi + 1 <= prio_lo) ? prio_hi + 1 : prio_lo;  // -1 high, 0 normal, 1 low on this runtime
    int place_prio = prio_hi, main_prio = prio_mid, aux_prio = prio_lo;
    if (e->opt.prio_mode == 0) place_prio = prio_lo, main_prio = prio_hi, aux_prio = prio_mid;
    if (e->opt.prio_mode == 2) place_prio = main_prio = aux_prio = prio_mid;
    for (int k = 0; k < e->nbuf && rc == BLSW_OK; k++) {
        GroupBuf& b = e->buf[k];
        b.base = reinterpret_cast<char*>(d_workspace) + (uint64_t)k * (need / e->nbuf);
        chk(hipHostMalloc(reinterpret_cast<void**>(&b.h_desc), sizeof(StepDesc) * max_steps, hipHostMallocDefault), "host alloc");
        chk(hipMalloc(reinterpret_cast<void**>(&b.d_desc), sizeof(StepDesc) * max_steps), "desc alloc");
        chk(hipStreamCreateWithPriority(&b.st[0], hipStreamNonBlocking, main_prio), "stream create");
        chk(hipStreamCreateWithPriority(&b.st[1], hipStreamNonBlocking, aux_prio), "stream create");
        hipEvent_t* single[] = {&b.ev_start, &b.ev_aux, &b.ev_sha, &b.ev_chains, &b.ev_done};
        for (hipEvent_t* ev : single) chk(hipEventCreateWithFlags(ev, hipEventDisableTiming), "event create");
        b.ev_in = new hipEvent_t[max_steps]();
        b.ev_x = new hipEvent_t[max_steps]();
        b.ev_step = new hipEvent_t[max_steps]();
        for (uint32_t s = 0; s < max_steps && rc == BLSW_OK; s++) {
            chk(hipEventCreateWithFlags(&b.ev_in[s], hipEventDisableTiming), "event create");
            chk(hipEventCreateWithFlags(&b.ev_x[s], hipEventDisableTiming), "event create");
            chk(hipEventCreateWithFlags(&b.ev_step[s], hipEventDisableTiming), "event create");
        }
    }
    chk(hipStreamCreateWithPriority(&e->sha, hipStreamNonBlocking, place_prio), "stream create");
    chk(hipStreamCreateWithPriority(&e->expand, hipStreamNonBlocking, place_prio), "stream create");
    chk(hipStreamCreateWithPriority(&e->place, hipStreamNonBlocking, place_prio), "stream create");
    // Scratch pre-warm. The pairing kernel has the largest stack (4.5 KB per lane): the first launch of a full-size group on
    // a queue makes the runtime grow that queue's scratch, which stalls the queue for ~60 ms (measured: the pairing of the
    // first group of every buffer started 60 ms late). One launch of the same grid with N = 0 (every wave exits at once)
    // per main stream pays that here instead of in the caller's first groups.
    if (rc == BLSW_OK) {
        Group g0;
        memset(&g0, 0, sizeof(g0));
        g0.n = (uint32_t)n;
        g0.K = 1;
        const uint64_t Nmax = n * max_steps;
        for (int k = 0; k < e->nbuf; k++) {
            if (e->modes.pairing_team)
                hipLaunchKernelGGL(k_pairing_team, dim3((unsigned)((Nmax + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, e->buf[k].st[0], g0);
            else
                hipLaunchKernelGGL(k_pairing, dim3((unsigned)((Nmax + 63) / 64)), dim3(64), 0, e->buf[k].st[0], g0);
        }
        for (int k = 0; k < e->nbuf; k++) chk(hipStreamSynchronize(e->buf[k].st[0]), "scratch pre-warm");
    }
    if (rc != BLSW_OK) {
        engine_free(e);
        return rc;
    }
    *out = e;
    return BLSW_OK;
}

int blsw_engine_create(blsw_engine_t** out, uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, void* d_workspace,
                       uint64_t workspace_bytes) {
    blsw_engine_options_t o;
    blsw_engine_options_default(&o);
    return blsw_engine_create_ex(out, n, msg_len, max_steps, n_buffers, &o, d_workspace, workspace_bytes);
}

int blsw_engine_destroy(blsw_engine_t* e) {
    if (!e) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    hipDeviceSynchronize();
    engine_free(e);
    return BLSW_OK;
}

static int engine_submit(blsw_engine_t* e, const StepDesc& step, void* stream_) {
    if (step.out && step.out_stride < e->L.n_witness) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    GroupBuf& b = e->buf[e->cur];
    if (e->pending == 0 && b.used) {
        // the buffer's previous group must have been fully placed before its staging is overwritten; in consumer mode some of
        // its steps may still wait for their outputs: the caller has to drain (wait_step / output_consumed) first
        if (b.jobs_left) return BLSW_ERR_BUSY;
        if (hip_ok(hipEventSynchronize(b.ev_done), "event sync")) return BLSW_ERR_HIP;
        b.used = false;
    }
    if (hip_ok(hipEventRecord(b.ev_in[e->pending], reinterpret_cast<hipStream_t>(stream_)), "event record")) return BLSW_ERR_HIP;
    b.h_desc[e->pending] = step;
    e->pending++;
    e->submitted++;
    if (e->pending == e->max_steps) return launch_group(e);
    return BLSW_OK;
}
int blsw_engine_submit(blsw_engine_t* e, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, uint64_t* d_witness,
                       uint64_t witness_stride, int32_t* d_result, void* stream_) {
    if (!e || e->L.n_keys || !d_pk_xy || !d_sig_xy || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {d_pk_xy, d_sig_xy, d_msg, d_witness, witness_stride, d_result, nullptr, nullptr, nullptr, nullptr};
    return engine_submit(e, d, stream_);
}
// aggregate_verify through the engine (an engine created with options.n_keys = K): one batch of n instances of K keys each
int blsw_engine_submit_aggregate(blsw_engine_t* e, const uint64_t* d_pks_xy, const uint8_t* d_bitmap, const uint64_t* d_sig_xy, const uint8_t* d_msg,
                                 uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, uint32_t* d_count, void* stream_) {
    if (!e || !e->L.n_keys || !d_pks_xy || !d_bitmap || !d_sig_xy || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {nullptr, d_sig_xy, d_msg, d_witness, witness_stride, d_result, d_pks_xy, d_bitmap, d_count, nullptr};
    return engine_submit(e, d, stream_);
}

// Compact wire form (the all-gather of full witness vectors is capped by xGMI at a fraction of the generation
// rate; 2.6 MB per instance travel instead of 34 MB and the receiver expands them).
int blsw_engine_compact_bytes(blsw_engine_t* e, uint64_t* bytes) {
    if (!e || !bytes || !e->staged || e->n % 64) return BLSW_ERR_ARG;
    *bytes = compact_form(e->n, carve(nullptr, e->n, e->L, true, e->modes)).total;
    return BLSW_OK;
}
int blsw_engine_submit_compact(blsw_engine_t* e, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, void* d_compact, int32_t* d_result,
                               void* stream_) {
    if (!e || e->L.n_keys || !e->staged || e->n % 64 || !d_compact || !d_pk_xy || !d_sig_xy || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {d_pk_xy, d_sig_xy, d_msg, nullptr, 0, d_result, nullptr, nullptr, nullptr, d_compact};
    return engine_submit(e, d, stream_);
}
int blsw_engine_submit_aggregate_compact(blsw_engine_t* e, const uint64_t* d_pks_xy, const uint8_t* d_bitmap, const uint64_t* d_sig_xy, const uint8_t* d_msg,
                                         void* d_compact, int32_t* d_result, uint32_t* d_count, void* stream_) {
    if (!e || !e->L.n_keys || !e->staged || e->n % 64 || !d_compact || !d_pks_xy || !d_bitmap || !d_sig_xy || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {nullptr, d_sig_xy, d_msg, nullptr, 0, d_result, d_pks_xy, d_bitmap, d_count, d_compact};
    return engine_submit(e, d, stream_);
}
// receiver side: one batch in compact form -> its n witness vectors, on `stream` (the expansion and placement kernels of the
// engine's own steps, pointed at the compact buffer)
int blsw_engine_expand_compact(blsw_engine_t* e, const void* d_compact, uint64_t* d_witness, uint64_t witness_stride, void* stream_) {
    if (!e || !e->staged || e->n % 64 || !d_compact || !d_witness || witness_stride < e->L.n_witness) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    const Workspace w = carve(nullptr, e->n, e->L, true, e->modes);
    const CompactForm cf = compact_form(e->n, w);
    const char* src = reinterpret_cast<const char*>(d_compact);
    ExpandArgs xa = {reinterpret_cast<const uint32_t*>(src), w.sha_words, 0, e->L.sha_bits, e->L.off_expand, d_witness, witness_stride, 1u, 0u, 0, (int)e->opt.output_form};
    launch_expand(e->opt.expand_variant, e->opt.expand_store, e->opt.place_lds, st, xa, (unsigned)e->n);
    launch_place(e, st, reinterpret_cast<const Fp*>(src + cf.off_staging), reinterpret_cast<const Fp*>(src + cf.off_pair), w.split_row, 0, d_witness, witness_stride);
    if (e->opt.output_form) launch_canonical(e, st, d_witness, witness_stride);
    return hip_ok(hipGetLastError(), "expand compact");
}

// launches whatever is pending and makes `stream` wait for every group issued so far
int blsw_engine_flush(blsw_engine_t* e, void* stream_) {
    if (!e) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    int rc = launch_group(e);
    if (rc) return rc;
    if ((rc = pump(e))) return rc;
    for (int k = 0; k < e->nbuf; k++)
        if (e->buf[k].used && e->buf[k].jobs_left == 0) hipStreamWaitEvent(st, e->buf[k].ev_done, 0);
    return hip_ok(hipGetLastError(), "flush");
}

int blsw_engine_submitted(blsw_engine_t* e, uint64_t* seq) {
    if (!e || !seq) return BLSW_ERR_ARG;
    *seq = e->submitted;
    return BLSW_OK;
}
int blsw_engine_launched(blsw_engine_t* e, uint64_t* seq) {
    if (!e || !seq) return BLSW_ERR_ARG;
    *seq = e->launched;
    return BLSW_OK;
}
int blsw_engine_materialised(blsw_engine_t* e, uint64_t* seq) {
    if (!e || !seq) return BLSW_ERR_ARG;
    *seq = e->materialised;
    return BLSW_OK;
}
// Step `seq` must have been issued (seq < launched) and its group buffer not yet recycled (at most n_buffers groups back:
// older steps completed before their buffer was reused, so there is nothing to wait for).
int blsw_engine_wait_step(blsw_engine_t* e, uint64_t seq, void* stream_) {
    if (!e || seq >= e->launched) return BLSW_ERR_ARG;
    if (seq >= e->materialised) return BLSW_ERR_BUSY;  // consumer mode: its output is still held by an earlier step
    DeviceGuard guard(e->device);
    for (int k = 0; k < e->nbuf; k++) {
        GroupBuf& b = e->buf[k];
        if (b.used && seq >= b.first_seq && seq < b.first_seq + b.steps)
            return hip_ok(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream_), b.ev_step[seq - b.first_seq], 0), "wait step");
    }
    return BLSW_OK;
}
int blsw_engine_output_consumed(blsw_engine_t* e, const void* d_output, void* stream_) {
    if (!e || !d_output) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    int slot = consumed_slot(e, d_output);
    for (int c = 0; c < BLSW_MAX_CONSUMED && slot < 0; c++)
        if (!e->consumed_live[c] && !e->held[c]) slot = c;
    if (slot < 0) return BLSW_ERR_ARG;  // more than BLSW_MAX_CONSUMED distinct outputs in use
    if (!e->consumed_ev[slot] && hip_ok(hipEventCreateWithFlags(&e->consumed_ev[slot], hipEventDisableTiming), "event create")) return BLSW_ERR_HIP;
    if (hip_ok(hipEventRecord(e->consumed_ev[slot], reinterpret_cast<hipStream_t>(stream_)), "event record")) return BLSW_ERR_HIP;
    e->consumed_ptr[slot] = d_output;
    e->consumed_live[slot] = true;
    e->held[slot] = false;
    return pump(e);  // consumer mode: steps that waited for this output go out now
}

// Average duration (ms) of the k_sha_expand launches issued since the last call (HIP events recorded on the stream the
// kernel ran on); blocks until they have finished, then resets the statistics. count may be 0.
int blsw_engine_expand_stats(blsw_engine_t* e, uint32_t* count, float* avg_ms) {
    if (!e || !count || !avg_ms) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
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

int blsw_witness_digest(const uint64_t* d_witness, uint64_t witness_stride, uint64_t n, uint32_t n_witness, uint64_t* d_digest, void* stream_) {
    if (!d_witness || !d_digest || n == 0 || n > 65535 || n_witness == 0 || witness_stride < n_witness) return BLSW_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    if (hip_ok(hipMemsetAsync(d_digest, 0, n * 2 * sizeof(uint64_t), st), "memset")) return BLSW_ERR_HIP;
    const uint64_t n_words = (uint64_t)n_witness * 6, per_block = 2ull * 256 * BLSW_DIGEST_ITERS;
    dim3 grid((unsigned)((n_words + per_block - 1) / per_block), (unsigned)n);
    hipLaunchKernelGGL(k_digest, grid, dim3(256), 0, st, d_witness, witness_stride, n_words, d_digest);
    return hip_ok(hipGetLastError(), "launch");
}

// one-step descriptor at the head of a caller workspace (direct-mode entry points); `h` is copied before returning
static int put_desc(StepDesc* d_desc, const StepDesc& h, hipStream_t st) {
    if (hip_ok(hipMemcpyAsync(d_desc, &h, sizeof(h), hipMemcpyHostToDevice, st), "memcpy")) return BLSW_ERR_HIP;
    return hip_ok(hipStreamSynchronize(st), "sync");  // `h` is a stack object
}
static Group direct_group(uint64_t n, uint32_t K, uint32_t msg_len, const blsw_layout_t& L, StepDesc* d_desc, const Workspace& ws) {
    Group g;
    g.N = n * K;
    g.n = (uint32_t)n;
    g.K = K;
    g.msg_len = msg_len;
    g.desc = d_desc;
    g.L = L;
    g.LS = L;
    g.ws = ws;
    g.chain_prio = 0;
    return g;
}

int blsw_hash_to_g2_batch(const uint8_t* d_msg, uint32_t msg_len, uint64_t n, uint64_t* d_out_affine, void* d_workspace, uint64_t workspace_bytes,
                          void* stream_) {
    if ((!d_msg && msg_len) || n == 0 || n > 0x7fffffffu || !d_workspace || !d_out_affine || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    // the step descriptor lives at the head of the workspace
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(d_workspace);
    Workspace ws = carve(reinterpret_cast<char*>(d_workspace) + 256, n, L, false, DEFAULT_MODES);
    if (ws.total_bytes + 256 > workspace_bytes) return BLSW_ERR_WORKSPACE;
    Group g = direct_group(n, 1, msg_len, L, d_desc, ws);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    StepDesc h = {nullptr, nullptr, d_msg, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (int rc = put_desc(d_desc, h, st)) return rc;
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64);
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_map_values, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_h_to_affine, dim3(g1), dim3(64), 0, st, n, g.ws, d_out_affine);
    return hip_ok(hipGetLastError(), "launch");
}
// BLS::sign + PublicKey::from(&sk) for a batch (bls.rs:411-425, 183-195). Workspace: blsw_hash_to_g2_workspace_bytes.
int blsw_sign_batch(const uint8_t* d_sk32_le, const uint8_t* d_msg, uint32_t msg_len, uint64_t n, uint8_t* d_sig96, uint64_t* d_sig_xy, uint8_t* d_pk48,
                    uint64_t* d_pk_xy, int32_t* d_status, void* d_workspace, uint64_t workspace_bytes, void* stream_) {
    if (!d_sk32_le || (!d_msg && msg_len) || n == 0 || n > 0x7fffffffu || !d_workspace || !d_status || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(d_workspace);
    Workspace ws = carve(reinterpret_cast<char*>(d_workspace) + 256, n, L, false, DEFAULT_MODES);
    if (ws.total_bytes + 256 > workspace_bytes) return BLSW_ERR_WORKSPACE;
    Group g = direct_group(n, 1, msg_len, L, d_desc, ws);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    StepDesc h = {nullptr, nullptr, d_msg, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (int rc = put_desc(d_desc, h, st)) return rc;
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64);
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_map_values, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor_values, dim3(g1), dim3(64), 0, st, g);
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
    return o + carve(nullptr, n, L, false, DEFAULT_MODES).total_bytes;
}
int blsw_aggregate_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t n_keys, uint64_t* bytes) {
    if (!bytes || n == 0 || n_keys == 0 || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L, n_keys);
    uint64_t a, b, c;
    *bytes = agg_workspace(n, L, &a, &b, &c);
    return BLSW_OK;
}
int blsw_aggregate_verify_batch(const uint64_t* d_pks_xy, const uint8_t* d_bitmap, uint32_t n_keys, const uint64_t* d_sig_xy, const uint8_t* d_msg,
                                uint32_t msg_len, uint64_t n, uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, uint32_t* d_count,
                                void* d_workspace, uint64_t workspace_bytes, void* stream_) {
    if (!d_pks_xy || !d_bitmap || n_keys == 0 || !d_sig_xy || (!d_msg && msg_len) || n == 0 || n > 65535 || !d_workspace || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L, n_keys);
    if (d_witness && witness_stride < L.n_witness) return BLSW_ERR_ARG;
    uint64_t off_desc, off_keyproj, off_ws;
    if (agg_workspace(n, L, &off_desc, &off_keyproj, &off_ws) > workspace_bytes) return BLSW_ERR_WORKSPACE;
    char* base = reinterpret_cast<char*>(d_workspace);
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(base + off_desc);
    Fp* keyproj = reinterpret_cast<Fp*>(base + off_keyproj);
    Group g = direct_group(n, 1, msg_len, L, d_desc, carve(base + off_ws, n, L, false, DEFAULT_MODES));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    StepDesc h = {nullptr, d_sig_xy, d_msg, d_witness, witness_stride, d_result, d_pks_xy, d_bitmap, d_count};
    if (int rc = put_desc(d_desc, h, st)) return rc;
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64), gk = (unsigned)((n * n_keys + 63) / 64);
    hipLaunchKernelGGL(k_agg_keys, dim3(gk), dim3(64), 0, st, g, keyproj);
    hipLaunchKernelGGL(k_agg_sum, dim3(g1), dim3(64), 0, st, g, (const Fp*)keyproj);
    hipLaunchKernelGGL(k_g2_alloc, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_prepare, dim3(g1), dim3(64), 0, st, g, 1);
    hipLaunchKernelGGL(k_sha, dim3(g1), dim3(64), 0, st, g, d_witness ? 1 : 0, 1);
    if (d_witness) {
        ExpandArgs xa = {g.ws.bits, g.ws.sha_words, 0, g.L.sha_bits, g.L.off_expand, d_witness, witness_stride, 1u, 0u, 0};
        launch_expand(BLSW_DEFAULT_EXPAND_VARIANT, 0, 0, st, xa, (unsigned)n);
    }
    hipLaunchKernelGGL(k_map, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_prepare, dim3(g1), dim3(64), 0, st, g, 0);
    launch_pairing(g, DEFAULT_MODES, st);
    return hip_ok(hipGetLastError(), "launch");
}

// ---- N+1-pair product of pairings (one signature over n_pairs (pk, msg) pairs)
int blsw_layout_multi(uint32_t msg_len, uint32_t n_pairs, blsw_layout_t* out) {
    if (!out || msg_len > 65535 || n_pairs == 0 || n_pairs > 4096) return BLSW_ERR_ARG;
    make_layout(msg_len, out, 0, n_pairs);
    // the witness vector must stay addressable with 32-bit element offsets
    const uint64_t total = (uint64_t)out->off_prep_h - out->off_expand;  // n_pairs * stride_hash
    if (total / n_pairs != out->stride_hash) return BLSW_ERR_ARG;
    return BLSW_OK;
}
int blsw_verify_multi_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t n_pairs, uint64_t* bytes) {
    blsw_layout_t L;
    if (!bytes || n == 0 || blsw_layout_multi(msg_len, n_pairs, &L)) return BLSW_ERR_ARG;
    *bytes = 256 + carve(nullptr, n * n_pairs, L, false, DEFAULT_MODES, n).total_bytes;
    return BLSW_OK;
}
int blsw_verify_multi_batch(const uint64_t* d_pks_xy, const uint8_t* d_msgs, uint32_t msg_len, uint32_t n_pairs, const uint64_t* d_sig_xy, uint64_t n,
                            uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, void* d_workspace, uint64_t workspace_bytes, void* stream_) {
    blsw_layout_t L;
    if (!d_pks_xy || (!d_msgs && msg_len) || !d_sig_xy || n == 0 || !d_workspace || blsw_layout_multi(msg_len, n_pairs, &L)) return BLSW_ERR_ARG;
    const uint64_t NP = n * n_pairs;  // per-pair lanes
    if (NP > 65535 * 16ull || n > 65535) return BLSW_ERR_ARG;
    if (d_witness && witness_stride < L.n_witness) return BLSW_ERR_ARG;
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(d_workspace);
    Workspace ws = carve(reinterpret_cast<char*>(d_workspace) + 256, NP, L, false, DEFAULT_MODES, n);
    if (ws.total_bytes + 256 > workspace_bytes) return BLSW_ERR_WORKSPACE;
    Group gp = direct_group(n, n_pairs, msg_len, L, d_desc, ws);  // per-pair work: N = n * n_pairs lanes
    Group gs = direct_group(n, 1, msg_len, L, d_desc, ws);        // per-signature work: N = n lanes
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    StepDesc h = {d_pks_xy, d_sig_xy, d_msgs, d_witness, witness_stride, d_result, nullptr, nullptr, nullptr};
    if (int rc = put_desc(d_desc, h, st)) return rc;
    const unsigned p1 = (unsigned)((NP + 63) / 64), p2 = (unsigned)((2 * NP + 63) / 64), s1 = (unsigned)((n + 63) / 64);
    // fork: the signature's allocation + prepare (one lane per instance: 57 ms of latency) and the keys' allocation run beside the
    // hash-to-G2 chains of the pairs; join in front of the Miller product. The two side streams and three events are created once
    // per host thread and device and kept (an event is re-recorded per call; a wait refers to the record that preceded it).
    struct Side {
        int device = -1;
        hipStream_t aux[2] = {nullptr, nullptr};
        hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
        bool ok = false;
    };
    static thread_local Side sides[16];
    int dev = 0;
    hipGetDevice(&dev);
    Side& sd = sides[dev & 15];
    if (sd.device != dev) {
        sd.device = dev;
        sd.ok = hipStreamCreateWithFlags(&sd.aux[0], hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&sd.aux[1], hipStreamNonBlocking) == hipSuccess &&
                hipEventCreateWithFlags(&sd.ev_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&sd.ev_join[0], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&sd.ev_join[1], hipEventDisableTiming) == hipSuccess;
    }
    const bool forked = sd.ok;
    hipEvent_t ev_fork = sd.ev_fork;
    hipEvent_t* ev_join = sd.ev_join;
    hipStream_t s_sig = forked ? sd.aux[0] : st, s_keys = forked ? sd.aux[1] : st;
    if (forked) {
        hipEventRecord(ev_fork, st);  // the descriptor copy
        hipStreamWaitEvent(s_sig, ev_fork, 0);
        hipStreamWaitEvent(s_keys, ev_fork, 0);
    }
    hipLaunchKernelGGL(k_g2_alloc, dim3(s1), dim3(64), 0, s_sig, gs);
    hipLaunchKernelGGL(k_prepare, dim3(s1), dim3(64), 0, s_sig, gs, 1);
    hipLaunchKernelGGL(k_g1, dim3(p1), dim3(64), 0, s_keys, gp);
    hipLaunchKernelGGL(k_sha, dim3(p1), dim3(64), 0, st, gp, d_witness ? 1 : 0, 1);
    if (d_witness) {
        // blockIdx.y = flat (instance, pair); grid.y <= 65535: several launches for larger batches
        const uint64_t per_launch = (65535 / n_pairs) * (uint64_t)n_pairs;
        for (uint64_t first = 0; first < NP; first += per_launch) {
            const uint64_t cnt = NP - first < per_launch ? NP - first : per_launch;
            ExpandArgs xa = {ws.bits, ws.sha_words, first, L.sha_bits, L.off_expand, d_witness + (first / n_pairs) * witness_stride * 6, witness_stride, n_pairs, L.stride_hash, 0};
            launch_expand(BLSW_DEFAULT_EXPAND_VARIANT, 0, 0, st, xa, (unsigned)cnt);
        }
    }
    hipLaunchKernelGGL(k_map, dim3(p2), dim3(64), 0, st, gp);
    hipLaunchKernelGGL(k_cofactor, dim3(p1), dim3(64), 0, st, gp);
    hipLaunchKernelGGL(k_prepare, dim3(p1), dim3(64), 0, st, gp, 0);
    if (forked) {
        hipEventRecord(ev_join[0], s_sig);
        hipEventRecord(ev_join[1], s_keys);
        hipStreamWaitEvent(st, ev_join[0], 0);
        hipStreamWaitEvent(st, ev_join[1], 0);
    }
    hipLaunchKernelGGL(k_pairing_team_multi, dim3((unsigned)((n + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, st, gs, n_pairs, NP);
    return hip_ok(hipGetLastError(), "launch");
}

int blsw_decode_batch(const uint8_t* d_pk48, const uint8_t* d_sig96, uint64_t n, uint64_t* d_pk_xy, uint64_t* d_sig_xy, int32_t* d_status, void* stream_) {
    if (!d_pk48 || !d_sig96 || !d_pk_xy || !d_sig_xy || !d_status || n == 0 || n > 0x3fffffffu) return BLSW_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_decode, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st, d_pk48, d_sig96, n, d_pk_xy, d_sig_xy, d_status);
    return hip_ok(hipGetLastError(), "launch");
}
int blsw_hash_to_g2_workspace_bytes(uint64_t n, uint32_t msg_len, uint64_t* bytes) {
    if (!bytes || n == 0 || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    *bytes = carve(nullptr, n, L, false, DEFAULT_MODES).total_bytes + 256;
    return BLSW_OK;
}

// which = 0: v_mad_u64_u32 issue rate (multiply-adds/s); 1: fp_mul (Fp products/s); 2: fp_inv (inversions/s); 3: Fp products/s inside
// witness-emitting Fp2 mul + sqr. Synchronous, on the current device.
int blsw_microbench(int which, uint32_t iters, uint32_t blocks, double* ops_per_s) {
    if (!ops_per_s || iters == 0 || blocks == 0 || which < 0 || which > 4) return BLSW_ERR_ARG;
    uint32_t* d = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = hip_ok(hipMalloc(&d, 4), "malloc");
    if (!rc) rc = hip_ok(hipEventCreate(&e0), "event create");
    if (!rc) rc = hip_ok(hipEventCreate(&e1), "event create");
    const int threads = which == 0 ? 256 : 64;
    const double per_iter[5] = {8.0, 2.0, 1.0, 5.0, 2.0};  // MADs, fp products, fp inversions, fp products (one Fp2 mul + one Fp2 sqr), fp products (32-bit CIOS)
    for (int rep = 0; rep < 2 && !rc; rep++) {        // first pass warms up
        hipEventRecord(e0, 0);
        if (which == 0)
            hipLaunchKernelGGL(k_bench_mad, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else if (which == 1)
            hipLaunchKernelGGL(k_bench_fpmul, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else if (which == 2)
            hipLaunchKernelGGL(k_bench_fpinv, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else if (which == 3)
            hipLaunchKernelGGL(k_bench_fp2mulw, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else
            hipLaunchKernelGGL(k_bench_fpmul32, dim3(blocks), dim3(threads), 0, 0, iters, d);
        hipEventRecord(e1, 0);
        rc = hip_ok(hipEventSynchronize(e1), "event sync");
    }
    if (!rc) {
        float ms = 0;
        rc = hip_ok(hipEventElapsedTime(&ms, e0, e1), "event elapsed");
        if (!rc) *ops_per_s = per_iter[which] * iters * blocks * threads / (ms * 1e-3);
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (d) hipFree(d);
    return rc;
}
}
