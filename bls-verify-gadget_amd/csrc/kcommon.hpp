// Shared by every translation unit of libblsw.so: the group / workspace / step descriptors, the witness cursor of the one-instance-per-lane
// kernels, and the declarations of the kernels (each family is compiled as its own translation unit, in parallel: build.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef BLSW_KVARIANT_QUAD  // the latency compilation of a chain unit: four lanes per item (fp.hpp: quads)
#define BLSW_QUAD 1
#endif
#include "chains.hpp"
#include "cofactor_vf.hpp"
#include "prepare_vf.hpp"
#include "layout.h"

namespace blsw {

#ifdef BLSW_KVARIANT_QUAD
#define BLSW_LPI 4u  // lanes per item (instance, pair, chunk) of this compilation's kernels
#else
#define BLSW_LPI 1u
#endif
// index of this thread's item, and whether it is the lane of its item that writes the item's outputs (values are identical on the lanes of a quad)
__device__ __forceinline__ uint64_t item_index() { return ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / BLSW_LPI; }
__device__ __forceinline__ bool item_leader() { return BLSW_LPI == 1u || (threadIdx.x & (BLSW_LPI - 1u)) == 0u; }
inline unsigned item_grid(uint64_t items, unsigned lpi) { return (unsigned)((items * lpi + 63) / 64); }

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
    Fp* staging;     // [N/64][rows_p][64] field witnesses (engine mode), or nullptr (direct mode): each wave of 64
                     // lanes owns one contiguous tile and appends 3 KiB rows to it (sequential HBM writes per wave)
    Fp* staging_inst;  // [n_sig/64][rows_i][64] the per-signature rows of the N+1-pair product staged the same way (sig allocation,
                       // prepare_g2(sig)); = staging (one lane is one instance) for the single-key and aggregate circuits
    uint32_t rows_p, rows_i;  // rows per tile of the two areas (= split_row when they are one)
    uint64_t staging_rows;
    Fp* pair;            // [n_sig][pair_rows]: rows >= split_row of the staging coordinates (Miller loop, final exponentiation,
    uint32_t split_row;  // is_one), instance-major: the six-lane pairing kernel appends each instance's segment sequentially
    uint32_t pair_rows;  // (split_row = staging_rows, pair_rows = 0 when the single-lane pairing kernel is in use)
    uint64_t sha_words;
    Fp* cofv;            // [BLSW_COFV_ELEMS][N] scratch of the values-first cofactor chain (cofactor_vf.hpp), or nullptr (N > BLSW_LATENCY_MAX_LANES)
    Fp* prepv_h;         // [BLSW_PREPV_ELEMS][N] / [BLSW_PREPV_ELEMS][n_sig] scratch of the values-first prepare chains (prepare_vf.hpp), or nullptr
    Fp* prepv_sig;
    uint64_t total_bytes;
};
// launch groups of at most this many lanes may take the latency kernels (quads, values-first cofactor chain): their scratch is carved for them
#define BLSW_LATENCY_MAX_LANES 8192
inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
BLSW_HD uint64_t bits_tile_words(uint64_t sha_words) { return sha_words * 64; }  // u32 per 64-instance tile
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
    if (L.off_prep_g1 > L.off_expand) S.off_prep_g1 = L.off_prep_g1 - L.sha_bits;  // off_params_alloc lies in front of the expansion
    if (m.g2_team) {
        const uint32_t lo = L.off_sig_alloc, len = L.off_pk_not_zero - L.off_sig_alloc;
        for (int k = 0; k < 15; k++)
            if (f[k] > lo) f[k] -= len;
        if (S.off_prep_g1 > lo) S.off_prep_g1 -= len;
        S.off_sig_alloc = L.n_witness - L.sha_bits - len;  // last rows of the staging coordinates
    }
    return S;
}
// Staging coordinates of the N+1-pair product in the grouped engine (K = L.n_pairs > 1): a PAIR lane stages the rows of its (pk, msg)
// pair — msg bits, key allocation, pk != 0, map0, map1, add, cofactor, prepare(H), prepare(pk) — in that order in its column of the
// pair tiles; an INSTANCE lane stages sig allocation and prepare(sig) in the instance tiles; Miller loop, final exponentiation and
// is_one are instance-major rows from `split` on. Strides are 0: a pair's copy is found by its lane, not by an offset.
struct MultiStaging {
    blsw_layout_t LS;
    uint32_t rows_p, rows_i, split, pair_rows;
};
inline MultiStaging staging_layout_multi(const blsw_layout_t& L) {
    MultiStaging m;
    m.LS = L;
    blsw_layout_t& S = m.LS;
    uint32_t o = 0;
    S.off_msg = o;
    o += L.stride_msg;
    S.off_pk_alloc = o;
    o += L.stride_pk_alloc;
    S.off_pk_not_zero = o;
    o += L.stride_pk_not_zero;
    S.off_map0 = o;
    S.off_map1 = o + (L.off_map1 - L.off_map0);
    S.off_add = o + (L.off_add - L.off_map0);
    S.off_cofactor = o + (L.off_cofactor - L.off_map0);
    o += L.stride_hash - L.sha_bits;
    S.off_prep_h = o;
    o += L.stride_prep_h;
    S.off_prep_pk = o;
    o += L.stride_prep_pk;
    m.rows_p = o;
    S.off_sig_alloc = 0;
    S.off_prep_sig = L.off_pk_not_zero - L.off_sig_alloc;  // after the sig allocation segment
    m.rows_i = S.off_prep_sig + (L.off_miller - L.off_prep_sig);
    m.split = m.rows_p > m.rows_i ? m.rows_p : m.rows_i;
    S.off_miller = m.split;
    S.off_final_exp = m.split + (L.off_final_exp - L.off_miller);
    S.off_is_one = m.split + (L.off_is_one - L.off_miller);
    m.pair_rows = L.n_witness - L.off_miller;
    S.off_expand = 0xffffffffu;  // never staged
    S.stride_msg = S.stride_pk_alloc = S.stride_pk_not_zero = S.stride_hash = S.stride_prep_h = S.stride_prep_pk = 0;
    return m;
}
// N lanes of per-(pk, msg) work, n_sig lanes of per-signature work (n_sig = N except for the N+1-pair product)
inline Workspace carve(void* base, uint64_t N, const blsw_layout_t& L, bool with_staging, const Modes& m, uint64_t n_sig = 0) {
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
    const bool small = N <= BLSW_LATENCY_MAX_LANES;  // (not a test of a carved pointer: with base == nullptr the carve only measures)
    w.cofv = small ? reinterpret_cast<Fp*>(take((uint64_t)BLSW_COFV_ELEMS * N * sizeof(Fp))) : nullptr;
    w.prepv_h = small ? reinterpret_cast<Fp*>(take((uint64_t)BLSW_PREPV_ELEMS * N * sizeof(Fp))) : nullptr;
    w.prepv_sig = small ? reinterpret_cast<Fp*>(take((uint64_t)BLSW_PREPV_ELEMS * n_sig * sizeof(Fp))) : nullptr;
    w.staging_rows = L.n_witness - L.sha_bits;
    if (L.n_pairs > 1 && with_staging) {  // N+1-pair product in the grouped engine: pair tiles, instance tiles, instance-major rows
        const MultiStaging ms = staging_layout_multi(L);
        w.rows_p = ms.rows_p;
        w.rows_i = ms.rows_i;
        w.split_row = ms.split;
        w.pair_rows = ms.pair_rows;
        w.staging = reinterpret_cast<Fp*>(take((uint64_t)w.rows_p * align_up(N, 64) * sizeof(Fp)));
        w.staging_inst = reinterpret_cast<Fp*>(take((uint64_t)w.rows_i * align_up(n_sig, 64) * sizeof(Fp)));
        w.pair = reinterpret_cast<Fp*>(take((uint64_t)w.pair_rows * n_sig * sizeof(Fp)));
        w.total_bytes = off;
        return w;
    }
    w.split_row = m.pairing_team ? staging_layout(L, m).off_miller : (uint32_t)w.staging_rows;
    w.pair_rows = (uint32_t)w.staging_rows - w.split_row;
    w.rows_p = w.rows_i = w.split_row;
    w.staging = with_staging ? reinterpret_cast<Fp*>(take((uint64_t)w.split_row * align_up(N, 64) * sizeof(Fp))) : nullptr;
    w.staging_inst = w.staging;
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
    // blsw_engine_submit_bytes: [n][2] decode statuses of (pk, sig); result[i] = gadget Boolean AND both statuses BLSW_ST_OK
    // (tests/tests.rs:244-263: a point that does not decode is replaced by the default and the case must verify false)
    const int32_t* status;
    // blsw_engine_submit_io: [n][n_instance_vars][6] instance_assignment of every instance (element 0 = one), or nullptr
    uint64_t* inst;
};
__device__ __forceinline__ int32_t step_result(const StepDesc& d, uint32_t i, bool res) {
    if (d.status && (d.status[2 * i] | d.status[2 * i + 1])) return 0;
    return res ? 1 : 0;
}
// Compact wire form of a step of n instances (n a multiple of 64): the step's slices of the group workspace, back to back —
// [n/64][sha_words/16][64][16] u32 bit words | [n/64][split_row][64] Fp tile-major rows | [n][pair_rows] Fp instance-major rows
// N+1-pair product (K pairs per instance, n K a multiple of 64 and n a divisor or a multiple of 64): bit words and pair tiles of the step's
// n K pair lanes | the instance tiles' rows of the step's n lanes, packed [rows_i][tile_w] with tile_w = min(n, 64) per tile | instance-major rows
struct CompactForm {
    uint64_t bits_bytes, staging_bytes, inst_bytes, pair_bytes, off_staging, off_inst, off_pair, total;
    uint32_t inst_tile_w;  // lanes per instance tile in the compact buffer (64, or n when a step is a part of one tile)
};
inline CompactForm compact_form(uint64_t n, const Workspace& w, uint32_t K = 1) {
    CompactForm c;
    c.bits_bytes = bits_tile_words(w.sha_words) * (n * K / 64) * 4;
    c.staging_bytes = (uint64_t)w.rows_p * n * K * sizeof(Fp);
    c.inst_bytes = K > 1 ? (uint64_t)w.rows_i * n * sizeof(Fp) : 0;
    c.inst_tile_w = n % 64 == 0 ? 64u : (uint32_t)n;
    c.pair_bytes = (uint64_t)w.pair_rows * n * sizeof(Fp);
    c.off_staging = align_up(c.bits_bytes, 256);
    c.off_inst = align_up(c.off_staging + c.staging_bytes, 256);
    c.off_pair = align_up(c.off_inst + c.inst_bytes, 256);
    c.total = align_up(c.off_pair + c.pair_bytes, 256);
    return c;
}
// can steps of n instances of K pairs leave in compact form (whole pair tiles; instance lanes = whole tiles or an aligned part of one)?
inline bool compact_shape_ok(uint64_t n, uint32_t K) { return K <= 1 ? n % 64 == 0 : ((n * K) % 64 == 0 && (n % 64 == 0 || 64 % n == 0)); }
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
    int canonical;         // instance_assignment elements as canonical integers (options.output_form 1) instead of Montgomery limbs
};
// workspace / staging / tensor accesses: global address space stated (see Emitter::put)
__device__ __forceinline__ Fp ld_fp(const Fp* p) {
    const blsw_global_u32x4* s = (const blsw_global_u32x4*)p;
    blsw_u32x4 a = s[0], b = s[1], c = s[2];
    Fp r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    return r;
}
__device__ __forceinline__ void st_fp(Fp* p, const Fp& v) {
    blsw_global_u32x4* d = (blsw_global_u32x4*)p;
    d[0] = blsw_u32x4{v.l[0], v.l[1], v.l[2], v.l[3]};
    d[1] = blsw_u32x4{v.l[4], v.l[5], v.l[6], v.l[7]};
    d[2] = blsw_u32x4{v.l[8], v.l[9], v.l[10], v.l[11]};
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
// element k of instance (step s, i)'s instance_assignment := v (blsw_engine_submit_io); the witness kernels write the inputs they allocate
__device__ __forceinline__ void put_instance(const Group& g, const LaneId& id, uint32_t k, const Fp& v) {
    uint64_t* base = g.desc[id.s].inst;
    if (!base) return;
    st_fp(reinterpret_cast<Fp*>(base) + ((uint64_t)id.i * g.L.n_instance_vars + k), g.canonical ? fp_to_canonical(v) : v);
}

// witness cursor for a segment: staging row (engine mode), the instance's dense vector (direct mode), or value-only
// per_pair: the segment belongs to a (pk, msg) pair (the lane is a pair lane) — else to the instance / signature
__device__ __forceinline__ Emitter emitter(const Group& g, const LaneId& id, uint32_t off_full, uint32_t off_staging, bool per_pair) {
    Emitter e;
    if (g.ws.staging) {
        if (!per_pair && off_staging >= g.ws.split_row) {  // instance-major rows of the pairing segments
            e.base = reinterpret_cast<uint32_t*>(g.ws.pair + id.I * g.ws.pair_rows);
            e.pos = off_staging - g.ws.split_row;
            e.stride = 12;
            return e;
        }
        Fp* tiles = per_pair ? g.ws.staging : g.ws.staging_inst;
        const uint32_t rows = per_pair ? g.ws.rows_p : g.ws.rows_i;
        e.base = reinterpret_cast<uint32_t*>(tiles + (id.I >> 6) * (uint64_t)rows * 64 + (id.I & 63));
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
#define EMIT(g, id, field) emitter(g, id, (g).L.field, (g).LS.field, false)
// segment that exists once per pair: in the vector pair j's copy starts j * stride further; in the staging a pair is a lane of its own
// (LS.stride = 0) unless the staging IS the vector (direct mode: LS = L)
#define EMITJ(g, id, field, stride) emitter(g, id, (g).L.field + (id).j * (g).L.stride, (g).LS.field + (id).j * (g).LS.stride, true)

__device__ __forceinline__ Proj<OpsFp2> ld_proj2(const Fp* p, uint64_t n) {
    Proj<OpsFp2> r;
    r.x = ld_fp2(p, n);
    r.y = ld_fp2(p + 2 * n, n);
    r.z = ld_fp2(p + 4 * n, n);
    return r;
}

// line coefficients, element-major: coefficient idx of instance I at p[idx * N]
struct CoeffStrided {
    Fp* p;
    uint64_t n;
    __device__ __forceinline__ void st(uint32_t idx, const Fp& v) const {
        if (item_leader()) st_fp(p + (uint64_t)idx * n, v);
    }
    __device__ __forceinline__ Fp ld(uint32_t idx) const { return ld_fp(p + (uint64_t)idx * n); }
};

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
#define BLSW_EXPAND_RESIDENT_WGS 512u  // expand_variant 13 (k_stream.hip): two 384-thread workgroups per compute unit

#define BLSW_TEAMS_PER_WAVE 10
#define BLSW_ATTR_W2 __attribute__((amdgpu_waves_per_eu(2, 2)))  // register budget of a kernel: two waves per SIMD
// The one-instance-per-lane chain units (k_sha, k_g1, k_g2, k_map, k_cofactor, k_prepare) are compiled TWICE (build.py):
//   k_map      the chain programs out of line (BLSW_FN functions): 410-420 registers per kernel, which leaves room for a few waves of the
//              streaming kernels on the same SIMD — what the grouped engine wants beside its HBM-bound expansion / placement
//   k_map_inl  (-DBLSW_KVARIANT_INL) the programs inlined into the kernel: the whole 512-register file, 0.1-0.7 KB of stack instead of
//              1.2-3.9 KB (no argument / callee-saved traffic through scratch): 1.2-3.5x shorter under HBM load (k_map 32.8 -> 9.2 ms,
//              k_g1 22 -> 8.5 ms in blsw_verify_multi_batch) — what the direct-mode entries want (few waves, latency-bound)
//   k_map_q    (-DBLSW_KVARIANT_QUAD) inlined, and every chain on the four lanes of a quad with the independent Fp products of an Fp2 operation
//              on different lanes (fp.hpp, gadgets.hpp): the latency compilation, for small launch groups that start a pipeline
#if defined(BLSW_KVARIANT_QUAD)
#define BLSW_K(name) name##_q
#elif defined(BLSW_KVARIANT_INL)
#define BLSW_K(name) name##_inl
#else
#define BLSW_K(name) name
#endif

#ifndef BLSW_PLACE_ITERS
#define BLSW_PLACE_ITERS 8
#endif
#define BLSW_DIGEST_ITERS 16
#define BLSW_DIGEST_MAX_BLOCKS 4096  // workgroups per instance of k_digest (a single-key vector has 2 073 chunks)
#define BLSW_DIGEST_KEY 0x9E3779B1u  // odd (golden ratio): key_q = (q + 1) * KEY mod 2^32 is a bijection of q
#define BLSW_DIGEST_A 0x85EBCA6Bu

// ---------------------------------------------------------------- kernels (defined in k_*.hip, launched by engine.hip)
__global__ void k_sha(Group g, int want_bits, int write_u);
__global__ void k_sha_inl(Group g, int want_bits, int write_u);
__global__ void k_sha_values(Group g);
__global__ void k_place_field(const Fp* __restrict__ staging, const Fp* __restrict__ pair, uint64_t first, uint32_t off_expand, uint32_t sha_bits,
                              uint32_t staging_rows, uint32_t split_row, uint64_t* __restrict__ d_witness, uint64_t stride, uint32_t n_inst, uint32_t moved_lo,
                              uint32_t moved_len, uint32_t moved_at);
__global__ void k_canonical_rows(uint64_t* __restrict__ d_witness, uint64_t stride, uint32_t off_expand, uint32_t sha_bits, uint32_t rows, uint32_t K, uint32_t stride_hash);
__global__ void k_digest(const uint64_t* __restrict__ w, uint64_t stride, uint64_t n_words, uint64_t* __restrict__ digest);
__global__ void k_g1(Group g);
__global__ void k_g1_inl(Group g);
__global__ void k_agg_keys(Group g, Fp* keyproj);
__global__ void k_agg_keys_inl(Group g, Fp* keyproj);
__global__ void k_agg_sum(Group g, const Fp* keyproj);
__global__ void k_agg_sum_inl(Group g, const Fp* keyproj);
__global__ void k_g2_alloc(Group g);
__global__ void k_g2_alloc_inl(Group g);
__global__ void k_map(Group g);
__global__ void k_map_inl(Group g);
__global__ void k_cofactor(Group g);
__global__ void k_cofactor_inl(Group g);
__global__ void k_cofactor_chunk(Group g);
__global__ void k_cofactor_chunk_inl(Group g);
__global__ void k_cofactor_join(Group g);
__global__ void k_cofactor_join_inl(Group g);
__global__ void k_map_q(Group g);
__global__ void k_prepare_q(Group g, int which);
__global__ void k_g2_alloc_q(Group g);
// values-first prepare chains (prepare_vf.hpp; k_prepare.hip): the serial value phase per point, then one lane per step
__global__ void k_prepv_chain(Group g, int which);
__global__ void k_prepv_chain_q(Group g, int which);
__global__ void k_prepv_step_w(Group g, int which);
// values-first cofactor chain (cofactor_vf.hpp; k_cofv.hip): serial value phases (one lane or one quad per item), parallel witness phases, join
__global__ void k_cofv_chain(Group g, int s);
__global__ void k_cofv_chain_q(Group g, int s);
__global__ void k_cofv_bwd(Group g, int s);
__global__ void k_cofv_bwd_q(Group g, int s);
__global__ void k_cofv_acc(Group g, int s);
__global__ void k_cofv_acc_q(Group g, int s);
__global__ void k_cofv_az(Group g);
__global__ void k_cofv_az_q(Group g);
__global__ void k_cofv_aff(Group g, uint32_t lo, uint32_t cnt);
__global__ void k_cofv_dbl_w(Group g);
__global__ void k_cofv_add_w(Group g);
__global__ void k_cofv_join(Group g);
__global__ void k_map_values(Group g);
__global__ void k_cofactor_values(Group g);
__global__ void k_prepare(Group g, int which);
__global__ void k_prepare_inl(Group g, int which);
__global__ void k_pairing(Group g);
__global__ void k_pairing_team(Group g);
__global__ void k_pairing_team_pv(Group g);
__global__ void k_g2_alloc_team(Group g);
__global__ void k_pairing_team_multi(Group gs, uint32_t K, uint64_t n_h);
__global__ void k_decode(const uint8_t* __restrict__ pk48, const uint8_t* __restrict__ sig96, uint64_t n, uint64_t* pk_xy, uint64_t* sig_xy, int32_t* status);
__global__ void k_h_to_affine(uint64_t n, Workspace ws, uint64_t* d_out);
__global__ void k_decode_points(uint32_t group, const uint8_t* __restrict__ in, uint64_t m, Fp* xy, int32_t* st);
__global__ void k_sum_points(uint32_t group, const Fp* __restrict__ xy, const int32_t* __restrict__ pst, uint32_t k, uint64_t n, uint8_t* out, int32_t* status);
__global__ void k_sign(uint64_t n, Workspace ws, const uint8_t* __restrict__ sk32, uint8_t* sig96, uint64_t* sig_xy, uint8_t* pk48, uint64_t* pk_xy, int32_t* status);
__global__ void k_bench_mad(uint32_t iters, uint32_t* out);
__global__ void k_bench_fill(uint4* __restrict__ dst, uint64_t n16);
__global__ void k_bench_fpmul(uint32_t iters, uint32_t* out);
__global__ void k_bench_fpmul32(uint32_t iters, uint32_t* out);
__global__ void k_bench_fpinv(uint32_t iters, uint32_t* out);
__global__ void k_bench_fp2mulw(uint32_t iters, uint32_t* out);
// pair-parallel Miller product (k_miller_par.hip, miller_par.hpp): value stores of n instances, element-major Fp12 rows
struct MillerParArgs {
    Fp* cprod;   // [12][n * 68 * C] chunk products
    Fp* q;       // [12][n * 68 * C] prefixes over the chunks of a step
    Fp* t;       // [12][n * 68]     product of all pairs of a step
    Fp* f1;      // [12][n * 68]     the running value after ell(sig) of a step
    Fp* ffinal;  // [12][n]          conj(f) after the loop
    uint32_t K, B, C;  // pairs, pairs per chunk, chunks
    uint64_t n_h;      // n * K: lanes of the per-pair launches (where the pairs' line coefficients / prepared keys live)
    uint32_t spine_lane;  // 1: the single-lane spine kernel (k_miller_m2; the statement shared with the host harness) instead of the team kernel
};
inline uint64_t miller_par_bytes(uint64_t n, uint32_t K, uint32_t B) {
    const uint64_t C = (K + B - 1) / B;
    return (2 * n * 68 * C + 2 * n * 68 + n) * 12 * sizeof(Fp) + 5 * 256;
}
void launch_miller_par(const Group& gs, const MillerParArgs& a, hipStream_t st, hipStream_t side, hipEvent_t ev_spine, hipEvent_t ev_side);
// Placement of the N+1-pair product's staged rows (k_stream.hip): up to six runs of consecutive staging rows of a lane, each going to
// dst_off + j * dst_stride of the lane's instance vector (j = pair index of the lane; 0 for instance lanes)
struct PlaceRuns {
    uint32_t n_runs;
    uint32_t src_row[7];  // first staging row of run r; src_row[n_runs] = rows of the tile
    uint32_t dst_off[6], dst_stride[6];
};
__global__ void k_place_runs(const Fp* __restrict__ tiles, uint64_t first, uint32_t rows, PlaceRuns runs, uint64_t* __restrict__ d_witness, uint64_t stride, uint32_t n_y,
                             uint32_t K, uint32_t tile_w);
__global__ void k_place_rows(const Fp* __restrict__ rows, uint32_t n_rows, uint32_t dst_off, uint64_t* __restrict__ d_witness, uint64_t stride);
// the two compilations of the chain units as one table
struct ChainKernels {
    void (*sha)(Group, int, int);
    void (*g1)(Group);
    void (*agg_keys)(Group, Fp*);
    void (*agg_sum)(Group, const Fp*);
    void (*g2_alloc)(Group);
    void (*map)(Group);
    void (*cofactor)(Group);
    void (*prepare)(Group, int);
    void (*cofactor_chunk)(Group);  // the cofactor segment with its three chunks on three lanes, and the join (cofactor_par.hpp)
    void (*cofactor_join)(Group);
};
inline ChainKernels chain_kernels(bool inlined) {
    if (inlined) return {k_sha_inl, k_g1_inl, k_agg_keys_inl, k_agg_sum_inl, k_g2_alloc_inl, k_map_inl, k_cofactor_inl, k_prepare_inl, k_cofactor_chunk_inl, k_cofactor_join_inl};
    return {k_sha, k_g1, k_agg_keys, k_agg_sum, k_g2_alloc, k_map, k_cofactor, k_prepare, k_cofactor_chunk, k_cofactor_join};
}
// Latency forms of a launch group's chains (small groups: Workspace::cofv exists): `quad` = the *_q compilation of map / prepare / G2 allocation and
// of the serial phases of the values-first cofactor chain (four lanes per item); `vf` = clear_cofactor2 values first (cofactor_vf.hpp)
struct Latency {
    bool quad, vf;
};
// clear_cofactor2 of N lanes on `st`: values first, chunked (three lanes per (pk, msg) pair + the join) or as one chain per lane.
// Values first with side streams (CofactorSide): `st` carries the forward doubling chain in its segments and the last segment's tail. Beside it, as soon
// as a segment's doublings are done (ev_seg): its inversion / affine points on pts, in segment order (ev_pts), and its part of its chunk's addition chain
// on acc[0] (chunks 0 and 2) or acc[1] (chunk 1); the join waits for the chunks (ev_acc). The witness phases (one lane per doubling / addition, and the
// chunks' 1 / Z1 sweeps: nothing but the segment's placement waits for them) are left to launch_cofactor_witness on `side`, after ev_join.
// side == nullptr: everything on `st`.
struct CofactorSide {
    hipStream_t side, pts, acc[2];
    hipEvent_t ev_seg[BLSW_COFV_NSEG], ev_pts[BLSW_COFV_NSEG], ev_acc[3], ev_join;
};
#define BLSW_COFV_EVENTS (2 * BLSW_COFV_NSEG + 4)
inline uint64_t cofv_total_adds() {
    constexpr CofvPlan plan = cofv_plan();
    return (uint64_t)plan.n_adds[0] + plan.n_adds[1] + plan.n_adds[2];
}
inline void launch_cofv_chain(Latency lat, const Group& g, int s, hipStream_t q) {
    if (lat.quad)
        hipLaunchKernelGGL(k_cofv_chain_q, dim3(item_grid(g.N, 4)), dim3(64), 0, q, g, s);
    else
        hipLaunchKernelGGL(k_cofv_chain, dim3(item_grid(g.N, 1)), dim3(64), 0, q, g, s);
}
// phases 1b and 2a of segment s: 1 / Z of its points, then the points in affine form
inline void launch_cofv_points(Latency lat, const Group& g, int s, hipStream_t q) {
    constexpr CofvSeg seg = cofv_seg();
    if (lat.quad)
        hipLaunchKernelGGL(k_cofv_bwd_q, dim3(item_grid(g.N, 4)), dim3(64), 0, q, g, s);
    else
        hipLaunchKernelGGL(k_cofv_bwd, dim3(item_grid(g.N, 1)), dim3(64), 0, q, g, s);
    const uint32_t lo = s == 0 ? 0 : seg.bnd[s] + 1u;
    const uint32_t hi = seg.bnd[s + 1] < BLSW_H_EFF_NBITS ? seg.bnd[s + 1] : BLSW_H_EFF_NBITS - 1;  // the last point of the chain is not an operand
    hipLaunchKernelGGL(k_cofv_aff, dim3(item_grid((uint64_t)(hi - lo + 1) * g.N, 1)), dim3(64), 0, q, g, lo, hi - lo + 1);
}
inline void launch_cofv_acc(Latency lat, const Group& g, int s, hipStream_t q) {
    if (lat.quad)
        hipLaunchKernelGGL(k_cofv_acc_q, dim3(item_grid(g.N, 4)), dim3(64), 0, q, g, s);
    else
        hipLaunchKernelGGL(k_cofv_acc, dim3(item_grid(g.N, 1)), dim3(64), 0, q, g, s);
}
inline void launch_cofv_witness_phases(Latency lat, const Group& g, hipStream_t q) {
    hipLaunchKernelGGL(k_cofv_dbl_w, dim3(item_grid((uint64_t)BLSW_H_EFF_NBITS * g.N, 1)), dim3(64), 0, q, g);
    if (lat.quad)
        hipLaunchKernelGGL(k_cofv_az_q, dim3(item_grid(3 * g.N, 4)), dim3(64), 0, q, g);
    else
        hipLaunchKernelGGL(k_cofv_az, dim3(item_grid(3 * g.N, 1)), dim3(64), 0, q, g);
    hipLaunchKernelGGL(k_cofv_add_w, dim3(item_grid(cofv_total_adds() * g.N, 1)), dim3(64), 0, q, g);
}
inline void launch_cofactor(const ChainKernels& ck, Latency lat, bool chunked, const Group& g, hipStream_t st, const CofactorSide* side = nullptr) {
    const unsigned g1 = item_grid(g.N, 1), g3 = item_grid(3 * g.N, 1);
    if (lat.vf && g.ws.cofv) {
        constexpr CofvSeg seg = cofv_seg();
        constexpr int last = BLSW_COFV_NSEG - 1;
        if (!side) {
            for (int s = 0; s <= last; s++) {
                launch_cofv_chain(lat, g, s, st);
                launch_cofv_points(lat, g, s, st);
                launch_cofv_acc(lat, g, s, st);
            }
            hipLaunchKernelGGL(k_cofv_join, dim3(g1), dim3(64), 0, st, g);
            launch_cofv_witness_phases(lat, g, st);
            return;
        }
        for (int s = 0; s <= last; s++) {
            launch_cofv_chain(lat, g, s, st);
            hipEventRecord(side->ev_seg[s], st);
        }
        for (int s = 0; s < last; s++) {
            hipStreamWaitEvent(side->pts, side->ev_seg[s], 0);
            launch_cofv_points(lat, g, s, side->pts);
            hipEventRecord(side->ev_pts[s], side->pts);
            hipStream_t q = side->acc[seg.chunk[s] == 1 ? 1 : 0];
            hipStreamWaitEvent(q, side->ev_pts[s], 0);
            launch_cofv_acc(lat, g, s, q);
            if (seg.last(s) || s == last - 1) hipEventRecord(side->ev_acc[seg.chunk[s]], q);
        }
        launch_cofv_points(lat, g, last, st);
        hipStreamWaitEvent(st, side->ev_acc[2], 0);  // the chunk's additions before the last segment's
        launch_cofv_acc(lat, g, last, st);
        hipStreamWaitEvent(st, side->ev_acc[0], 0);
        hipStreamWaitEvent(st, side->ev_acc[1], 0);
        hipLaunchKernelGGL(k_cofv_join, dim3(g1), dim3(64), 0, st, g);
        hipEventRecord(side->ev_join, st);
    } else if (chunked) {
        hipLaunchKernelGGL(ck.cofactor_chunk, dim3(g3), dim3(64), 0, st, g);
        hipLaunchKernelGGL(ck.cofactor_join, dim3(g1), dim3(64), 0, st, g);
    } else {
        hipLaunchKernelGGL(ck.cofactor, dim3(g1), dim3(64), 0, st, g);
    }
}
// the deferred witness phases of a values-first cofactor chain (launch_cofactor with side streams), enqueued on the side stream
inline void launch_cofactor_witness(Latency lat, const Group& g, const CofactorSide& side) {
    hipStreamWaitEvent(side.side, side.ev_join, 0);
    launch_cofv_witness_phases(lat, g, side.side);
}
inline void launch_map(const ChainKernels& ck, Latency lat, const Group& g, hipStream_t st) {
    if (lat.quad)
        hipLaunchKernelGGL(k_map_q, dim3(item_grid(2 * g.N, 4)), dim3(64), 0, st, g);
    else
        hipLaunchKernelGGL(ck.map, dim3(item_grid(2 * g.N, 1)), dim3(64), 0, st, g);
}
inline void launch_prepare(const ChainKernels& ck, Latency lat, const Group& g, int which, hipStream_t st) {
    if (lat.vf && g.ws.prepv_h) {
        if (lat.quad)
            hipLaunchKernelGGL(k_prepv_chain_q, dim3(item_grid(g.N, 4)), dim3(64), 0, st, g, which);
        else
            hipLaunchKernelGGL(k_prepv_chain, dim3(item_grid(g.N, 1)), dim3(64), 0, st, g, which);
        hipLaunchKernelGGL(k_prepv_step_w, dim3(item_grid((uint64_t)BLSW_PREPV_STEPS * g.N, 1)), dim3(64), 0, st, g, which);
    } else if (lat.quad)
        hipLaunchKernelGGL(k_prepare_q, dim3(item_grid(g.N, 4)), dim3(64), 0, st, g, which);
    else
        hipLaunchKernelGGL(ck.prepare, dim3(item_grid(g.N, 1)), dim3(64), 0, st, g, which);
}
inline void launch_g2_alloc(const ChainKernels& ck, Latency lat, const Group& g, hipStream_t st) {
    if (lat.quad)
        hipLaunchKernelGGL(k_g2_alloc_q, dim3(item_grid(g.N, 4)), dim3(64), 0, st, g);
    else
        hipLaunchKernelGGL(ck.g2_alloc, dim3(item_grid(g.N, 1)), dim3(64), 0, st, g);
}
// host-side launch helpers that live next to their (templated) kernels
void launch_expand(uint32_t variant, uint32_t store, unsigned lds, hipStream_t st, ExpandArgs a, unsigned n_y);
void launch_pairing(const Group& g, const Modes& m, hipStream_t st);
#define BLSW_VLINE_ROWS (6u * 68u)  // vpairing.hpp: per G2 point, 68 steps x (c0, c1 x_P, c2 y_P), two Fp each
// blsw_verify_batch: projective lines of (sig, H(m)) then the six-lane value-only pairing check (k_team.hip, vpairing.hpp)
void launch_verify_values(uint64_t n, const Workspace& ws, const uint64_t* pk_xy, const uint64_t* sig_xy, Fp* lines_sig, Fp* lines_h, const int32_t* status, int32_t* result,
                          hipStream_t st);

}  // namespace blsw
