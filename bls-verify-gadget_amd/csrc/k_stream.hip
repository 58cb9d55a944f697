// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
#include "kcommon.hpp"

namespace blsw {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ void expand_store(uint4* dst, const uint4& v) {
    if (NT == 2) {
        u32x4 vv = {v.x, v.y, v.z, v.w};
        // (two wait states after a store of more than 64 bits before a vector instruction may overwrite its data: the compiler does not look into asm)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
    } else if (NT == 3) {
        u32x4 vv = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
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
// loop invariant; a workgroup writes THREADS * ITERS consecutive pieces, THREADS of them per iteration. (bx, by) = the block's place in the
// (blocks per instance, instances) grid: blockIdx of variant 0, a loop variable of the persistent variant 13.
template <int THREADS, int ITERS, int ALIGN_PIECES, int NT>
__device__ __forceinline__ void expand_block(const ExpandArgs& a, uint32_t bx, uint32_t by) {
    const uint64_t inst = a.K == 1 ? by : by / a.K;
    const uint32_t pair = a.K == 1 ? 0u : by - (uint32_t)inst * a.K;
    uint4* out = reinterpret_cast<uint4*>(a.d_witness + (inst * a.stride + a.off_expand + (uint64_t)pair * a.stride_hash) * 6);
    const uint64_t lane = a.first + by;
    const uint32_t* b = a.bits + (lane >> 6) * bits_tile_words(a.sha_words) + (lane & 63) * BLSW_BITS_CHUNK_WORDS;
    const uint32_t n_pieces = a.sha_bits * 3;
    const uint32_t P0 = (ALIGN_PIECES - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) % ALIGN_PIECES)) % ALIGN_PIECES;
    if (bx == 0 && threadIdx.x < P0 && threadIdx.x < n_pieces) {  // the pieces in front of the first boundary
        const uint32_t e = threadIdx.x / 3, c = threadIdx.x - 3 * e;
        const uint32_t m = 0u - ((expand_word(b, e >> 5) >> (e & 31)) & 1u);
        const uint4 rc = expand_column(c, a.canonical);
        expand_store<NT>(&out[threadIdx.x], make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m));
    }
    const uint32_t pt = P0 + threadIdx.x, c = pt % 3;
    const uint32_t e0 = bx * ((THREADS / 3) * ITERS) + pt / 3;
    const uint4 rc = expand_column(c, a.canonical);
    // THREADS / 3 is a multiple of 32: the bit position of a thread is a loop invariant too, its word advances by THREADS / 96
    static_assert((THREADS / 3) % 32 == 0, "bit position must be loop invariant");
    const uint32_t sh = e0 & 31, w0 = e0 >> 5;
    uint4* dst = out + (uint64_t)e0 * 3 + c;
    if (bx * ((THREADS / 3) * ITERS) + (P0 + THREADS - 1) / 3 + (THREADS / 3) * (ITERS - 1) < a.sha_bits) {
        // whole workgroup in range (all but the last one or two of an instance): all bit words first, then the stores back to
        // back — no bounds checks, no wait between a store and the next load
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
template <int THREADS, int ITERS, int ALIGN_PIECES, int NT>
__global__ __launch_bounds__(THREADS) void k_sha_expand(ExpandArgs a) {
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    expand_block<THREADS, ITERS, ALIGN_PIECES, NT>(a, blockIdx.x, blockIdx.y);
}
// variant 13: the blocks of variant 0 walked by a RESIDENT grid (BLSW_EXPAND_RESIDENT_WGS workgroups, block = blockIdx.x + k * gridDim.x). A grid of
// 246 144 workgroups keeps the dispatcher of its hardware queue's pipe busy for the whole run of the kernel, and kernels of other queues on that pipe
// are not dispatched meanwhile (a chain kernel launched beside an expansion waited for the expansion's END: engine.hip, materialise); this grid is
// dispatched in microseconds. Two workgroups (twelve 24-register waves) per compute unit leave every SIMD five wave slots and 440 registers.
template <int THREADS, int ITERS, int ALIGN_PIECES, int NT>
__global__ __launch_bounds__(THREADS) void k_sha_expand_resident(ExpandArgs a, uint32_t blocks_x, uint32_t n_y) {
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t total = (uint64_t)blocks_x * n_y;
#pragma unroll 1
    for (uint64_t blk = blockIdx.x; blk < total; blk += gridDim.x) {
        const uint32_t by = (uint32_t)(blk / blocks_x);
        expand_block<THREADS, ITERS, ALIGN_PIECES, NT>(a, (uint32_t)(blk - (uint64_t)by * blocks_x), by);
    }
}
// variants 1 / 6 / 7: one piece per thread, workgroup b >= 1 writes exactly one chunk of THREADS pieces aligned to its own size
// (256 threads: 4 KiB — alone the fastest geometry on this chip, 6.7 TB/s; 512 / 1024 threads: 8 / 16 KiB chunks, 2 / 4 x fewer workgroups)
template <int THREADS, int NT>
__global__ __launch_bounds__(THREADS) void k_sha_expand_chunk(ExpandArgs a) {
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    uint4* out;
    const uint32_t* b;
    expand_locate(a, out, b);
    const uint32_t n_pieces = a.sha_bits * 3;
    const uint32_t P0 = (THREADS - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) % THREADS)) % THREADS;
    expand_head<NT>(out, b, P0, n_pieces, a.canonical);
    const uint32_t p = P0 + blockIdx.x * THREADS + threadIdx.x;
    if (p >= n_pieces) return;
    const uint32_t e = p / 3, c = p - 3 * e;
    const uint32_t m = 0u - ((expand_word(b, e >> 5) >> (e & 31)) & 1u);
    const uint4 rc = expand_column(c, a.canonical);
    expand_store<NT>(&out[p], make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m));
}
// variant 8: the geometry of variant 0 (384 threads x ITERS pieces 6 KiB apart, pieces counted from the first 256-byte boundary) with the
// bit words in SCALAR registers: the 64 lanes of a wave cover 21.3 consecutive elements, i.e. one or two bit words per iteration, which
// the wave fetches with scalar loads (wave-uniform addresses, constant address space) — 12 vector registers per lane instead of 21, so
// that five of these waves (instead of three) fit beside a 417-register chain wave on the same SIMD
typedef const uint32_t __attribute__((address_space(4))) * blsw_cptr;
template <int ITERS>
__global__ __launch_bounds__(384) void k_sha_expand_s(ExpandArgs a) {
    constexpr int THREADS = 384;
    uint4* out;
    const uint32_t* b;
    expand_locate(a, out, b);
    const uint32_t n_pieces = a.sha_bits * 3;
    const uint32_t P0 = (16 - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) % 16)) % 16;
    expand_head<0>(out, b, P0, n_pieces, a.canonical);
    const uint32_t pt = P0 + threadIdx.x, c = pt % 3;
    const uint32_t e0 = blockIdx.x * ((THREADS / 3) * ITERS) + pt / 3;
    const uint4 rc = expand_column(c, a.canonical);
    const uint32_t sh = e0 & 31, w0 = e0 >> 5;
    uint4* dst = out + (uint64_t)e0 * 3 + c;
    // one path for every workgroup (the last one or two of an instance check each piece; their word indices are clamped into the
    // instance's padded stream: a piece beyond the segment is never stored)
    const uint32_t wu = __builtin_amdgcn_readfirstlane(w0);  // the wave's first word; a lane needs word wu or wu + 1
    const bool second = w0 != wu;
    const uint32_t w_last = (uint32_t)a.sha_words - 1;
    blsw_cptr bc = (blsw_cptr)(uintptr_t)b;
    uint32_t wa[ITERS], wb[ITERS];
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
        uint32_t w = wu + k * (THREADS / 96), w1 = w + 1;
        w = w < w_last ? w : w_last;
        w1 = w1 < w_last ? w1 : w_last;
        wa[k] = bc[(uint64_t)(w / BLSW_BITS_CHUNK_WORDS) * (64 * BLSW_BITS_CHUNK_WORDS) + (w % BLSW_BITS_CHUNK_WORDS)];
        wb[k] = bc[(uint64_t)(w1 / BLSW_BITS_CHUNK_WORDS) * (64 * BLSW_BITS_CHUNK_WORDS) + (w1 % BLSW_BITS_CHUNK_WORDS)];
    }
    const bool whole = blockIdx.x * ((THREADS / 3) * ITERS) + (P0 + THREADS - 1) / 3 + (THREADS / 3) * (ITERS - 1) < a.sha_bits;
#pragma unroll
    for (int k = 0; k < ITERS; k++) {
        const uint32_t m = 0u - (((second ? wb[k] : wa[k]) >> sh) & 1u);
        if (whole || e0 + (THREADS / 3) * k < a.sha_bits) expand_store<0>(dst + (uint64_t)k * THREADS, make_uint4(rc.x & m, rc.y & m, rc.z & m, rc.w & m));
    }
}
// variant 10: the geometry of variant 0 with a LIGHT instruction stream — 5 vector instructions per 16-byte store instead of ~20 (profiles/r04_digest.txt:
// 6.2 x 10^8 VALU wave-instructions per launch, a third of what a step's chain kernels issue). Piece index of a thread = workgroup base + pt + 384 k, so
//   * the store address is a scalar base (advanced per iteration by the scalar unit) + a 32-bit lane offset pt * 16: global_store ... saddr, no vector add;
//   * a wave covers 22 consecutive elements: the scalar unit loads the one or two bit words, shifts the pair so that the wave's first element is bit 0
//     (the shift is a loop invariant: the element advances by 128 per iteration), and a lane extracts its bit with ONE v_bfe_i32 at a fixed index.
// store to (scalar base + 32-bit lane offset), saddr form. Inline assembly, because instruction selection folds the eight bases of a thread into vector
// adds + immediate offsets otherwise; the five wait states gfx9 wants between a scalar-unit write of an SGPR and a vector-memory instruction that reads it
// are part of the statement (the compiler's hazard recogniser does not look into asm: without them the store used a stale base whenever the scheduler
// had put the scalar add right in front of it — caught by test_expansion_geometries_bit_exact), and so are the two after it that a
// store of more than 64 bits wants before a vector instruction may overwrite its data registers
// The wait states in the inline assembly of this file (s_nop 4 between a scalar write of a store's base and the store, s_nop 1 after a 128-bit
// store whose data registers are overwritten next) are counted by hand for ONE target. Another target must not silently inherit them.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_stream.hip: the hand-counted hazard wait states of expand_store / expand_store_s are valid for gfx950 only"
#endif
__device__ __forceinline__ void expand_store_s(uint64_t sbase, uint32_t voff, const u32x4& v) {
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
template <int ITERS>
__global__ __launch_bounds__(384) void k_sha_expand_l(ExpandArgs a) {
    constexpr int THREADS = 384;
    uint4* out;
    const uint32_t* b;
    expand_locate(a, out, b);
    const uint32_t n_pieces = a.sha_bits * 3;
    const uint32_t P0 = (16 - (uint32_t)((reinterpret_cast<uintptr_t>(out) >> 4) % 16)) % 16;
    expand_head<0>(out, b, P0, n_pieces, a.canonical);
    const uint32_t pt = P0 + threadIdx.x, c = pt % 3;
    const uint32_t e_blk = blockIdx.x * ((THREADS / 3) * ITERS);
    const uint32_t e0 = e_blk + pt / 3;
    const uint4 rc = expand_column(c, a.canonical);
    const uint32_t e_first = __builtin_amdgcn_readfirstlane(e0);  // the wave's first element
    const uint32_t idx = e0 - e_first;                            // 0 .. 21: the lane's bit in the wave's window
    const uint32_t sh = e_first & 31, wu = e_first >> 5, w_last = (uint32_t)a.sha_words - 1;
    blsw_cptr bc = (blsw_cptr)(uintptr_t)b;
    auto window = [&](int k) {  // bits of the wave's 22 elements of iteration k, the first one in bit 0 (scalar loads, scalar shift)
        uint32_t w = wu + k * (THREADS / 96), w1 = w + 1;
        w = w < w_last ? w : w_last;  // (the last workgroups of an instance reach beyond its stream: clamped, those pieces are never stored)
        w1 = w1 < w_last ? w1 : w_last;
        const uint32_t lo = bc[(uint64_t)(w / BLSW_BITS_CHUNK_WORDS) * (64 * BLSW_BITS_CHUNK_WORDS) + (w % BLSW_BITS_CHUNK_WORDS)];
        const uint32_t hi = bc[(uint64_t)(w1 / BLSW_BITS_CHUNK_WORDS) * (64 * BLSW_BITS_CHUNK_WORDS) + (w1 % BLSW_BITS_CHUNK_WORDS)];
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
    };
    const bool whole = e_blk + (P0 + THREADS - 1) / 3 + (THREADS / 3) * (ITERS - 1) < a.sha_bits;
    const uint64_t sbase = reinterpret_cast<uint64_t>(out) + (uint64_t)e_blk * 48;  // scalar: first byte of the workgroup's pieces
    const uint32_t voff = pt * 16;
    if (whole) {  // all but the last one or two workgroups of an instance: every window first (one scalar-load latency per wave), then the stores back to back
        uint32_t win[ITERS];  // scalar registers
#pragma unroll
        for (int k = 0; k < ITERS; k++) win[k] = window(k);
#pragma unroll
        for (int k = 0; k < ITERS; k++) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe(win[k], idx, 1);
            const u32x4 v = {rc.x & m, rc.y & m, rc.z & m, rc.w & m};
            expand_store_s(sbase + (uint64_t)k * (THREADS * 16), voff, v);
        }
        return;
    }
#pragma unroll 1
    for (int k = 0; k < ITERS; k++) {
        const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe(window(k), idx, 1);
        const u32x4 v = {rc.x & m, rc.y & m, rc.z & m, rc.w & m};
        if (e0 + (THREADS / 3) * k < a.sha_bits) expand_store_s(sbase + (uint64_t)k * (THREADS * 16), voff, v);
    }
}
// variant: low byte 0..12 = geometry, bit 8 = raised wave priority; store: 0 plain, 1 nontemporal, 2 sc1, 3 sc0 sc1 (variant 0 only)
void launch_expand(uint32_t variant, uint32_t store, unsigned lds, hipStream_t st, ExpandArgs a, unsigned n_y) {
    a.prio = (variant >> 8) & 1;
    const uint32_t n_pieces = a.sha_bits * 3;
    auto grid = [&](uint32_t per_wg) { return dim3((n_pieces + per_wg - 1) / per_wg + 1, n_y); };
    switch (variant & 0xff) {
        case 1: hipLaunchKernelGGL((k_sha_expand_chunk<256, 0>), grid(256), dim3(256), lds, st, a); break;
        case 6: hipLaunchKernelGGL((k_sha_expand_chunk<512, 0>), grid(512), dim3(512), lds, st, a); break;
        case 7: hipLaunchKernelGGL((k_sha_expand_chunk<1024, 0>), grid(1024), dim3(1024), lds, st, a); break;
        case 8: hipLaunchKernelGGL(k_sha_expand_s<8>, grid(384 * 8), dim3(384), lds, st, a); break;
        case 9: hipLaunchKernelGGL(k_sha_expand_s<4>, grid(384 * 4), dim3(384), lds, st, a); break;
        case 10: hipLaunchKernelGGL(k_sha_expand_l<8>, grid(384 * 8), dim3(384), lds, st, a); break;
        case 11: hipLaunchKernelGGL(k_sha_expand_l<16>, grid(384 * 16), dim3(384), lds, st, a); break;
        case 12: hipLaunchKernelGGL(k_sha_expand_l<32>, grid(384 * 32), dim3(384), lds, st, a); break;
        case 13: hipLaunchKernelGGL((k_sha_expand_resident<384, 8, 16, 0>), dim3(BLSW_EXPAND_RESIDENT_WGS), dim3(384), lds, st, a, grid(384 * 8).x, n_y); break;
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

// N+1-pair product in the grouped engine: the staged rows of a step go to their places in the instance vectors.
// k_place_runs: lanes first .. first + n_y of tile-major staging (pair tiles with K pairs per instance, or instance tiles with K = 1);
// a lane's rows are up to six runs of consecutive rows (PlaceRuns). Same gather shape and XCD-aware block order as k_place_field.
__global__ __launch_bounds__(256) void k_place_runs(const Fp* __restrict__ tiles, uint64_t first, uint32_t rows, PlaceRuns runs, uint64_t* __restrict__ d_witness,
                                                    uint64_t stride, uint32_t n_y, uint32_t K, uint32_t tile_w) {
    const uint32_t L = blockIdx.x, s_in_xcd = L >> 3;
    const uint32_t chunk = (L & 7) + 8 * (s_in_xcd / n_y);
    const uint32_t y = s_in_xcd % n_y;
    const uint32_t npieces = rows * 3;
    if (chunk * (256u * BLSW_PLACE_ITERS) >= npieces) return;
    const uint64_t lane = first + y;
    const uint32_t inst = y / K, j = y - inst * K;
    // tile_w lanes per tile: 64 in the workspace; a compact buffer packs the instance rows of a step that is a part of one tile
    const uint4* src = reinterpret_cast<const uint4*>(tiles + (lane / tile_w) * (uint64_t)rows * tile_w + (lane % tile_w));
    uint4* out = reinterpret_cast<uint4*>(d_witness + (uint64_t)inst * stride * 6);
    const uint32_t q0 = chunk * (256 * BLSW_PLACE_ITERS) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < BLSW_PLACE_ITERS; k++) {
        const uint32_t q = q0 + k * 256;
        if (q < npieces) {
            const uint32_t e = q / 3, c = q - e * 3;
            uint32_t r = 0;
#pragma unroll
            for (int t = 1; t < 6; t++) r += (t < (int)runs.n_runs && e >= runs.src_row[t]) ? 1u : 0u;
            const uint32_t dst_e = runs.dst_off[r] + j * runs.dst_stride[r] + (e - runs.src_row[r]);
            out[(uint64_t)dst_e * 3 + c] = src[(uint64_t)e * tile_w * 3 + c];
        }
    }
}
// k_place_rows: instance-major rows (Miller loop, final exponentiation, is_one) -> one contiguous run of each vector. grid (chunks, n)
__global__ __launch_bounds__(256) void k_place_rows(const Fp* __restrict__ rows, uint32_t n_rows, uint32_t dst_off, uint64_t* __restrict__ d_witness, uint64_t stride) {
    const uint64_t inst = blockIdx.y;
    const uint4* src = reinterpret_cast<const uint4*>(rows + inst * n_rows);
    uint4* out = reinterpret_cast<uint4*>(d_witness + (inst * stride + dst_off) * 6);
    const uint32_t npieces = n_rows * 3;
    uint32_t q = blockIdx.x * (256 * BLSW_PLACE_ITERS) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < BLSW_PLACE_ITERS; k++, q += 256)
        if (q < npieces) out[q] = src[q];
}

// Digest of witness vectors (blsw_witness_digest; the definition is in include/blsw.h). With x_j the little-endian u32 words of an
// instance's vector, 16-byte piece q = (x_4q .. x_4q+3) and key_q = (q + 1) * BLSW_DIGEST_KEY mod 2^32:
//   d[0] = sum_q (x_4q + key_q) (x_4q+1 + key_q + A) + (x_4q+2 + key_q + 2 A) (x_4q+3 + key_q + 3 A)   mod 2^64
//          (32-bit sums, 32 x 32 -> 64-bit products: the NH family of UMAC with a position-derived key; ONE multiply per 8 bytes)
//   d[1] = lo | hi << 32:  lo = sum_q (x_4q ^ key_q) + (x_4q+2 ^ ~key_q),  hi = sum_q (x_4q+1 ^ key_q) + (x_4q+3 ^ ~key_q)   mod 2^32
// 12 vector instructions per 16 bytes (round 3: four splitmix finalizers = 90, VALU-bound at 3.9 TB/s). grid (chunks, n); 256 threads,
// BLSW_DIGEST_ITERS pieces per thread 4 KiB apart, the loads of four pieces in flight per thread; one atomic triple per workgroup.
__global__ __launch_bounds__(256) void k_digest(const uint64_t* __restrict__ w, uint64_t stride, uint64_t n_words, uint64_t* __restrict__ digest) {
    const uint64_t inst = blockIdx.y;
    const u32x4* src = reinterpret_cast<const u32x4*>(w + inst * stride * 6);
    const uint64_t n_pieces = n_words / 2;  // n_witness * 6 is even; 64-bit: a 4 096-pair vector has 8.4 x 10^9 pieces (the key is (q + 1) * KEY mod 2^32)
    uint32_t key = 0;
    uint64_t d0 = 0;
    uint32_t lo = 0, hi = 0;
    auto piece = [&](const u32x4& v) {
        d0 += (uint64_t)(v.x + key) * (uint64_t)(v.y + key + BLSW_DIGEST_A);
        d0 += (uint64_t)(v.z + key + 2 * BLSW_DIGEST_A) * (uint64_t)(v.w + key + 3 * BLSW_DIGEST_A);
        lo += (v.x ^ key) + (v.z ^ ~key);
        hi += (v.y ^ key) + (v.w ^ ~key);
        key += 256 * BLSW_DIGEST_KEY;
    };
    // a workgroup walks chunks blockIdx.x, blockIdx.x + gridDim.x, ... of its instance (the host caps gridDim.x: a 4.19 GB vector of the
    // N+1-pair circuit would otherwise end in 64 k atomic triples on the same three addresses)
    const uint32_t n_chunks = (uint32_t)((n_pieces + BLSW_DIGEST_ITERS * 256 - 1) / (BLSW_DIGEST_ITERS * 256));
#pragma unroll 1
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t q0 = (uint64_t)chunk * (BLSW_DIGEST_ITERS * 256) + threadIdx.x;
        key = ((uint32_t)q0 + 1) * BLSW_DIGEST_KEY;
        if ((uint64_t)chunk * (BLSW_DIGEST_ITERS * 256) + BLSW_DIGEST_ITERS * 256 <= n_pieces) {  // whole chunk in range
#pragma unroll 1
            for (int it = 0; it < BLSW_DIGEST_ITERS; it += 4) {
                u32x4 v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = __builtin_nontemporal_load(&src[q0 + (it + k) * 256]);
#pragma unroll
                for (int k = 0; k < 4; k++) piece(v[k]);
            }
        } else {
            uint64_t q = q0;
#pragma unroll 1
            for (int it = 0; it < BLSW_DIGEST_ITERS; it++, q += 256) {
                if (q < n_pieces) piece(src[q]);
            }
        }
    }
    // wave reduction, workgroup reduction through LDS, then one atomic triple per workgroup (the halves of d[1] are sums mod 2^32)
    for (int off = 32; off > 0; off >>= 1) {
        d0 += __shfl_down(d0, off, 64);
        lo += __shfl_down(lo, off, 64);
        hi += __shfl_down(hi, off, 64);
    }
    __shared__ uint64_t s_d0[4];
    __shared__ uint32_t s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) {
        s_d0[threadIdx.x >> 6] = d0;
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(digest + inst * 2), (unsigned long long)(s_d0[0] + s_d0[1] + s_d0[2] + s_d0[3]));
        uint32_t* d1 = reinterpret_cast<uint32_t*>(digest + inst * 2 + 1);
        atomicAdd(d1, s_lo[0] + s_lo[1] + s_lo[2] + s_lo[3]);
        atomicAdd(d1 + 1, s_hi[0] + s_hi[1] + s_hi[2] + s_hi[3]);
    }
}

// options.output_form = 1: the field witnesses of a step, in place, from Montgomery form to canonical integers (what
// CanonicalSerialize writes for an Fq: 48 bytes little-endian); the SHA segment is written in that form by the expansion itself
// K SHA segments (one per pair, stride_hash apart; K = 1: the single-key and aggregate circuits) are skipped: row idx of the field rows
// lies in front of the first segment, in the tail of pair j's hash block (hash_tail rows behind each segment), or behind the last block
__global__ __launch_bounds__(256) void k_canonical_rows(uint64_t* __restrict__ d_witness, uint64_t stride, uint32_t off_expand, uint32_t sha_bits, uint32_t rows, uint32_t K,
                                                        uint32_t stride_hash) {
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows) return;
    uint32_t el;
    if (idx < off_expand)
        el = idx;
    else if (K <= 1)
        el = idx + sha_bits;
    else {
        const uint32_t hash_tail = stride_hash - sha_bits, r = idx - off_expand;
        const uint32_t j = r / hash_tail;
        el = j < K ? off_expand + j * stride_hash + sha_bits + (r - j * hash_tail) : idx + K * sha_bits;
    }
    Fp* p = reinterpret_cast<Fp*>(d_witness + ((uint64_t)blockIdx.y * stride + el) * 6);
    st_fp(p, fp_to_canonical(ld_fp(p)));
}

}  // namespace blsw
