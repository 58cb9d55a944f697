// In-circuit expand_message_xmd / hash_to_field of the reference (src/hasher.rs:58-173), one message per lane.
// The SHA-256 gadget (ark-crypto-primitives ^0.4.0 crh/sha256/constraints.rs) allocates one
// boolean witness per AND / XOR on variables and 33/34/35 booleans per UInt32::addmany; bits that are constants
// (padding, DST, the all-constant Z_pad block, initial state) fold away. Which positions are constant is a
// property of the circuit shape, identical for every lane; it is tracked in uniform masks next to the per-lane
// values, so the kernel emits exactly the bits arkworks would allocate, in allocation order.
// Output of this stage is a BITSTREAM (1 bit per boolean witness); sha_expand turns it into 48-byte Fp elements.
#pragma once
#include "fp.hpp"

namespace blsw {

struct W32 {
    uint32_t v;   // boolean values of the 32 bits (negation applied)
    uint32_t cm;  // 1 = bit is Boolean::Constant
    uint32_t nm;  // 1 = bit is Boolean::Not(var)   (0 where constant)
};
BLSW_HD W32 w_const(uint32_t v) { return {v, 0xffffffffu, 0u}; }
BLSW_HD uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
BLSW_HD W32 w_rotr(const W32& a, int n) { return {rotr32(a.v, n), rotr32(a.cm, n), rotr32(a.nm, n)}; }
BLSW_HD W32 w_shr(const W32& a, int n) { return {a.v >> n, (a.cm >> n) | ~(0xffffffffu >> n), a.nm >> n}; }
BLSW_HD W32 w_not(const W32& a) { return {~a.v, a.cm, ~a.nm & ~a.cm}; }

BLSW_FN uint32_t pext32(uint32_t v, uint32_t m) {
    if (m == 0xffffffffu) return v;
    if ((m & (m + 1)) == 0) return v & m;  // contiguous low mask
    uint32_t r = 0, k = 0;
    while (m) {
        uint32_t low = m & (0u - m);
        r |= ((v & low) ? 1u : 0u) << k;
        k++;
        m ^= low;
    }
    return r;
}
BLSW_HD int popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}

// Bit sink: packs the appended bits into 32-bit words.
//  host  : words stored at out[word * stride] (nullptr = count only: the layout's bit count)
//  device: the wave's 64 lanes are 64 instances of one tile. Words are collected in LDS ([16][64] u32, 4 KiB per wave, written
//          and read back by the same lane) and leave as 64-BYTE RUNS: chunk c of lane l lives at tile + (c * 64 + l) * 64 bytes,
//          so that the expansion kernel, which walks ONE instance's bits, uses whole 64-byte pieces of every line it fetches
//          (with one word per lane per row a 128-byte line held the words of 32 instances that are expanded at 32 different
//          times: 33x the bytes, 1.4 GB of re-reads per 1024-instance step).
// Hot code works on a LOCAL COPY of the sink (sha_block_w): through a reference, every store through a uint32_t* may alias the
// sink's own fields and forces them through memory around each append. On the device the sink always stores (kernels that do
// not want the bits run the value-only SHA instead).
#ifndef BLSW_BITS_CHUNK_WORDS
#define BLSW_BITS_CHUNK_WORDS 16
#endif
struct BitSink {
    uint32_t* out;    // host: nullptr = count only; device: unused
    uint64_t stride;  // host: distance (in u32) between consecutive words of this stream
    uint64_t acc;
    uint32_t fill;
    uint32_t widx;
    uint64_t nbits;
#if defined(__HIPCC__)  // members exist in both passes of a HIP translation unit; only the device pass uses them
    uint32_t* lds;  // this lane's column of the wave's [16][64] word buffer (LDS address space behind a generic pointer)
    uint4* gcur;    // where this lane's next 64-byte run goes
    BLSW_HD void init_device(uint32_t* lds_lane, uint4* first_run) {
        out = nullptr;
        stride = 0;
        acc = 0;
        fill = 0;
        widx = 0;
        nbits = 0;
        lds = lds_lane;
        gcur = first_run;
    }
    BLSW_HD void run_out() {  // the 16 words this lane collected -> one 64-byte run
        uint32_t w[BLSW_BITS_CHUNK_WORDS];
#pragma unroll
        for (int k = 0; k < BLSW_BITS_CHUNK_WORDS; k++) w[k] = lds[k * 64];
#pragma unroll
        for (int k = 0; k < BLSW_BITS_CHUNK_WORDS / 4; k++) gcur[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
        gcur += 64 * BLSW_BITS_CHUNK_WORDS / 4;  // next run of this lane: one tile row (64 lanes x one run) further
    }
#endif
    BLSW_HD void init(uint32_t* o, uint64_t s) {
        out = o;
        stride = s;
        acc = 0;
        fill = 0;
        widx = 0;
        nbits = 0;
    }
    BLSW_HD void word_out() {
#if defined(__HIP_DEVICE_COMPILE__)
        lds[(widx & (BLSW_BITS_CHUNK_WORDS - 1)) * 64] = (uint32_t)acc;
        widx++;
        if ((widx & (BLSW_BITS_CHUNK_WORDS - 1)) == 0) run_out();
#else
        if (out) out[(uint64_t)widx * stride] = (uint32_t)acc;
        widx++;
#endif
        acc >>= 32;
    }
    BLSW_HD void push(uint32_t bits, uint32_t n) {  // n <= 32, bits above n must be zero
#if !defined(__HIP_DEVICE_COMPILE__)
        nbits += n;  // host: the layout's bit count
#endif
        acc |= (uint64_t)bits << fill;
        fill += n;
        if (fill >= 32) {
            word_out();
            fill -= 32;
        }
    }
    BLSW_HD void push32(uint32_t bits) {  // fill is unchanged
#if !defined(__HIP_DEVICE_COMPILE__)
        nbits += 32;
#endif
        acc |= (uint64_t)bits << fill;
        word_out();
    }
    BLSW_HD void flush() {
        if (fill) {
            word_out();
            acc = 0;
            fill = 0;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if (widx & (BLSW_BITS_CHUNK_WORDS - 1)) run_out();  // the last, partial run (words beyond the stream are never read)
#endif
    }
};

BLSW_HD W32 w_xor(BitSink& s, const W32& a, const W32& b) {
    uint32_t wm = ~a.cm & ~b.cm;
    if (wm) s.push(pext32((a.v ^ a.nm) ^ (b.v ^ b.nm), wm), popc32(wm));
    W32 r;
    r.v = a.v ^ b.v;
    r.cm = a.cm & b.cm;
    r.nm = (a.nm ^ b.nm ^ (a.cm & a.v) ^ (b.cm & b.v)) & ~r.cm;
    return r;
}
BLSW_HD W32 w_and(BitSink& s, const W32& a, const W32& b) {
    uint32_t wm = ~a.cm & ~b.cm;
    if (wm) s.push(pext32(a.v & b.v, wm), popc32(wm));
    W32 r;
    r.v = a.v & b.v;
    uint32_t a_false = a.cm & ~a.v, b_false = b.cm & ~b.v, a_true = a.cm & a.v, b_true = b.cm & b.v;
    r.cm = (a.cm & b.cm) | a_false | b_false;
    r.nm = ((a_true & b.nm) | (b_true & a.nm)) & ~r.cm;
    return r;
}
// UInt32::addmany over k operands
BLSW_HD W32 w_addmany(BitSink& s, const W32* ops, int k) {
    uint64_t sum = 0;
    uint32_t allc = 0xffffffffu;
    for (int i = 0; i < k; i++) {
        sum += ops[i].v;
        allc &= ops[i].cm;
    }
    if (allc == 0xffffffffu) return w_const((uint32_t)sum);
    int nbits = (k == 2) ? 33 : (k <= 4 ? 34 : 35);
    s.push((uint32_t)sum, 32);
    s.push((uint32_t)(sum >> 32), nbits - 32);
    return {(uint32_t)sum, 0u, 0u};
}

#define BLSW_SHA_K                                                                                                                                  \
    {                                                                                                                                               \
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,         \
            0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,     \
            0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,     \
            0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,     \
            0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,     \
            0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2                              \
    }
#define BLSW_SHA_H0                                                                                          \
    {                                                                                                        \
        0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19 \
    }

// Sha256Gadget::update_state, any mix of constant / variable bits (mask tracking on every operation; the schedule lives in a
// dynamically indexed array). Reference statement of the gadget: sha_block_w below must emit the same bits.
BLSW_FN void sha_block_generic(BitSink& sink, W32 st[8], const W32 data[16]) {
    constexpr uint32_t K[64] = BLSW_SHA_K;
    BitSink s = sink;  // registers from here on (see BitSink)
    W32 w[64];
    for (int i = 0; i < 16; i++) w[i] = data[i];
#pragma unroll 1
    for (int i = 16; i < 64; i++) {
        W32 a1 = w_xor(s, w_rotr(w[i - 15], 7), w_rotr(w[i - 15], 18));
        W32 s0 = w_xor(s, a1, w_shr(w[i - 15], 3));
        W32 b1 = w_xor(s, w_rotr(w[i - 2], 17), w_rotr(w[i - 2], 19));
        W32 s1 = w_xor(s, b1, w_shr(w[i - 2], 10));
        W32 ops[4] = {w[i - 16], s0, w[i - 7], s1};
        w[i] = w_addmany(s, ops, 4);
    }
    W32 h[8];
    for (int i = 0; i < 8; i++) h[i] = st[i];
#pragma unroll 1
    for (int i = 0; i < 64; i++) {
        W32 c1 = w_and(s, h[4], h[5]);
        W32 c2 = w_and(s, w_not(h[4]), h[6]);
        W32 ch = w_xor(s, c1, c2);
        W32 m1 = w_and(s, h[0], h[1]);
        W32 m2 = w_and(s, h[0], h[2]);
        W32 m3 = w_and(s, h[1], h[2]);
        W32 m12 = w_xor(s, m1, m2);
        W32 ma = w_xor(s, m12, m3);
        W32 p1 = w_xor(s, w_rotr(h[0], 2), w_rotr(h[0], 13));
        W32 s0 = w_xor(s, p1, w_rotr(h[0], 22));
        W32 q1 = w_xor(s, w_rotr(h[4], 6), w_rotr(h[4], 11));
        W32 s1 = w_xor(s, q1, w_rotr(h[4], 25));
        W32 o5[5] = {h[7], s1, ch, w_const(K[i]), w[i]};
        W32 t0 = w_addmany(s, o5, 5);
        W32 o2[2] = {s0, ma};
        W32 t1 = w_addmany(s, o2, 2);
        h[7] = h[6];
        h[6] = h[5];
        h[5] = h[4];
        W32 o3[2] = {h[3], t0};
        h[4] = w_addmany(s, o3, 2);
        h[3] = h[2];
        h[2] = h[1];
        h[1] = h[0];
        W32 o4[2] = {t0, t1};
        h[0] = w_addmany(s, o4, 2);
    }
    for (int i = 0; i < 8; i++) {
        W32 o[2] = {st[i], h[i]};
        st[i] = w_addmany(s, o, 2);
    }
    sink = s;
}

// ---- fast path. Which bits are constants is the same for every lane, and after a few operations almost every word is
// "pure": all 32 bits variables (cm = nm = 0) or all constants. Pure words need no mask arithmetic and no bit extraction:
// an XOR / AND of two variable words is one 32-bit append, the sigma functions append 32 + 29 (or 22) bits, addmany appends
// 33 / 34 / 35 bits. The message schedule runs in a rolling 16-word window in registers (static indexing) and is recomputed,
// values only, during the rounds (the gadget allocates the whole schedule before the first round, so its witnesses cannot
// be produced on the fly); words with partially constant bits take the generic operation in place.
BLSW_HD bool w_is_const(const W32& a) { return a.cm == 0xffffffffu; }
BLSW_HD bool w_is_var(const W32& a) { return (a.cm | a.nm) == 0u; }
BLSW_HD uint32_t sigma_var(BitSink& s, uint32_t x, int r1, int r2, int sh) {  // both XOR witnesses of a sigma on a variable word
    const uint32_t a = rotr32(x, r1) ^ rotr32(x, r2);
    s.push32(a);
    const uint32_t r = a ^ (x >> sh);
    s.push(r & (0xffffffffu >> sh), 32 - sh);
    return r;
}
// one word of the message schedule: w[i] = addmany(w[i-16], sigma0(w[i-15]), w[i-7], sigma1(w[i-2]))
BLSW_HD W32 sha_sched_word(BitSink& s, const W32& w16, const W32& w15, const W32& w7, const W32& w2) {
    W32 s0, s1;
    if (w_is_var(w15))
        s0 = {sigma_var(s, w15.v, 7, 18, 3), 0u, 0u};
    else if (w_is_const(w15))
        s0 = w_const(rotr32(w15.v, 7) ^ rotr32(w15.v, 18) ^ (w15.v >> 3));
    else
        s0 = w_xor(s, w_xor(s, w_rotr(w15, 7), w_rotr(w15, 18)), w_shr(w15, 3));
    if (w_is_var(w2))
        s1 = {sigma_var(s, w2.v, 17, 19, 10), 0u, 0u};
    else if (w_is_const(w2))
        s1 = w_const(rotr32(w2.v, 17) ^ rotr32(w2.v, 19) ^ (w2.v >> 10));
    else
        s1 = w_xor(s, w_xor(s, w_rotr(w2, 17), w_rotr(w2, 19)), w_shr(w2, 10));
    const uint32_t sum32 = w16.v + s0.v + w7.v + s1.v;
    if ((w16.cm & s0.cm & w7.cm & s1.cm) == 0xffffffffu) return w_const(sum32);
    const uint64_t sum = (uint64_t)w16.v + s0.v + w7.v + s1.v;
    s.push32((uint32_t)sum);
    s.push((uint32_t)(sum >> 32), 2);
    return {sum32, 0u, 0u};
}
BLSW_HD uint32_t sha_sched_value(uint32_t w16, uint32_t w15, uint32_t w7, uint32_t w2) {
    return w16 + (rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3)) + w7 + (rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10));
}
// one round on mask-carrying state (the generic statement of a round; used for the first rounds on a constant state)
BLSW_HD void sha_round_generic(BitSink& s, W32 h[8], const W32& wi, uint32_t k) {
    W32 c1 = w_and(s, h[4], h[5]);
    W32 c2 = w_and(s, w_not(h[4]), h[6]);
    W32 ch = w_xor(s, c1, c2);
    W32 m1 = w_and(s, h[0], h[1]);
    W32 m2 = w_and(s, h[0], h[2]);
    W32 m3 = w_and(s, h[1], h[2]);
    W32 m12 = w_xor(s, m1, m2);
    W32 ma = w_xor(s, m12, m3);
    W32 p1 = w_xor(s, w_rotr(h[0], 2), w_rotr(h[0], 13));
    W32 s0 = w_xor(s, p1, w_rotr(h[0], 22));
    W32 q1 = w_xor(s, w_rotr(h[4], 6), w_rotr(h[4], 11));
    W32 s1 = w_xor(s, q1, w_rotr(h[4], 25));
    W32 o5[5] = {h[7], s1, ch, w_const(k), wi};
    W32 t0 = w_addmany(s, o5, 5);
    W32 o2[2] = {s0, ma};
    W32 t1 = w_addmany(s, o2, 2);
    h[7] = h[6];
    h[6] = h[5];
    h[5] = h[4];
    W32 o3[2] = {h[3], t0};
    h[4] = w_addmany(s, o3, 2);
    h[3] = h[2];
    h[2] = h[1];
    h[1] = h[0];
    W32 o4[2] = {t0, t1};
    h[0] = w_addmany(s, o4, 2);
}
// one round on a fully variable state: twelve 32-bit appends (ch: 3, maj: 5, Sigma0: 2, Sigma1: 2) and four addmany results
BLSW_HD void sha_round_var(BitSink& s, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t& e, uint32_t& f, uint32_t& g, uint32_t& h, uint32_t kw_lo,
                           uint32_t kw_hi) {
    const uint32_t c1 = e & f;
    s.push32(c1);
    const uint32_t c2 = ~e & g;
    s.push32(c2);
    const uint32_t ch = c1 ^ c2;
    s.push32(ch);
    const uint32_t m1 = a & b, m2 = a & c, m3 = b & c;
    s.push32(m1);
    s.push32(m2);
    s.push32(m3);
    const uint32_t m12 = m1 ^ m2;
    s.push32(m12);
    const uint32_t ma = m12 ^ m3;
    s.push32(ma);
    const uint32_t p1 = rotr32(a, 2) ^ rotr32(a, 13);
    s.push32(p1);
    const uint32_t s0 = p1 ^ rotr32(a, 22);
    s.push32(s0);
    const uint32_t q1 = rotr32(e, 6) ^ rotr32(e, 11);
    s.push32(q1);
    const uint32_t s1 = q1 ^ rotr32(e, 25);
    s.push32(s1);
    // t0 = addmany(h, s1, ch, K, w): 35 result bits; (kw_lo, kw_hi) = K + w as a 33-bit number
    const uint64_t t0 = (uint64_t)h + s1 + ch + kw_lo + ((uint64_t)kw_hi << 32);
    s.push32((uint32_t)t0);
    s.push((uint32_t)(t0 >> 32), 3);
    const uint64_t t1 = (uint64_t)s0 + ma;
    s.push32((uint32_t)t1);
    s.push((uint32_t)(t1 >> 32), 1);
    const uint64_t ne = (uint64_t)d + (uint32_t)t0;
    s.push32((uint32_t)ne);
    s.push((uint32_t)(ne >> 32), 1);
    const uint64_t na = (uint64_t)(uint32_t)t0 + (uint32_t)t1;
    s.push32((uint32_t)na);
    s.push((uint32_t)(na >> 32), 1);
    h = g;
    g = f;
    f = e;
    e = (uint32_t)ne;
    d = c;
    c = b;
    b = a;
    a = (uint32_t)na;
}
// Sha256Gadget::update_state
BLSW_FN void sha_block_w(BitSink& sink, W32 st[8], const W32 data[16]) {
    constexpr uint32_t K[64] = BLSW_SHA_K;
    // the state is fully variable (continuation blocks), or fully constant with a first data word that is not (first block
    // of a digest: a and e become variables in round 0, the whole state after four rounds); anything else: generic
    bool all_var = true, all_const = true;
    for (int i = 0; i < 8; i++) {
        all_var = all_var && w_is_var(st[i]);
        all_const = all_const && w_is_const(st[i]);
    }
    if (!(all_var || (all_const && !w_is_const(data[0])))) {
        sha_block_generic(sink, st, data);
        return;
    }
    BitSink s = sink;  // registers from here on (see BitSink)
    // ---- message schedule: all 48 words, in allocation order
    W32 win[16];
#pragma unroll
    for (int j = 0; j < 16; j++) win[j] = data[j];
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int j = 0; j < 16; j++) win[j] = sha_sched_word(s, win[j], win[(j + 1) & 15], win[(j + 9) & 15], win[(j + 14) & 15]);
    }
    // ---- rounds
    W32 h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = st[i];
    const int prefix = all_var ? 0 : 4;
    if (!all_var) {
#pragma unroll 1
        for (int r = 0; r < 4; r++) sha_round_generic(s, h, data[r], K[r]);
    }
    uint32_t a = h[0].v, b = h[1].v, cc = h[2].v, d = h[3].v, e = h[4].v, f = h[5].v, g = h[6].v, hh = h[7].v;
    uint32_t wv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) wv[j] = data[j].v;
#pragma unroll 1
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (c > 0) wv[j] = sha_sched_value(wv[j], wv[(j + 1) & 15], wv[(j + 9) & 15], wv[(j + 14) & 15]);
            if (c > 0 || j >= prefix) {
                const uint64_t kw = (uint64_t)K[16 * c + j] + wv[j];
                sha_round_var(s, a, b, cc, d, e, f, g, hh, (uint32_t)kw, (uint32_t)(kw >> 32));
            }
        }
    }
    const uint32_t fin[8] = {a, b, cc, d, e, f, g, hh};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t sum = (uint64_t)st[i].v + fin[i];  // addmany(st[i], h[i]): h[i] is a variable, 33 result bits
        s.push32((uint32_t)sum);
        s.push((uint32_t)(sum >> 32), 1);
        st[i] = {(uint32_t)sum, 0u, 0u};
    }
    sink = s;
}

#define BLSW_DST "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"
#define BLSW_DST_LEN 43

// byte k of msg_prime = Z_pad(64) | msg | I2OSP(256,2) (witness) | 0 | DST | len(DST), followed by SHA padding
BLSW_FN void b0_byte(const uint8_t* msg, uint32_t msg_len, bool msg_const, uint32_t k, uint32_t total, uint32_t& val, bool& konst) {
    const char dst[] = BLSW_DST;
    konst = true;
    val = 0;
    if (k < 64) return;
    uint32_t o = k - 64;
    if (o < msg_len) {
        val = msg[o];
        konst = msg_const;
        return;
    }
    o -= msg_len;
    if (o < 2) {  // lib_str = 0x0100, allocated as witness bytes (hasher.rs:130-132)
        val = (o == 0) ? 1 : 0;
        konst = false;
        return;
    }
    o -= 2;
    if (o == 0) return;  // the single zero byte
    o -= 1;
    if (o < BLSW_DST_LEN) {
        val = (uint8_t)dst[o];
        return;
    }
    if (o == BLSW_DST_LEN) {
        val = BLSW_DST_LEN;
        return;
    }
    // SHA padding of a `total`-byte message
    uint32_t padded = ((total + 9 + 63) / 64) * 64;
    if (k == total) {
        val = 0x80;
        return;
    }
    if (k >= padded - 8) {
        uint64_t bitlen = (uint64_t)total * 8;
        val = (uint32_t)(bitlen >> (8 * (padded - 1 - k))) & 0xff;
    }
}

// Runs the whole expand_message_xmd (len_in_bytes = 256) gadget for one message.
// Emits the 16 lib_str booleans first (they are allocated before the first digest), then every SHA witness.
// out_words: the 8 digests b1..b8 as 64 big-endian words (uniform_bytes).
BLSW_FN void expand_message_w(BitSink& s, const uint8_t* msg, uint32_t msg_len, bool msg_const, uint32_t uniform_words[64]) {
    constexpr uint32_t H0[8] = BLSW_SHA_H0;
    const char dst[] = BLSW_DST;
    // lib_str_var: two witness bytes, little-endian bit order each: 0x01, 0x00
    s.push(0x0001u, 16);
    // ---- b0
    W32 st[8];
    for (int i = 0; i < 8; i++) st[i] = w_const(H0[i]);
    uint32_t total = 64 + msg_len + 3 + BLSW_DST_LEN + 1;
    uint32_t nblocks = (total + 9 + 63) / 64;
#pragma unroll 1
    for (uint32_t blk = 0; blk < nblocks; blk++) {
        W32 data[16];
        for (int wi = 0; wi < 16; wi++) {
            uint32_t v = 0, cm = 0;
            for (int b = 0; b < 4; b++) {
                uint32_t bv;
                bool bc;
                b0_byte(msg, msg_len, msg_const, blk * 64 + wi * 4 + b, total, bv, bc);
                v |= bv << (8 * (3 - b));
                if (bc) cm |= 0xffu << (8 * (3 - b));
            }
            data[wi] = {v, cm, 0u};
        }
        sha_block_w(s, st, data);
    }
    W32 b0[8];
    for (int i = 0; i < 8; i++) b0[i] = st[i];
    // ---- b1 .. b8 : H(prev(32) | i | DST') ; 77 bytes -> 2 blocks
    W32 last[8];
#pragma unroll 1
    for (uint32_t i = 1; i <= 8; i++) {
        W32 in[8];
        if (i == 1) {
            for (int k = 0; k < 8; k++) in[k] = b0[k];
        } else {
            // bytewise UInt8 xor of b0 and b_(i-1): witnesses in byte order (big-endian bytes of each word),
            // little-endian bits inside a byte
            for (int k = 0; k < 8; k++) {
                const W32 &a = b0[k], &b = last[k];
                uint32_t wm = ~a.cm & ~b.cm;
                uint32_t x = (a.v ^ a.nm) ^ (b.v ^ b.nm);
                for (int by = 3; by >= 0; by--) {
                    uint32_t m8 = (wm >> (8 * by)) & 0xff;
                    if (m8) s.push(pext32((x >> (8 * by)) & 0xff, m8), popc32(m8));
                }
                in[k].v = a.v ^ b.v;
                in[k].cm = a.cm & b.cm;
                in[k].nm = (a.nm ^ b.nm ^ (a.cm & a.v) ^ (b.cm & b.v)) & ~in[k].cm;
            }
        }
        for (int k = 0; k < 8; k++) st[k] = w_const(H0[k]);
        // bytes 32..76: i, DST(43), 43 ; byte 77: 0x80 ; bytes 120..127: bit length 616
        uint8_t tailb[96];
        for (int k = 0; k < 96; k++) tailb[k] = 0;
        tailb[0] = (uint8_t)i;
        for (int k = 0; k < BLSW_DST_LEN; k++) tailb[1 + k] = (uint8_t)dst[k];
        tailb[1 + BLSW_DST_LEN] = BLSW_DST_LEN;
        tailb[2 + BLSW_DST_LEN] = 0x80;
        tailb[94] = (uint8_t)((77 * 8) >> 8);
        tailb[95] = (uint8_t)((77 * 8) & 0xff);
        W32 data[16];
        for (int k = 0; k < 8; k++) data[k] = in[k];
        for (int k = 0; k < 8; k++)
            data[8 + k] = w_const(((uint32_t)tailb[4 * k] << 24) | ((uint32_t)tailb[4 * k + 1] << 16) | ((uint32_t)tailb[4 * k + 2] << 8) | tailb[4 * k + 3]);
        sha_block_w(s, st, data);
        for (int k = 0; k < 16; k++)
            data[k] = w_const(((uint32_t)tailb[32 + 4 * k] << 24) | ((uint32_t)tailb[33 + 4 * k] << 16) | ((uint32_t)tailb[34 + 4 * k] << 8) | tailb[35 + 4 * k]);
        sha_block_w(s, st, data);
        for (int k = 0; k < 8; k++) {
            last[k] = st[k];
            uniform_words[(i - 1) * 8 + k] = st[k].v;
        }
    }
    s.flush();
}

// ---- value-only path: plain SHA-256 / expand_message_xmd (no witness bits). Used to hand u0, u1 to the
// map_to_curve chain immediately, while the witness-bit pass of the same message runs on another stream.
BLSW_FN void sha256_compress_plain(uint32_t st[8], const uint32_t w_in[16]) {
    constexpr uint32_t K[64] = BLSW_SHA_K;
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = w_in[i];
#pragma unroll 1
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll 1
    for (int i = 0; i < 64; i++) {
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t ma = (a & b) ^ (a & c) ^ (b & c);
        uint32_t s0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t s1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t t0 = h + s1 + ch + K[i] + w[i];
        uint32_t t1 = s0 + ma;
        h = g;
        g = f;
        f = e;
        e = d + t0;
        d = c;
        c = b;
        b = a;
        a = t0 + t1;
    }
    st[0] += a;
    st[1] += b;
    st[2] += c;
    st[3] += d;
    st[4] += e;
    st[5] += f;
    st[6] += g;
    st[7] += h;
}
BLSW_FN void expand_message_values(const uint8_t* msg, uint32_t msg_len, uint32_t uniform_words[64]) {
    constexpr uint32_t H0[8] = BLSW_SHA_H0;
    const char dst[] = BLSW_DST;
    uint32_t st[8];
    for (int i = 0; i < 8; i++) st[i] = H0[i];
    uint32_t total = 64 + msg_len + 3 + BLSW_DST_LEN + 1;
    uint32_t nblocks = (total + 9 + 63) / 64;
#pragma unroll 1
    for (uint32_t blk = 0; blk < nblocks; blk++) {
        uint32_t data[16];
        for (int wi = 0; wi < 16; wi++) {
            uint32_t v = 0;
            for (int b = 0; b < 4; b++) {
                uint32_t bv;
                bool bc;
                b0_byte(msg, msg_len, false, blk * 64 + wi * 4 + b, total, bv, bc);
                v |= bv << (8 * (3 - b));
            }
            data[wi] = v;
        }
        sha256_compress_plain(st, data);
    }
    uint32_t b0[8], last[8];
    for (int i = 0; i < 8; i++) b0[i] = st[i];
    uint8_t tailb[96];
    for (int k = 0; k < 96; k++) tailb[k] = 0;
    for (int k = 0; k < BLSW_DST_LEN; k++) tailb[1 + k] = (uint8_t)dst[k];
    tailb[1 + BLSW_DST_LEN] = BLSW_DST_LEN;
    tailb[2 + BLSW_DST_LEN] = 0x80;
    tailb[94] = (uint8_t)((77 * 8) >> 8);
    tailb[95] = (uint8_t)((77 * 8) & 0xff);
#pragma unroll 1
    for (uint32_t i = 1; i <= 8; i++) {
        uint32_t data[16];
        for (int k = 0; k < 8; k++) data[k] = (i == 1) ? b0[k] : (b0[k] ^ last[k]);
        tailb[0] = (uint8_t)i;
        for (int k = 0; k < 8; k++)
            data[8 + k] = ((uint32_t)tailb[4 * k] << 24) | ((uint32_t)tailb[4 * k + 1] << 16) | ((uint32_t)tailb[4 * k + 2] << 8) | tailb[4 * k + 3];
        for (int k = 0; k < 8; k++) st[k] = H0[k];
        sha256_compress_plain(st, data);
        for (int k = 0; k < 16; k++)
            data[k] = ((uint32_t)tailb[32 + 4 * k] << 24) | ((uint32_t)tailb[33 + 4 * k] << 16) | ((uint32_t)tailb[34 + 4 * k] << 8) | tailb[35 + 4 * k];
        sha256_compress_plain(st, data);
        for (int k = 0; k < 8; k++) {
            last[k] = st[k];
            uniform_words[(i - 1) * 8 + k] = st[k];
        }
    }
}

// hash_to_field (hasher.rs:58-107): element j (0..3) = OS2IP(uniform_bytes[64j .. 64j+64)) mod p as
// head(47 high bytes) * 256^17 + tail(17 low bytes); linear combinations only (no witnesses).
BLSW_FN Fp hash_to_field_elem(const uint32_t* W /*16 big-endian words*/) {
    constexpr uint32_t R2[12] = BLSW_R2_LIMBS;
    uint32_t L[16];
    for (int i = 0; i < 16; i++) L[i] = W[15 - i];
    Fp tail = fp_zero(), head = fp_zero(), r2;
    for (int i = 0; i < 12; i++) r2.l[i] = R2[i];
    for (int i = 0; i < 4; i++) tail.l[i] = L[i];
    tail.l[4] = L[4] & 0xff;
    for (int k = 0; k < 11; k++) head.l[k] = (L[k + 4] >> 8) | (L[k + 5] << 24);
    head.l[11] = L[15] >> 8;
    Fp two136 = fp_zero();  // 2^136 as a canonical integer
    two136.l[4] = 1u << 8;
    Fp hm = fp_mul(head, r2), tm = fp_mul(tail, r2), km = fp_mul(two136, r2);
    return fp_add(fp_mul(hm, km), tm);
}

}  // namespace blsw
