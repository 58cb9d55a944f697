// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Restatement of the reference's hot path:
//   src/hasher.rs:37-174   DefaultFieldHasherWithCons::{expand, hash_to_field}
//   src/hasher.rs:176-207  DensePolynomialVar::evaluate
//   src/hasher.rs:228-348  CurveMapperWithCons::{new, map_to_curve, isogeny_map}
//   src/hasher.rs:352-559  map_to_curve_9mod16, cmov, is_zero, sgn0, pow, to_projective_short
//   src/hasher.rs:569-583  to_affine_unchecked
//   src/hasher.rs:641-673  MapToCurveHasherWithCons::{hash, clear_cofactor2}
//   src/hasher.rs:727-740  hash_to_g2_with_cons
//   src/constraints.rs:90-128  BlsSignatureVerifyGadget::verify
//   src/constraints.rs:335-366 allocation order of the test circuit (msg, params, pk, sig)
// plus the third-party (not vendored) pieces it calls:
//   ark-crypto-primitives ^0.4.0 crh/sha256/constraints.rs (Sha256Gadget)        — SURVEY App. A.4
//   ark-r1cs-std ^0.4.0 pairing/bls12/mod.rs, groups/bls12/mod.rs (G2PreparedVar) — SURVEY App. A.7-A.9
#pragma once
#include <algorithm>
#include "curves_var.h"

namespace orc {

// ------------------------------------------------------------------ native SHA-256 (FIPS 180-4)
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static const uint32_t SHA_H0[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
inline void sha256_compress(uint32_t st[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t h[8];
    memcpy(h, st, 32);
    for (int i = 0; i < 64; i++) {
        uint32_t ch = (h[4] & h[5]) ^ (~h[4] & h[6]);
        uint32_t ma = (h[0] & h[1]) ^ (h[0] & h[2]) ^ (h[1] & h[2]);
        uint32_t s0 = rotr32(h[0], 2) ^ rotr32(h[0], 13) ^ rotr32(h[0], 22);
        uint32_t s1 = rotr32(h[4], 6) ^ rotr32(h[4], 11) ^ rotr32(h[4], 25);
        uint32_t t0 = h[7] + s1 + ch + SHA_K[i] + w[i];
        uint32_t t1 = s0 + ma;
        h[7] = h[6];
        h[6] = h[5];
        h[5] = h[4];
        h[4] = h[3] + t0;
        h[3] = h[2];
        h[2] = h[1];
        h[1] = h[0];
        h[0] = t0 + t1;
    }
    for (int i = 0; i < 8; i++) st[i] += h[i];
}
inline void sha256(const uint8_t* data, size_t n, uint8_t out[32]) {
    uint32_t st[8];
    memcpy(st, SHA_H0, 32);
    std::vector<uint8_t> buf(data, data + n);
    buf.push_back(0x80);
    while (buf.size() % 64 != 56) buf.push_back(0);
    uint64_t bl = (uint64_t)n * 8;
    for (int i = 7; i >= 0; i--) buf.push_back((uint8_t)(bl >> (8 * i)));
    for (size_t o = 0; o < buf.size(); o += 64) sha256_compress(st, buf.data() + o);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(st[i] >> (24 - 8 * j));
}

// ------------------------------------------------------------------ Sha256Gadget  [ark-crypto-primitives 0.4 crh/sha256/constraints.rs]
inline void sha_update_state(U32 state[8], const U8* data /*64 bytes*/) {
    opcount().sha_blocks++;
    std::vector<U32> w(64, u32const(0));
    for (int i = 0; i < 16; i++) w[i] = u32from_bytes_be(data + 4 * i);
    for (int i = 16; i < 64; i++) {
        U32 a1 = u32xor(u32rotr(w[i - 15], 7), u32rotr(w[i - 15], 18));
        U32 s0 = u32xor(a1, u32shr(w[i - 15], 3));
        U32 b1 = u32xor(u32rotr(w[i - 2], 17), u32rotr(w[i - 2], 19));
        U32 s1 = u32xor(b1, u32shr(w[i - 2], 10));
        w[i] = u32addmany({w[i - 16], s0, w[i - 7], s1});
    }
    U32 h[8];
    for (int i = 0; i < 8; i++) h[i] = state[i];
    for (int i = 0; i < 64; i++) {
        U32 c1 = u32and(h[4], h[5]);
        U32 c2 = u32and(u32not(h[4]), h[6]);
        U32 ch = u32xor(c1, c2);
        U32 m1 = u32and(h[0], h[1]);
        U32 m2 = u32and(h[0], h[2]);
        U32 m3 = u32and(h[1], h[2]);
        U32 m12 = u32xor(m1, m2);
        U32 ma = u32xor(m12, m3);
        U32 p1 = u32xor(u32rotr(h[0], 2), u32rotr(h[0], 13));
        U32 s0 = u32xor(p1, u32rotr(h[0], 22));
        U32 q1 = u32xor(u32rotr(h[4], 6), u32rotr(h[4], 11));
        U32 s1 = u32xor(q1, u32rotr(h[4], 25));
        U32 t0 = u32addmany({h[7], s1, ch, u32const(SHA_K[i]), w[i]});
        U32 t1 = u32addmany({s0, ma});
        h[7] = h[6];
        h[6] = h[5];
        h[5] = h[4];
        h[4] = u32addmany({h[3], t0});
        h[3] = h[2];
        h[2] = h[1];
        h[1] = h[0];
        h[0] = u32addmany({t0, t1});
    }
    for (int i = 0; i < 8; i++) state[i] = u32addmany({state[i], h[i]});
}
// Sha256Gadget::digest = default().update(data).finalize()
inline std::vector<U8> sha256_gadget_digest(const std::vector<U8>& data) {
    U32 state[8];
    for (int i = 0; i < 8; i++) state[i] = u32const(SHA_H0[i]);
    std::vector<U8> buf = data;
    uint64_t bitlen = (uint64_t)data.size() * 8;
    buf.push_back(u8const(0x80));
    while (buf.size() % 64 != 56) buf.push_back(u8const(0));
    for (int i = 7; i >= 0; i--) buf.push_back(u8const((uint8_t)(bitlen >> (8 * i))));
    for (size_t o = 0; o < buf.size(); o += 64) sha_update_state(state, buf.data() + o);
    std::vector<U8> out(32);
    for (int i = 0; i < 8; i++) u32to_bytes_be(state[i], out.data() + 4 * i);
    return out;
}

// ------------------------------------------------------------------ hasher.rs
static const char* BLS_DST = "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_";  // hasher.rs:734

// hasher.rs:110-173  expand_message_xmd; `lib_str` is allocated as WITNESS bytes (hasher.rs:130-132)
inline std::vector<U8> hasher_expand(const std::vector<U8>& message, const std::vector<U8>& dst, size_t len_in_bytes) {
    const size_t b_len = 32;
    size_t ell = (len_in_bytes + b_len - 1) / b_len;
    assert(ell <= 255 && len_in_bytes <= 65535);
    std::vector<U8> dst_prime = dst;
    dst_prime.push_back(u8const((uint8_t)dst.size()));
    std::vector<U8> z_pad(64, u8const(0));
    uint8_t lib_str[2] = {(uint8_t)(len_in_bytes >> 8), (uint8_t)(len_in_bytes & 0xff)};
    std::vector<U8> lib_str_var = u8witness_vec(lib_str, 2);
    std::vector<U8> msg_prime = z_pad;
    msg_prime.insert(msg_prime.end(), message.begin(), message.end());
    msg_prime.insert(msg_prime.end(), lib_str_var.begin(), lib_str_var.end());
    msg_prime.push_back(u8const(0));
    msg_prime.insert(msg_prime.end(), dst_prime.begin(), dst_prime.end());
    std::vector<U8> b0 = sha256_gadget_digest(msg_prime);
    std::vector<U8> data = b0;
    data.push_back(u8const(1));
    data.insert(data.end(), dst_prime.begin(), dst_prime.end());
    std::vector<U8> b1 = sha256_gadget_digest(data);
    std::vector<U8> ret = b1, last_b = b1;
    for (size_t i = 2; i <= ell; i++) {
        std::vector<U8> bx(32);
        for (int k = 0; k < 32; k++) bx[k] = u8xor(b0[k], last_b[k]);
        bx.push_back(u8const((uint8_t)i));
        bx.insert(bx.end(), dst_prime.begin(), dst_prime.end());
        std::vector<U8> bi = sha256_gadget_digest(bx);
        ret.insert(ret.end(), bi.begin(), bi.end());
        last_b = bi;
    }
    ret.resize(len_in_bytes);
    return ret;
}
// hasher.rs:58-107  hash_to_field(msg, 2) over Fp2; linear combinations only
inline std::vector<Fp2Var> hasher_hash_to_field(const std::vector<U8>& message, const std::vector<U8>& dst, size_t len_per_base_elem = 64) {
    const size_t count = 2, m = 2;
    std::vector<U8> uniform = hasher_expand(message, dst, count * m * len_per_base_elem);
    std::vector<Fp2Var> out;
    Fp c256 = fp_from_u64(256);
    for (size_t i = 0; i < count; i++) {
        FpVar e[2];
        for (size_t j = 0; j < m; j++) {
            size_t off = len_per_base_elem * (j + i * m);
            std::vector<U8> le(uniform.begin() + off, uniform.begin() + off + len_per_base_elem);
            std::reverse(le.begin(), le.end());
            const size_t pos = 47;  // (381-1)/8
            size_t tail_len = le.size() - pos;
            FpVar f_head = le_bytes_to_fp_var(le.data() + tail_len, pos);
            FpVar f_tail = le_bytes_to_fp_var(le.data(), tail_len);
            FpVar f = f_head;
            for (size_t l = 0; l < tail_len; l++) f = fmulc(f, c256);
            f = fadd(f, f_tail);
            e[j] = f;
        }
        out.push_back({e[0], e[1]});
    }
    return out;
}

struct MapperConsts {
    Fp2 A, B, Z, C2, C3, C4, C5;
    std::vector<uint8_t> c1_bits_be;  // hasher.rs:242 exponent, as iterated by pow (hasher.rs:532-548)
    Fp2 xnum[4], xden[3], ynum[4], yden[4];
    std::vector<bool> h_eff_bits_le;  // hasher.rs:666, little-endian, 640 bits
    MapperConsts() {
        A = {fp_zero(), fp_from_u64(240)};
        B = {fp_from_u64(1012), fp_from_u64(1012)};
        Z = {fp_neg(fp_from_u64(2)), fp_neg(fp_from_u64(1))};
        C2 = {fp_zero(), fp_one()};
        C3 = {fp_from_dec("2973677408986561043442465346520108879172042883009249989176415018091420807192182638567116318576472649347015917690530"),
              fp_from_dec("1028732146235106349975324479215795277384839936929757896155643118032610843298655225875571310552543014690878354869257")};
        C4 = {fp_from_dec("1015919005498129635886032702454337503112659152043614931979881174103627376789972962005013361970813319613593700736144"),
              fp_from_dec("1244231661155348484223428017511856347821538750986231559855759541903146219579071812422210818684355842447591283616181")};
        C5 = {fp_from_dec("1637752706019426886789797193293828301565549384974986623510918743054325021588194075665960171838131772227885159387073"),
              fp_from_dec("2356393562099837637521906572659114847248791943663835535137223682689832134851362912628461394915339516530489788841108")};
        const char* c1 =
            "2a437a4b8c35fc74bd278eaa22f25e9e2dc90e50e7046b466e59e49349e8bd050a62cfd16ddca6ef53149330978ef011d68619c86185c7b292e85a87091a04966bf9"
            "1ed3e71b743162c338362113cfd7ced6b1d76382eab26aa00001c718e3";
        // from_hex -> bytes (big-endian), reverse -> constant_vec, to_bits_be(): per UInt8 bits reversed... the net
        // effect (hasher.rs:533-536) is: bytes reversed, then each byte's bits most-significant first? No:
        // Vec<UInt8>::to_bits_be = to_bits_le() reversed as a whole, so the iteration is the ORIGINAL hex string
        // most-significant bit first. (test_pow, hasher.rs:927: pow(2, "000014") == 2^20 pins this.)
        for (const char* s = c1; *s; s++) {
            int v = (*s <= '9') ? *s - '0' : *s - 'a' + 10;
            for (int k = 3; k >= 0; k--) c1_bits_be.push_back((v >> k) & 1);
        }
        auto H = [](const char* a, const char* b) { return Fp2{fp_from_hex(a), fp_from_hex(b)}; };
        // 3-isogeny coefficients of ark-bls12-381 g2 WBConfig::ISOGENY_MAP (not in the reference)
        const char* k_1_0 = "5c759507e8e333ebb5b7a9a47d7ed8532c52d39fd3a042a88b58423c50ae15d5c2638e343d9c71c6238aaaaaaaa97d6";
        xnum[0] = H(k_1_0, k_1_0);
        xnum[1] = H("0", "11560bf17baa99bc32126fced787c88f984f87adf7ae0c7f9a208c6b4f20a4181472aaa9cb8d555526a9ffffffffc71a");
        xnum[2] = H("11560bf17baa99bc32126fced787c88f984f87adf7ae0c7f9a208c6b4f20a4181472aaa9cb8d555526a9ffffffffc71e",
                    "8ab05f8bdd54cde190937e76bc3e447cc27c3d6fbd7063fcd104635a790520c0a395554e5c6aaaa9354ffffffffe38d");
        xnum[3] = H("171d6541fa38ccfaed6dea691f5fb614cb14b4e7f4e810aa22d6108f142b85757098e38d0f671c7188e2aaaaaaaa5ed1", "0");
        xden[0] = H("0", "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaa63");
        xden[1] = H("c", "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaa9f");
        xden[2] = H("1", "0");
        const char* k_3_0 = "1530477c7ab4113b59a4c18b076d11930f7da5d4a07f649bf54439d87d27e500fc8c25ebf8c92f6812cfc71c71c6d706";
        ynum[0] = H(k_3_0, k_3_0);
        ynum[1] = H("0", "5c759507e8e333ebb5b7a9a47d7ed8532c52d39fd3a042a88b58423c50ae15d5c2638e343d9c71c6238aaaaaaaa97be");
        ynum[2] = H("11560bf17baa99bc32126fced787c88f984f87adf7ae0c7f9a208c6b4f20a4181472aaa9cb8d555526a9ffffffffc71c",
                    "8ab05f8bdd54cde190937e76bc3e447cc27c3d6fbd7063fcd104635a790520c0a395554e5c6aaaa9354ffffffffe38f");
        ynum[3] = H("124c9ad43b6cf79bfbf7043de3811ad0761b0f37a1e26286b0e977c69aa274524e79097a56dc4bd9e1b371c71c718b10", "0");
        const char* k_4_0 = "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffa8fb";
        yden[0] = H(k_4_0, k_4_0);
        yden[1] = H("0", "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffa9d3");
        yden[2] = H("12", "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaa99");
        yden[3] = H("1", "0");
        const char* h_eff =
            "0bc69f08f2ee75b3584c6a0ea91b352888e2a8e9145ad7689986ff031508ffe1329c2f178731db956d82bf015d1212b02ec0ec69d7477c1ae954cbc06689f6a359"
            "894c0adebbf6b4e8020005aaa95551";
        std::vector<bool> be;
        for (const char* s = h_eff; *s; s++) {
            int v = (*s <= '9') ? *s - '0' : *s - 'a' + 10;
            for (int k = 3; k >= 0; k--) be.push_back((v >> k) & 1);
        }
        h_eff_bits_le.assign(be.rbegin(), be.rend());
    }
};
inline const MapperConsts& mapper_consts() {
    static MapperConsts c;
    return c;
}

typedef ProjectiveVar<Fp2T> G2Var;
typedef ProjectiveVar<FpT> G1Var;

// hasher.rs:195-206
inline Fp2Var poly_evaluate(const Fp2* coeffs, int n, const Fp2Var& point) {
    Fp2Var result = f2zero();
    Fp2Var curr_pow_x = f2one();
    for (int i = 0; i < n; i++) {
        Fp2Var term = f2mul(curr_pow_x, f2const(coeffs[i]));
        result = f2add(result, term);
        curr_pow_x = f2mul(curr_pow_x, point);
    }
    return result;
}
// hasher.rs:569-583
inline void to_affine_unchecked(const G2Var& p, Fp2Var& x, Fp2Var& y) {
    Fp2Var z_inv = f2inv(p.z);
    Fp2Var z_inv_2 = f2sqr(z_inv);
    Fp2Var z_inv_3 = f2mul(z_inv_2, z_inv);
    x = f2mul(p.x, z_inv_2);
    y = f2mul(p.y, z_inv_3);
}
// hasher.rs:294-348
inline G2Var isogeny_map(const G2Var& point) {
    const MapperConsts& K = mapper_consts();
    Bool is_infinity = f2is_zero(point.z);
    Fp2Var x, y;
    to_affine_unchecked(point, x, y);
    Fp2Var x_den_at_x = poly_evaluate(K.xden, 3, x);
    Fp2Var x_den_inv = f2inv(x_den_at_x);
    Fp2Var y_den_at_x = poly_evaluate(K.yden, 4, x);
    Fp2Var y_den_inv = f2inv(y_den_at_x);
    Fp2Var x_num_at_x = poly_evaluate(K.xnum, 4, x);
    Fp2Var y_num_at_x = poly_evaluate(K.ynum, 4, x);
    Fp2Var img_x = f2mul(x_num_at_x, x_den_inv);
    Fp2Var t = f2mul(y_num_at_x, y);
    Fp2Var img_y = f2mul(t, y_den_inv);
    G2Var projective = {img_x, img_y, f2one()};
    G2Var zero = {f2zero(), f2zero(), f2zero()};
    return pv_select<Fp2T>(is_infinity, zero, projective);
}
// hasher.rs:520-530
inline Bool mapper_sgn0(const Fp2Var& v) {
    std::vector<Bool> c0_bits = fto_bits_le(v.c0);
    std::vector<Bool> c1_bits = fto_bits_le(v.c1);
    Bool sign_0 = c0_bits[0];
    Bool zero_0 = fis_eq(v.c0, fconst(fp_zero()));
    Bool sign_1 = c1_bits[0];
    Bool r = band(zero_0, sign_1);
    return bor(sign_0, r);
}
// hasher.rs:532-548
inline Fp2Var mapper_pow(const Fp2Var& v, const std::vector<uint8_t>& bits_be) {
    Fp2Var one = f2one();
    Fp2Var r = one;
    for (uint8_t bit : bits_be) {
        r = f2sqr(r);
        const Fp2Var& tv = bit ? v : one;  // select on a constant bit
        r = f2mul(r, tv);
    }
    return r;
}
inline Fp2Var cmov(const Fp2Var& f, const Fp2Var& t, const Bool& cond) { return f2select(cond, t, f); }
// hasher.rs:352-502 + to_projective_short (hasher.rs:551-559)
inline G2Var map_to_curve_9mod16(const Fp2Var& u) {
    const MapperConsts& K = mapper_consts();
    Fp2Var Z = f2const(K.Z), A = f2const(K.A), B = f2const(K.B), C2 = f2const(K.C2), C3 = f2const(K.C3), C4 = f2const(K.C4), C5 = f2const(K.C5);
    Fp2Var tv1 = f2sqr(u);                      // 1
    Fp2Var tv3 = f2mul(Z, tv1);                 // 2
    Fp2Var tv5 = f2sqr(tv3);                    // 3
    Fp2Var xd = f2add(tv5, tv3);                // 4
    Fp2Var x1n = f2add(xd, f2one());            // 5
    x1n = f2mul(x1n, B);                        // 6
    xd = f2mul(f2neg(A), xd);                   // 7
    Bool e1 = f2is_zero(xd);                    // 8
    xd = cmov(xd, f2mul(Z, A), e1);             // 9
    Fp2Var tv2 = f2sqr(xd);                     // 10
    Fp2Var gxd = f2mul(tv2, xd);                // 11
    tv2 = f2mul(A, tv2);                        // 12
    Fp2Var gx1 = f2add(f2sqr(x1n), tv2);        // 13,14
    gx1 = f2mul(gx1, x1n);                      // 15
    tv2 = f2mul(B, gxd);                        // 16
    gx1 = f2add(gx1, tv2);                      // 17
    Fp2Var tv4 = f2sqr(gxd);                    // 18
    tv2 = f2mul(tv4, gxd);                      // 19
    tv4 = f2sqr(tv4);                           // 20
    tv2 = f2mul(tv2, tv4);                      // 21
    tv2 = f2mul(tv2, gx1);                      // 22
    tv4 = f2sqr(tv4);                           // 23
    tv4 = f2mul(tv2, tv4);                      // 24
    Fp2Var y = mapper_pow(tv4, K.c1_bits_be);   // 25
    y = f2mul(y, tv2);                          // 26
    tv4 = f2mul(y, C2);                         // 27
    tv2 = f2sqr(tv4);                           // 28
    tv2 = f2mul(tv2, gxd);                      // 29
    Bool e2 = f2is_eq(tv2, gx1);                // 30
    y = cmov(y, tv4, e2);                       // 31
    tv4 = f2mul(y, C3);                         // 32
    tv2 = f2sqr(tv4);                           // 33
    tv2 = f2mul(tv2, gxd);                      // 34
    Bool e3 = f2is_eq(tv2, gx1);                // 35
    y = cmov(y, tv4, e3);                       // 36
    tv4 = f2mul(tv4, C2);                       // 37
    tv2 = f2sqr(tv4);                           // 38
    tv2 = f2mul(tv2, gxd);                      // 39
    Bool e4 = f2is_eq(tv2, gx1);                // 40
    y = cmov(y, tv4, e4);                       // 41
    Fp2Var gx2 = f2mul(gx1, tv5);               // 42
    gx2 = f2mul(gx2, tv3);                      // 43
    tv5 = f2mul(y, tv1);                        // 44
    tv5 = f2mul(tv5, u);                        // 45
    tv1 = f2mul(tv5, C4);                       // 46
    tv4 = f2mul(tv1, C2);                       // 47
    tv2 = f2sqr(tv4);                           // 48
    tv2 = f2mul(tv2, gxd);                      // 49
    Bool e5 = f2is_eq(tv2, gx2);                // 50
    tv1 = cmov(tv1, tv4, e5);                   // 51
    tv4 = f2mul(tv5, C5);                       // 52
    tv2 = f2sqr(tv4);                           // 53
    tv2 = f2mul(tv2, gxd);                      // 54
    Bool e6 = f2is_eq(tv2, gx2);                // 55
    tv1 = cmov(tv1, tv4, e6);                   // 56
    tv4 = f2mul(tv4, C2);                       // 57
    tv2 = f2sqr(tv4);                           // 58
    tv2 = f2mul(tv2, gxd);                      // 59
    Bool e7 = f2is_eq(tv2, gx2);                // 60
    tv1 = cmov(tv1, tv4, e7);                   // 61
    tv2 = f2sqr(y);                             // 62
    tv2 = f2mul(tv2, gxd);                      // 63
    Bool e8 = f2is_eq(tv2, gx1);                // 64
    y = cmov(tv1, y, e8);                       // 65
    tv2 = f2mul(tv3, x1n);                      // 66
    Fp2Var xn = cmov(tv2, x1n, e8);             // 67
    Bool sgn0_u = mapper_sgn0(u);               // 68
    Bool sgn0_y = mapper_sgn0(y);
    Bool e9 = bis_eq(sgn0_u, sgn0_y);
    Fp2Var y_neg = f2neg(y);
    y = cmov(y_neg, y, e9);                     // 69
    // to_projective_short(xd, xn, y)
    Fp2Var xd2 = f2sqr(xd);
    Fp2Var xd3 = f2mul(xd2, xd);
    Fp2Var px = f2mul(xn, xd);
    Fp2Var py = f2mul(y, xd3);
    return {px, py, xd};
}
inline G2Var map_to_curve(const Fp2Var& u) { return isogeny_map(map_to_curve_9mod16(u)); }
// hasher.rs:664-673
inline G2Var clear_cofactor2(const G2Var& p) {
    const MapperConsts& K = mapper_consts();
    std::vector<Bool> bits;
    for (bool b : K.h_eff_bits_le) bits.push_back(bconst(b));
    return pv_scalar_mul_le<Fp2T>(p, bits);
}
struct HashTrace {
    Fp2 u[2];
    G2Aff q[2], r, h;
};
// hasher.rs:641-661 / 727-740
inline G2Var hash_to_g2_with_cons(const std::vector<U8>& message, HashTrace* tr = nullptr) {
    std::vector<U8> dst = u8const_vec((const uint8_t*)BLS_DST, strlen(BLS_DST));
    CSREF.mark("hash.expand");
    std::vector<Fp2Var> u = hasher_hash_to_field(message, dst);
    CSREF.mark("hash.map0");
    G2Var q0 = map_to_curve(u[0]);
    CSREF.mark("hash.map1");
    G2Var q1 = map_to_curve(u[1]);
    CSREF.mark("hash.add");
    G2Var r = pv_add<Fp2T>(q0, q1);
    CSREF.mark("hash.clear_cofactor");
    G2Var h = clear_cofactor2(r);
    if (tr) {
        tr->u[0] = u[0].val();
        tr->u[1] = u[1].val();
        tr->q[0] = q0.value_affine();
        tr->q[1] = q1.value_affine();
        tr->r = r.value_affine();
        tr->h = h.value_affine();
    }
    return h;
}

// ------------------------------------------------------------------ pairing  [ark-r1cs-std pairing/bls12/mod.rs, groups/bls12/mod.rs]
struct G2Prepared {
    std::vector<std::pair<Fp2Var, Fp2Var>> ell_coeffs;
};
inline std::pair<Fp2Var, Fp2Var> g2prep_double(Fp2Var& rx, Fp2Var& ry, const Fp& two_inv) {
    Fp2Var a = f2inv(ry);
    Fp2Var b = f2sqr(rx);
    Fp2Var b_tmp = b;
    b = f2mul_fp_const(b, two_inv);
    b = f2add(b, b_tmp);
    Fp2Var c = f2mul(a, b);
    Fp2Var d = f2dbl(rx);
    Fp2Var x3 = f2sub(f2sqr(c), d);
    Fp2Var cx = f2mul(c, rx);
    Fp2Var e = f2sub(cx, ry);
    Fp2Var c_x3 = f2mul(c, x3);
    Fp2Var y3 = f2sub(e, c_x3);
    Fp2Var f = f2neg(c);
    rx = x3;
    ry = y3;
    return {e, f};  // M-twist: (e, -c)
}
inline std::pair<Fp2Var, Fp2Var> g2prep_add(Fp2Var& rx, Fp2Var& ry, const Fp2Var& qx, const Fp2Var& qy) {
    Fp2Var a = f2inv(f2sub(qx, rx));
    Fp2Var b = f2sub(qy, ry);
    Fp2Var c = f2mul(a, b);
    Fp2Var d = f2add(rx, qx);
    Fp2Var x3 = f2sub(f2sqr(c), d);
    Fp2Var e = f2mul(f2sub(rx, x3), c);
    Fp2Var y3 = f2sub(e, ry);
    Fp2Var cr = f2mul(c, rx);
    Fp2Var g = f2sub(cr, ry);
    Fp2Var f = f2neg(c);
    rx = x3;
    ry = y3;
    return {g, f};  // M-twist: (g, -c)
}
inline G2Prepared g2_prepare(const G2Var& q_) {
    AffineVar<Fp2T> q = pv_to_affine<Fp2T>(q_);
    Fp two_inv = fp_inv(fp_from_u64(2));
    benforce_not_equal_const_true(q.infinity);
    G2Prepared out;
    Fp2Var rx = q.x, ry = q.y;
    for (int i = 62; i >= 0; i--) {  // BitIteratorBE::new(X).skip(1)
        out.ell_coeffs.push_back(g2prep_double(rx, ry, two_inv));
        if ((BLS_X >> i) & 1) out.ell_coeffs.push_back(g2prep_add(rx, ry, q.x, q.y));
    }
    return out;
}
struct G1Prepared {
    AffineVar<FpT> p;
};
inline G1Prepared g1_prepare(const G1Var& p) { return {pv_to_affine<FpT>(p)}; }
inline void pairing_ell(Fp12Var& f, const std::pair<Fp2Var, Fp2Var>& coeffs, const AffineVar<FpT>& p) {
    Fp2Var c0 = coeffs.first;
    Fp2Var c1 = coeffs.second;
    Fp2Var c2 = {p.y, fconst(fp_zero())};
    FpVar k0 = fmul(c1.c0, p.x);
    FpVar k1 = fmul(c1.c1, p.x);
    c1 = {k0, k1};
    f = f12mul_by_014(f, c0, c1, c2);
}
inline Fp12Var miller_loop(const std::vector<G1Prepared>& ps, const std::vector<G2Prepared>& qs) {
    std::vector<size_t> idx(ps.size(), 0);
    Fp12Var f = f12one();
    for (int i = 62; i >= 0; i--) {
        f = f12sqr(f);
        for (size_t k = 0; k < ps.size(); k++) pairing_ell(f, qs[k].ell_coeffs[idx[k]++], ps[k].p);
        if ((BLS_X >> i) & 1)
            for (size_t k = 0; k < ps.size(); k++) pairing_ell(f, qs[k].ell_coeffs[idx[k]++], ps[k].p);
    }
    return f12conj(f);  // X_IS_NEGATIVE
}
inline Fp12Var final_exponentiation(const Fp12Var& f) {
    Fp12Var f1 = f12conj(f);
    Fp12Var f2 = f12inv(f);
    Fp12Var r = f12mul(f1, f2);
    f2 = r;
    r = f12frobenius(r, 2);
    r = f12mul(r, f2);
    Fp12Var y0 = f12cyclotomic_square(r);
    y0 = f12conj(y0);
    Fp12Var y5 = f12exp_by_x(r);
    Fp12Var y1 = f12cyclotomic_square(y5);
    Fp12Var y3 = f12mul(y0, y5);
    y0 = f12exp_by_x(y3);
    Fp12Var y2 = f12exp_by_x(y0);
    Fp12Var y4 = f12exp_by_x(y2);
    y4 = f12mul(y4, y1);
    y1 = f12exp_by_x(y4);
    y3 = f12conj(y3);
    y1 = f12mul(y1, y3);
    y1 = f12mul(y1, r);
    y3 = f12conj(r);
    y0 = f12mul(y0, r);
    y0 = f12frobenius(y0, 3);
    y4 = f12mul(y4, y3);
    y4 = f12frobenius(y4, 1);
    y5 = f12mul(y5, y2);
    y5 = f12frobenius(y5, 2);
    y5 = f12mul(y5, y0);
    y5 = f12mul(y5, y4);
    y5 = f12mul(y5, y1);
    return y5;
}

// ------------------------------------------------------------------ constraints.rs
struct VerifyTrace {
    HashTrace hash;
    Fp12 f_miller, f_final;
    bool result;
};
// constraints.rs:90-128  (parameters.g1_generator is a Constant, as in constraints.rs:347-352)
inline Bool bls_verify_gadget(const G1Var& g1_generator, const G1Var& pk, const std::vector<U8>& message, const G2Var& sig, VerifyTrace* tr = nullptr) {
    CSREF.mark("verify.pk_not_zero");
    pv_enforce_not_equal<FpT>(pk, pv_zero<FpT>());
    G1Var g1_neg = pv_negate<FpT>(g1_generator);
    G2Var h = hash_to_g2_with_cons(message, tr ? &tr->hash : nullptr);
    CSREF.mark("prepare.g1_neg");
    G1Prepared g1_neg_prepared = g1_prepare(g1_neg);
    CSREF.mark("prepare.h");
    G2Prepared h_prepared = g2_prepare(h);
    CSREF.mark("prepare.pk");
    G1Prepared pk_prepared = g1_prepare(pk);
    CSREF.mark("prepare.sig");
    G2Prepared sig_prepared = g2_prepare(sig);
    CSREF.mark("miller");
    Fp12Var ml = miller_loop({g1_neg_prepared, pk_prepared}, {sig_prepared, h_prepared});
    CSREF.mark("final_exp");
    Fp12Var fe = final_exponentiation(ml);
    CSREF.mark("is_one");
    Bool res = f12is_eq(fe, f12one());
    CSREF.mark("end");
    if (tr) {
        tr->f_miller = ml.val();
        tr->f_final = fe.val();
        tr->result = res.val;
    }
    return res;
}
// the circuit of constraints.rs:335-366: msg witness bytes, params Constant, pk Witness, sig Witness, verify.
// params_witness: ParametersVar::new_variable with AllocationMode::Witness instead (constraints.rs:198-211 takes any mode): the
// generator goes through G1Var::new_variable like a public key, between the message and the key (argument order of :346-364).
// pk_input / sig_input: PublicKeyVar / SignatureVar::new_variable with AllocationMode::Input (constraints.rs:214-249 take any mode): the point's
// x, y, z are public inputs (instance_assignment = [1, pk.x, pk.y, pk.z, sig.x.c0, .., sig.z.c1] in allocation order) and the allocation
// segment has no witnesses (pv_new_input: no prime-order check). The caller sets CSREF.n_inst = 1 + 3 pk_input + 6 sig_input first.
inline Bool bls_verify_circuit(const G1Aff& pk, const uint8_t* msg, size_t msg_len, const G2Aff& sig, VerifyTrace* tr = nullptr,
                               bool params_witness = false, bool pk_input = false, bool sig_input = false) {
    CSREF.mark("msg");
    std::vector<U8> msg_var = u8witness_vec(msg, msg_len);
    if (params_witness) CSREF.mark("params_alloc");
    G1Var g1 = params_witness ? g1_new_witness(g1_generator()) : pv_constant<FpT>(g1_generator());
    CSREF.mark("pk_alloc");
    G1Var pk_var = pk_input ? pv_new_input<FpT>(pk) : g1_new_witness(pk);
    CSREF.mark("sig_alloc");
    G2Var sig_var = sig_input ? pv_new_input<Fp2T>(sig) : g2_new_witness(sig);
    return bls_verify_gadget(g1, pk_var, msg_var, sig_var, tr);
}


// ------------------------------------------------------------------ N+1-pair product (BASELINE configs[3])
// One signature over K (pk_j, msg_j) pairs: e(-g1, sig) * prod_j e(pk_j, H(msg_j)) == 1. The reference has no such
// function; this is constraints.rs:90-128 with every per-key statement turned into a loop over the pairs, in the same
// statement order, and product_of_pairings called on slices of K + 1 prepared points (constraints.rs:121-125 passes
// slices of 2; miller_loop above already takes slices). K == 1 is bls_verify_gadget statement for statement.
inline Bool bls_verify_multi_gadget(const G1Var& g1_generator, const std::vector<G1Var>& pks, const std::vector<std::vector<U8>>& messages,
                                    const G2Var& sig) {
    assert(pks.size() == messages.size() && !pks.empty());
    CSREF.mark("verify.pk_not_zero");
    for (auto& pk : pks) pv_enforce_not_equal<FpT>(pk, pv_zero<FpT>());
    G1Var g1_neg = pv_negate<FpT>(g1_generator);
    std::vector<G2Var> hs;
    for (auto& m : messages) hs.push_back(hash_to_g2_with_cons(m));
    CSREF.mark("prepare.g1_neg");
    std::vector<G1Prepared> ps;
    std::vector<G2Prepared> qs(1);
    ps.push_back(g1_prepare(g1_neg));
    CSREF.mark("prepare.h");
    for (auto& h : hs) qs.push_back(g2_prepare(h));
    CSREF.mark("prepare.pk");
    for (auto& pk : pks) ps.push_back(g1_prepare(pk));
    CSREF.mark("prepare.sig");
    qs[0] = g2_prepare(sig);
    CSREF.mark("miller");
    Fp12Var ml = miller_loop(ps, qs);
    CSREF.mark("final_exp");
    Fp12Var fe = final_exponentiation(ml);
    CSREF.mark("is_one");
    Bool res = f12is_eq(fe, f12one());
    CSREF.mark("end");
    return res;
}
// allocation order of constraints.rs:335-366 with K messages and K keys: msgs, params Constant, pks Witness, sig Witness
inline Bool bls_verify_multi_circuit(const std::vector<G1Aff>& pks, const uint8_t* msgs, size_t msg_len, const G2Aff& sig) {
    CSREF.mark("msg");
    std::vector<std::vector<U8>> msg_vars;
    for (size_t j = 0; j < pks.size(); j++) msg_vars.push_back(u8witness_vec(msgs + j * msg_len, msg_len));
    G1Var g1 = pv_constant<FpT>(g1_generator());
    CSREF.mark("pk_alloc");
    std::vector<G1Var> pk_vars;
    for (auto& pk : pks) pk_vars.push_back(g1_new_witness(pk));
    CSREF.mark("sig_alloc");
    G2Var sig_var = g2_new_witness(sig);
    return bls_verify_multi_gadget(g1, pk_vars, msg_vars, sig_var);
}

// ------------------------------------------------------------------ aggregate_verify (constraints.rs:153-191)
inline U32 u32witness(uint32_t v) {
    U32 r;
    for (int i = 0; i < 32; i++) r.b[i] = balloc((v >> i) & 1);
    return r;
}
// Boolean::conditionally_select on a variable condition with constant arms (bits of count_one / count_zero)
inline Bool bselect_const_arms(const Bool& cond, bool t, bool f) {
    if (cond.kind == 0) return bconst(cond.val ? t : f);
    if (!f) return band(cond, bconst(t));            // (x, Constant(false)) => cond.and(x)
    if (!t) return band(bnot(cond), bconst(f));      // (Constant(false), x) => cond.not().and(x)
    return bconst(true);                             // both true
}
// mapped_aggregate (constraints.rs:169-191)
inline G1Var mapped_aggregate(const std::vector<G1Var>& keys, const std::vector<Bool>& bitmap, U32* count_out) {
    G1Var zero = pv_zero<FpT>();
    G1Var ret = zero;
    CSREF.mark("agg.count");
    U32 count = u32witness(0);
    CSREF.mark("agg.loop");
    for (size_t i = 0; i < keys.size(); i++) {
        G1Var sel = pv_select<FpT>(bitmap[i], keys[i], zero);
        ret = pv_add<FpT>(ret, sel);
        U32 inc;
        for (int b = 0; b < 32; b++) inc.b[b] = bselect_const_arms(bitmap[i], b == 0, false);
        count = u32addmany({count, inc});
    }
    if (count_out) *count_out = count;
    return ret;
}
// the circuit of constraints.rs:378-441: keys (Witness), bitmap booleans (Witness), msg bytes (Witness), params Constant,
// sig (Witness), then aggregate_verify
inline Bool bls_aggregate_verify_circuit(const std::vector<G1Aff>& pks, const std::vector<uint8_t>& bitmap, const uint8_t* msg, size_t msg_len,
                                         const G2Aff& sig, uint32_t* count_value, VerifyTrace* tr = nullptr) {
    assert(pks.size() == bitmap.size() && !pks.empty());
    CSREF.mark("agg.keys");
    std::vector<G1Var> keys;
    for (auto& pk : pks) keys.push_back(g1_new_witness(pk));
    CSREF.mark("agg.bitmap");
    std::vector<Bool> bits;
    for (uint8_t b : bitmap) bits.push_back(balloc(b != 0));
    CSREF.mark("msg");
    std::vector<U8> msg_var = u8witness_vec(msg, msg_len);
    G1Var g1 = pv_constant<FpT>(g1_generator());
    CSREF.mark("sig_alloc");
    G2Var sig_var = g2_new_witness(sig);
    U32 count;
    G1Var agg = mapped_aggregate(keys, bits, &count);
    if (count_value) *count_value = count.value();
    return bls_verify_gadget(g1, agg, msg_var, sig_var, tr);
}

}  // namespace orc
