// Witness-vector segment table (host side). Segment sizes are properties of the circuit shape:
// the SHA-256 segment is obtained by running the mask propagation of sha.hpp in count-only mode
// (no values, no stores); the field segments are the fixed constants below (pinned by tests against
// the CPU oracle's allocation trace).
#pragma once
#include <vector>
#include "../../include/blsw.h"
#include "sha.hpp"

namespace blsw {

enum : uint32_t {
    SEG_PK_ALLOC = 1942,     // 3 + 125 doublings * 11 + 47 additions * 12
    SEG_SIG_ALLOC = 12413,   // 6 + 254 * 30 + 132 * 36 + 35 (enforce_equal)
    SEG_PK_NOT_ZERO = 8,
    SEG_MAP = 5889,
    SEG_ADD = 36,
    SEG_COFACTOR = 8979,
    SEG_PREP_G2 = 1096,      // 18 + 63 * 16 + 5 * 14
    SEG_PREP_PK = 7,
    SEG_MILLER = 6826,       // 62 * 36 + 68 * (30 + 38) - 30
    SEG_FINAL_EXP = 7848,
    SEG_IS_ONE = 35,
};

// n_keys > 0: the aggregate_verify circuit (constraints.rs:378-441). n_pairs > 1: the N+1-pair product (one signature over
// n_pairs (pk, msg) pairs; every statement of constraints.rs:97-125 becomes a loop over the pairs, allocation order
// msgs, params, pks, sig as in constraints.rs:335-366). n_pairs == 1 is exactly the single-key circuit.
inline uint32_t seg_miller(uint32_t n_pairs) { return 62 * 36 + 68 * 30 - 30 + 68 * 38 * n_pairs; }
// params_witness: ParametersVar::new_variable(Witness) (constraints.rs:198-211), single-key circuit only: the generator's allocation
// segment follows the message (argument order of constraints.rs:346-364), prepare_g1(-g1) emits its to_affine in front of prepare(H),
// and every ell of the (-g1, sig) pair has a variable point: 38 witnesses instead of 30, 2 instead of 0 in the first one (f = 1).
// pk_input / sig_input: PublicKeyVar / SignatureVar::new_variable(Input) (constraints.rs:214-249), single-key circuit with Constant parameters only: the
// point's coordinates are instance variables and its allocation segment is empty (no in-circuit prime-order check for public inputs).
inline void make_layout(uint32_t msg_len, blsw_layout_t* L, uint32_t n_keys = 0, uint32_t n_pairs = 1, bool params_witness = false, bool pk_input = false,
                        bool sig_input = false) {
    std::vector<uint8_t> msg(msg_len ? msg_len : 1, 0);
    BitSink s;
    s.init(nullptr, 0);
    uint32_t uw[64];
    expand_message_w(s, msg.data(), msg_len, false, uw);
    L->msg_len = msg_len;
    L->n_instance_vars = 1 + (pk_input ? 3 : 0) + (sig_input ? 6 : 0);
    L->pk_mode = pk_input ? 1 : 0;
    L->sig_mode = sig_input ? 1 : 0;
    L->sha_bits = (uint32_t)s.nbits;
    uint32_t o = 0;
    L->n_keys = n_keys;
    L->off_keys = L->off_bitmap = L->off_count = L->off_agg = 0;
    if (n_keys) {  // keys, bitmap booleans, msg, sig, count, per-key (select 3 + add 12 (not the first) + addmany 33)
        L->off_keys = o;
        o += n_keys * SEG_PK_ALLOC;
        L->off_bitmap = o;
        o += n_keys;
    }
    const uint32_t K = n_keys ? 1 : (n_pairs ? n_pairs : 1);
    L->n_pairs = K;
    L->stride_msg = 8 * msg_len;
    L->stride_pk_alloc = pk_input ? 0 : SEG_PK_ALLOC;
    L->stride_pk_not_zero = SEG_PK_NOT_ZERO;
    L->stride_hash = L->sha_bits + 2 * SEG_MAP + SEG_ADD + SEG_COFACTOR;
    L->stride_prep_h = SEG_PREP_G2;
    L->stride_prep_pk = SEG_PREP_PK;
    L->off_msg = o;
    o += 8 * msg_len * K;
    L->params_mode = params_witness ? 1 : 0;
    L->off_params_alloc = L->off_prep_g1 = 0;
    if (params_witness) {
        L->off_params_alloc = o;
        o += SEG_PK_ALLOC;
    }
    L->off_pk_alloc = o;
    if (!n_keys) o += L->stride_pk_alloc * K;
    L->off_sig_alloc = o;
    o += sig_input ? 0 : SEG_SIG_ALLOC;
    if (n_keys) {
        L->off_count = o;
        o += 32;
        L->off_agg = o;
        o += 48 * n_keys - 12;
    }
    L->off_pk_not_zero = o;
    o += SEG_PK_NOT_ZERO * K;
    // hash_to_g2 of pair j: expand, map0, map1, add, cofactor, contiguous; pair j + 1 follows at + stride_hash
    L->off_expand = o;
    L->off_map0 = L->off_expand + L->sha_bits;
    L->off_map1 = L->off_map0 + SEG_MAP;
    L->off_add = L->off_map1 + SEG_MAP;
    L->off_cofactor = L->off_add + SEG_ADD;
    o += L->stride_hash * K;
    if (params_witness) {
        L->off_prep_g1 = o;
        o += SEG_PREP_PK;
    }
    L->off_prep_h = o;
    o += SEG_PREP_G2 * K;
    L->off_prep_pk = o;
    o += SEG_PREP_PK * K;
    L->off_prep_sig = o;
    o += SEG_PREP_G2;
    L->off_miller = o;
    o += seg_miller(K) + (params_witness ? 67 * 8 + 2 : 0);
    L->off_final_exp = o;
    o += SEG_FINAL_EXP;
    L->off_is_one = o;
    o += SEG_IS_ONE;
    L->n_witness = o;
}

}  // namespace blsw
