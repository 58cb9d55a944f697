// TEST HARNESS ONLY (never linked into libblsw.so): compiles the device headers of
// bls-verify-gadget_amd/csrc for the HOST with g++ and runs the witness chains for ONE instance on one
// "lane", so that kernel logic can be checked against the CPU oracle without a GPU (pytest -m "not gpu").
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../bls-verify-gadget_amd/csrc/chains.hpp"
#include "../../bls-verify-gadget_amd/csrc/decode.hpp"
#include "../../bls-verify-gadget_amd/csrc/layout.h"
#include "../../bls-verify-gadget_amd/csrc/team.hpp"
#include "../../bls-verify-gadget_amd/csrc/vsign.hpp"
#include "../../bls-verify-gadget_amd/csrc/miller_par.hpp"
#include "../../bls-verify-gadget_amd/csrc/cofactor_par.hpp"
#include "../../bls-verify-gadget_amd/csrc/cofactor_vf.hpp"
#include "../../bls-verify-gadget_amd/csrc/prepare_vf.hpp"
#include "../../bls-verify-gadget_amd/csrc/vpairing.hpp"
#include <array>

using namespace blsw;

static Fp load_fp(const uint64_t* p) {
    Fp r;
    memcpy(r.l, p, 48);
    return r;
}


// the six-lanes-per-instance program of team.hpp, lanes run one after the other per phase
struct TeamHost {
    typedef std::array<Fp2, 6> Reg;
    Fp2 slots[TS_NSLOTS];
    Emitter e;
    CoeffLinear coeff_sig, coeff_h;
    Reg zero() const {
        Reg r;
        r.fill(fp2_zero());
        return r;
    }
    Reg exec(const TeamOp& T, const Reg& a, const Reg& b) {
        for (uint32_t j = 0; j < 6; j++) {
            team_st(slots, TS_IN0 + j, a[j]);
            team_st(slots, TS_IN1 + j, b[j]);
        }
        for (uint32_t r = 0; r < T.rounds; r++)
            for (uint32_t j = 0; j < 6; j++) team_task(T.task[r][j], slots, e);
        Reg out;
        for (uint32_t j = 0; j < 6; j++) out[j] = team_gather(T.out[j], slots);
        e.pos += T.n_witness;
        return out;
    }
    Reg exec_hot(const TeamOp& T, const Reg& a, const Reg& b) { return exec(T, a, b); }
    Reg exp_by_x(const Reg& f) { return team_exp_by_x_body(*this, f); }
    Reg conj(const Reg& a) const {
        Reg r;
        for (uint32_t j = 0; j < 6; j++) r[j] = team_conj(j, a[j]);
        return r;
    }
    Reg frob(const Reg& a, int power) const {
        Reg r;
        for (uint32_t j = 0; j < 6; j++) r[j] = team_frob(j, a[j], power);
        return r;
    }
    void set_consts(const Fp& pkx, const Fp& pky) {
        for (uint32_t j = 0; j < 6; j++) team_set_consts_lane(j, slots, pkx, pky);
    }
    void load_coeffs(uint32_t k) {
        for (uint32_t j = 0; j < 6; j++) team_load_coeff_lane(j, slots, coeff_sig, coeff_h, k);
    }
    Reg first_f() const {
        Reg r;
        for (uint32_t j = 0; j < 6; j++) r[j] = team_first_f(j, slots);
        return r;
    }
    // native pairing (vpairing.hpp: team_miller_values)
    Reg one() const {
        Reg r = zero();
        r[0] = fp2_one();
        return r;
    }
    void load_lines(uint32_t k) {
        for (uint32_t j = 0; j < 6; j++) team_load_lines_lane(j, slots, coeff_sig, coeff_h, k);
    }
    // ParametersVar allocated as witnesses (team_miller_pv)
    Fp pkx, pky;
    void load_pair_sig(uint32_t k) {
        for (uint32_t j = 0; j < 6; j++) team_load_pair_lane(j, slots, coeff_sig, k, K_G1_GEN_X(), K_G1_GEN_NEG_Y());
    }
    void load_pair_h(uint32_t k) {
        for (uint32_t j = 0; j < 6; j++) team_load_pair_lane(j, slots, coeff_h, k, pkx, pky);
    }
    Reg first_f_var() {
        Reg r;
        for (uint32_t j = 0; j < 6; j++) r[j] = team_first_f_var(j, slots, e);
        e.pos += 2;
        return r;
    }
    // N+1-pair product: per-pair inputs
    const std::vector<std::vector<Fp>>* pair_coeff = nullptr;
    const std::vector<G1ChainOut>* pair_pk = nullptr;
    void load_coeff_sig(uint32_t k) {
        for (uint32_t j = 0; j < 6; j++) team_load_coeff_sig_lane(j, slots, coeff_sig, k);
    }
    void load_pair(uint32_t jp, uint32_t k) {
        CoeffLinear c{const_cast<Fp*>((*pair_coeff)[jp].data())};
        for (uint32_t j = 0; j < 6; j++) team_load_pair_lane(j, slots, c, k, (*pair_pk)[jp].ax, (*pair_pk)[jp].ay);
    }
    Reg inverse_w(const Reg& a) {
        for (uint32_t j = 0; j < 6; j++) team_st(slots, TS_IN0 + j, a[j]);
        team_inverse_lane0(slots);
        Reg inv;
        for (uint32_t j = 0; j < 6; j++) {
            inv[j] = team_ld(slots, TS_IN1 + j);
            Emitter w = e;
            w.pos += 2 * j;
            w.put(inv[j].c0);
            w.put(inv[j].c1);
        }
        e.pos += 12;
        exec(TEAM_OP_INVCHK, a, inv);
        return inv;
    }
    bool is_one_w(const Reg& a, const Emitter& e_one) {
        bool b[6];
        for (uint32_t j = 0; j < 6; j++) b[j] = team_is_one_coeff(j, a[j], e_one);
        bool res = false;
        for (uint32_t j = 0; j < 6; j++) res = team_is_one_tree(j, b, e_one);
        return res;
    }
};
static int g_use_team = 0;
static int g_cofactor_par = 0;  // 1: clear_cofactor2 through cofactor_par.hpp (chunks in the order 2, 0, 1, then the join); 2: cofactor_vf.hpp
static uint32_t g_miller_chunk = 2;  // pairs per chunk of the pair-parallel Miller product (mode 2)
// G2 allocation with the scalar multiplication of the subgroup check on the team program (lanes 0..2 own x, y, z)
static void g2_alloc_segment(uint32_t* base, const blsw_layout_t& L, const Fp2& sx, const Fp2& sy) {
    if (!g_use_team) {
        chain_g2_alloc({base, L.off_sig_alloc}, sx, sy);
        return;
    }
    constexpr uint32_t RM1[8] = BLSW_RM1_WORDS;
    bool inf = fp2_is_zero(sx) && fp2_is_zero(sy);
    TeamHost t;
    t.e = {base, L.off_sig_alloc};
    TeamHost::Reg ge = t.zero();
    ge[0] = inf ? fp2_zero() : sx;
    ge[1] = inf ? fp2_one() : sy;
    ge[2] = inf ? fp2_zero() : fp2_one();
    for (int j = 0; j < 3; j++) {
        t.e.put(ge[j].c0);
        t.e.put(ge[j].c1);
    }
    (void)team_g2_mul_bits(t, ge, RM1, BLSW_RM1_NBITS);
    chain_g2_alloc_tail(t.e, {ge[0], ge[1], ge[2]});
}

static bool pairing_segment(uint32_t* base, const blsw_layout_t& L, const Fp& ax, const Fp& ay, Fp* cs, Fp* ch) {
    if (L.params_mode) {  // six-lane program only (k_pairing_team_pv)
        TeamHost t;
        t.coeff_sig = CoeffLinear{cs};
        t.coeff_h = CoeffLinear{ch};
        t.e = {base, L.off_miller};
        t.pkx = ax;
        t.pky = ay;
        TeamHost::Reg f = team_miller_pv(t);
        if (t.e.pos != L.off_final_exp) return false;
        bool r = team_final_exp_is_one(t, f, Emitter{base, L.off_is_one});
        if (t.e.pos != L.off_is_one) return false;
        return r;
    }
    if (!g_use_team) {
        Fp12 fm = chain_miller({base, L.off_miller}, ax, ay, CoeffLinear{cs}, CoeffLinear{ch});
        return chain_final_exp_is_one({base, L.off_final_exp}, {base, L.off_is_one}, fm);
    }
    TeamHost t;
    t.coeff_sig = CoeffLinear{cs};
    t.coeff_h = CoeffLinear{ch};
    t.e = {base, L.off_miller};
    t.set_consts(ax, ay);
    TeamHost::Reg f = team_miller(t);
    if (t.e.pos != L.off_final_exp) return false;
    bool r = team_final_exp_is_one(t, f, Emitter{base, L.off_is_one});
    if (t.e.pos != L.off_is_one) return false;
    return r;
}

static int g_prepare_vf = 0;  // 1: prepare_g2 through prepare_vf.hpp (value chain, then the steps in reverse order)
static void run_prepare(Emitter e, const Proj<OpsFp2>& q, const CoeffLinear& out) {
    if (!g_prepare_vf) {
        chain_prepare_g2(e, q, out);
        return;
    }
    std::vector<Fp> scr(BLSW_PREPV_ELEMS);
    prepv_chain(e, q, CoeffLinear{scr.data()});
    for (int k = BLSW_PREPV_STEPS - 1; k >= 0; k--) prepv_step_w(e, (uint32_t)k, CoeffLinear{scr.data()}, out);
}
struct ParkHost {
    Jac2* p;
    void st(int slot, const Jac2& v) const { p[slot] = v; }
    Jac2 ld(int slot) const { return p[slot]; }
};
static Proj<OpsFp2> run_cofactor(Emitter e_add, Emitter e, const Proj<OpsFp2>& q0, const Proj<OpsFp2>& q1) {
    if (!g_cofactor_par) return chain_cofactor(e_add, e, q0, q1);
    Fp rows[BLSW_COFACTOR_ROWS];
    if (g_cofactor_par == 2) {  // cofactor_vf.hpp: values first, the parallel phases in reverse order of their lanes
        constexpr CofvPlan vp = cofv_plan();
        std::vector<Fp> scr(BLSW_COFV_ELEMS);
        const CoeffLinear S{scr.data()}, R{rows};
        constexpr CofvSeg sg = cofv_seg();
        // the forward doubling chain first (the device's main stream), then each chunk's pipeline (other streams), the chunks in an arbitrary order
        for (int s = 0; s < BLSW_COFV_NSEG; s++)
            cofv_chain_seg(s, e_add, e, [&](Proj<OpsFp2>& a, Proj<OpsFp2>& b) { a = q0; b = q1; }, S, R);
        const int chunk_order[3] = {2, 0, 1};
        for (int k = 0; k < 3; k++)
            for (int s = 0; s < BLSW_COFV_NSEG; s++) {
                if (sg.chunk[s] != chunk_order[k]) continue;
                cofv_bwd(s, S);
                const int hi = sg.bnd[s + 1] < BLSW_H_EFF_NBITS ? sg.bnd[s + 1] : BLSW_H_EFF_NBITS - 1;
                for (int D = hi; D > sg.bnd[s]; D--) cofv_affine((uint32_t)D, S);
                if (s == 0) cofv_affine(0, S);
                cofv_acc_seg(s, S, R);
            }
        for (int c = 0; c < 3; c++) cofv_acc_az(c, S);
        for (int D = BLSW_H_EFF_NBITS - 1; D >= 0; D--) cofv_dbl_w(e, (uint32_t)D, S);  // beside the addition chains on the device
        for (int c = 2; c >= 0; c--)
            for (int j = (int)vp.n_adds[c] - 1; j >= 0; j--) cofv_add_w(e, c, (uint32_t)j, S);
        return cofv_join(e, S, R);
    }
    const int order[3] = {2, 0, 1};  // the chunks do not depend on each other
    for (int k = 0; k < 3; k++) chain_cofactor_chunk(e_add, e, q0, q1, order[k], CoeffLinear{rows});
    return chain_cofactor_join(e, CoeffLinear{rows});
}
extern "C" {
void hostsim_use_team(int on) { g_use_team = on; }
void hostsim_cofactor_par(int on) { g_cofactor_par = on; }
void hostsim_prepare_vf(int on) { g_prepare_vf = on; }
void hostsim_miller_chunk(uint32_t b) { g_miller_chunk = b ? b : 1; }
int hostsim_layout(uint32_t msg_len, blsw_layout_t* L) {
    make_layout(msg_len, L);
    return 0;
}
// out: n_witness * 6 u64. returns gadget result (0/1); seg_ends (optional, 16 u32): cursor after each chain
int hostsim_witness_params(const uint64_t* pk_xy, const uint8_t* msg, uint32_t msg_len, const uint64_t* sig_xy, uint32_t params_mode, uint64_t* out);
int hostsim_witness(const uint64_t* pk_xy, const uint8_t* msg, uint32_t msg_len, const uint64_t* sig_xy, uint64_t* out, uint32_t* seg_ends) {
    (void)seg_ends;
    return hostsim_witness_params(pk_xy, msg, msg_len, sig_xy, 0, out);
}
int hostsim_layout_params(uint32_t msg_len, uint32_t params_mode, blsw_layout_t* L) {
    make_layout(msg_len, L, 0, 1, params_mode == 1);
    return 0;
}
// params_mode 1: ParametersVar::new_variable(Witness) — the statements of k_g1's params lanes and k_pairing_team_pv
int hostsim_witness_params(const uint64_t* pk_xy, const uint8_t* msg, uint32_t msg_len, const uint64_t* sig_xy, uint32_t params_mode, uint64_t* out) {
    blsw_layout_t L;
    make_layout(msg_len, &L, 0, 1, params_mode == 1);
    uint32_t* base = reinterpret_cast<uint32_t*>(out);
    if (L.params_mode) {
        Emitter none = {nullptr, 0};
        Proj<OpsFp> g = chain_g1_alloc_only({base, L.off_params_alloc}, K_G1_GEN_X(), fp_neg(K_G1_GEN_NEG_Y()));
        g.y = fp_neg(g.y);
        G1ChainOut gn = chain_g1_post(none, {base, L.off_prep_g1}, g);
        if (memcmp(gn.ax.l, K_G1_GEN_X().l, 48) || memcmp(gn.ay.l, K_G1_GEN_NEG_Y().l, 48)) return -2;
    }
    // msg bits (UInt8::new_witness_vec)
    Emitter em = {base, L.off_msg};
    for (uint32_t i = 0; i < msg_len; i++)
        for (int j = 0; j < 8; j++) em.put_bool((msg[i] >> j) & 1);
    // G1 / G2 allocation
    Fp pkx = load_fp(pk_xy), pky = load_fp(pk_xy + 6);
    G1ChainOut g1 = chain_g1_alloc({base, L.off_pk_alloc}, {base, L.off_pk_not_zero}, {base, L.off_prep_pk}, pkx, pky);
    Fp2 sx = {load_fp(sig_xy), load_fp(sig_xy + 6)}, sy = {load_fp(sig_xy + 12), load_fp(sig_xy + 18)};
    g2_alloc_segment(base, L, sx, sy);
    // expand_message: bitstream then expansion
    std::vector<uint32_t> bits((L.sha_bits + 31) / 32 + 1, 0);
    BitSink s;
    s.init(bits.data(), 1);
    uint32_t uw[64];
    expand_message_w(s, msg, msg_len, false, uw);
    if (s.nbits != L.sha_bits) return -1;
    Emitter ex = {base, L.off_expand};
    for (uint32_t i = 0; i < L.sha_bits; i++) ex.put_bool((bits[i >> 5] >> (i & 31)) & 1);
    Fp2 u0 = {hash_to_field_elem(uw), hash_to_field_elem(uw + 16)};
    Fp2 u1 = {hash_to_field_elem(uw + 32), hash_to_field_elem(uw + 48)};
    Proj<OpsFp2> q0 = chain_map_to_curve({base, L.off_map0}, u0);
    Proj<OpsFp2> q1 = chain_map_to_curve({base, L.off_map1}, u1);
    Proj<OpsFp2> h = run_cofactor({base, L.off_add}, {base, L.off_cofactor}, q0, q1);
    std::vector<Fp> ch(68 * 4), cs(68 * 4);
    run_prepare({base, L.off_prep_h}, h, CoeffLinear{ch.data()});
    bool sinf = fp2_is_zero(sx) && fp2_is_zero(sy);
    Proj<OpsFp2> sp = {sinf ? fp2_zero() : sx, sinf ? fp2_one() : sy, sinf ? fp2_zero() : fp2_one()};
    run_prepare({base, L.off_prep_sig}, sp, CoeffLinear{cs.data()});
    bool res = pairing_segment(base, L, g1.ax, g1.ay, cs.data(), ch.data());
    return res ? 1 : 0;
}
// PublicKeyVar / SignatureVar allocated as public inputs (constraints.rs:214-249 with AllocationMode::Input): the statements of k_g1's / k_prepare's
// Input paths. out_instance: [n_instance_vars][6] = instance_assignment (element 0 = one)
int hostsim_witness_io(const uint64_t* pk_xy, const uint8_t* msg, uint32_t msg_len, const uint64_t* sig_xy, uint32_t pk_mode, uint32_t sig_mode, uint64_t* out,
                       uint64_t* out_instance) {
    blsw_layout_t L;
    make_layout(msg_len, &L, 0, 1, false, pk_mode == 1, sig_mode == 1);
    uint32_t* base = reinterpret_cast<uint32_t*>(out);
    Fp* inst = reinterpret_cast<Fp*>(out_instance);
    inst[0] = fp_one();
    Emitter em = {base, L.off_msg};
    for (uint32_t i = 0; i < msg_len; i++)
        for (int j = 0; j < 8; j++) em.put_bool((msg[i] >> j) & 1);
    Fp pkx = load_fp(pk_xy), pky = load_fp(pk_xy + 6);
    Proj<OpsFp> pk;
    if (pk_mode) {
        const bool inf = fp_is_zero(pkx) && fp_is_zero(pky);
        pk = {inf ? fp_zero() : pkx, inf ? fp_one() : pky, inf ? fp_zero() : fp_one()};
        inst[1] = pk.x;
        inst[2] = pk.y;
        inst[3] = pk.z;
    } else {
        pk = chain_g1_alloc_only({base, L.off_pk_alloc}, pkx, pky);
    }
    G1ChainOut g1 = chain_g1_post({base, L.off_pk_not_zero}, {base, L.off_prep_pk}, pk);
    Fp2 sx = {load_fp(sig_xy), load_fp(sig_xy + 6)}, sy = {load_fp(sig_xy + 12), load_fp(sig_xy + 18)};
    bool sinf = fp2_is_zero(sx) && fp2_is_zero(sy);
    Proj<OpsFp2> sp = {sinf ? fp2_zero() : sx, sinf ? fp2_one() : sy, sinf ? fp2_zero() : fp2_one()};
    if (sig_mode) {
        const uint32_t k0 = 1 + (pk_mode ? 3 : 0);
        const Fp v[6] = {sp.x.c0, sp.x.c1, sp.y.c0, sp.y.c1, sp.z.c0, sp.z.c1};
        for (int k = 0; k < 6; k++) inst[k0 + k] = v[k];
    } else {
        g2_alloc_segment(base, L, sx, sy);
    }
    std::vector<uint32_t> bits((L.sha_bits + 31) / 32 + 1, 0);
    BitSink s;
    s.init(bits.data(), 1);
    uint32_t uw[64];
    expand_message_w(s, msg, msg_len, false, uw);
    if (s.nbits != L.sha_bits) return -1;
    Emitter ex = {base, L.off_expand};
    for (uint32_t i = 0; i < L.sha_bits; i++) ex.put_bool((bits[i >> 5] >> (i & 31)) & 1);
    Fp2 u0 = {hash_to_field_elem(uw), hash_to_field_elem(uw + 16)};
    Fp2 u1 = {hash_to_field_elem(uw + 32), hash_to_field_elem(uw + 48)};
    Proj<OpsFp2> q0 = chain_map_to_curve({base, L.off_map0}, u0);
    Proj<OpsFp2> q1 = chain_map_to_curve({base, L.off_map1}, u1);
    Proj<OpsFp2> h = run_cofactor({base, L.off_add}, {base, L.off_cofactor}, q0, q1);
    std::vector<Fp> ch(68 * 4), cs(68 * 4);
    run_prepare({base, L.off_prep_h}, h, CoeffLinear{ch.data()});
    run_prepare({base, L.off_prep_sig}, sp, CoeffLinear{cs.data()});
    bool res = pairing_segment(base, L, g1.ax, g1.ay, cs.data(), ch.data());
    return res ? 1 : 0;
}
int hostsim_layout_io(uint32_t msg_len, uint32_t pk_mode, uint32_t sig_mode, blsw_layout_t* L) {
    make_layout(msg_len, L, 0, 1, false, pk_mode == 1, sig_mode == 1);
    return 0;
}
// N+1-pair product circuit for one instance (blsw_verify_multi_batch): pks_xy [K][12], msgs [K][msg_len]
struct HostPairs {
    const std::vector<std::vector<Fp>>* coeff;
    const std::vector<G1ChainOut>* keys;
    void pk(uint32_t j, Fp& x, Fp& y) const {
        x = (*keys)[j].ax;
        y = (*keys)[j].ay;
    }
    CoeffLinear coeff_h(uint32_t j) const { return CoeffLinear{const_cast<Fp*>((*coeff)[j].data())}; }
};
int hostsim_witness_multi(const uint64_t* pks_xy, const uint8_t* msgs, uint32_t msg_len, uint32_t K, const uint64_t* sig_xy, uint64_t* out, blsw_layout_t* Lout) {
    blsw_layout_t L;
    make_layout(msg_len, &L, 0, K);
    if (Lout) *Lout = L;
    if (!out) return 0;
    uint32_t* base = reinterpret_cast<uint32_t*>(out);
    std::vector<G1ChainOut> pk(K);
    std::vector<std::vector<Fp>> ch(K, std::vector<Fp>(68 * 4));
    for (uint32_t j = 0; j < K; j++) {
        const uint8_t* msg = msgs + (size_t)j * msg_len;
        Emitter em = {base, L.off_msg + j * L.stride_msg};
        for (uint32_t i = 0; i < msg_len; i++)
            for (int b = 0; b < 8; b++) em.put_bool((msg[i] >> b) & 1);
        pk[j] = chain_g1_alloc({base, L.off_pk_alloc + j * L.stride_pk_alloc}, {base, L.off_pk_not_zero + j * L.stride_pk_not_zero},
                               {base, L.off_prep_pk + j * L.stride_prep_pk}, load_fp(pks_xy + 12 * j), load_fp(pks_xy + 12 * j + 6));
        std::vector<uint32_t> bits((L.sha_bits + 31) / 32 + 1, 0);
        BitSink s;
        s.init(bits.data(), 1);
        uint32_t uw[64];
        expand_message_w(s, msg, msg_len, false, uw);
        if (s.nbits != L.sha_bits) return -1;
        const uint32_t ho = j * L.stride_hash;
        Emitter ex = {base, L.off_expand + ho};
        for (uint32_t i = 0; i < L.sha_bits; i++) ex.put_bool((bits[i >> 5] >> (i & 31)) & 1);
        Fp2 u0 = {hash_to_field_elem(uw), hash_to_field_elem(uw + 16)};
        Fp2 u1 = {hash_to_field_elem(uw + 32), hash_to_field_elem(uw + 48)};
        Proj<OpsFp2> q0 = chain_map_to_curve({base, L.off_map0 + ho}, u0);
        Proj<OpsFp2> q1 = chain_map_to_curve({base, L.off_map1 + ho}, u1);
        Proj<OpsFp2> h = run_cofactor({base, L.off_add + ho}, {base, L.off_cofactor + ho}, q0, q1);
        run_prepare({base, L.off_prep_h + j * L.stride_prep_h}, h, CoeffLinear{ch[j].data()});
    }
    Fp2 sx = {load_fp(sig_xy), load_fp(sig_xy + 6)}, sy = {load_fp(sig_xy + 12), load_fp(sig_xy + 18)};
    g2_alloc_segment(base, L, sx, sy);
    std::vector<Fp> cs(68 * 4);
    bool sinf = fp2_is_zero(sx) && fp2_is_zero(sy);
    Proj<OpsFp2> sp = {sinf ? fp2_zero() : sx, sinf ? fp2_one() : sy, sinf ? fp2_zero() : fp2_one()};
    run_prepare({base, L.off_prep_sig}, sp, CoeffLinear{cs.data()});
    if (g_use_team == 2) {  // miller_par.hpp: the four phases of the pair-parallel Miller product, tasks run one after the other
        HostPairs hp = {&ch, &pk};
        const uint32_t B = g_miller_chunk, C = miller_chunks(K, B), S = BLSW_MILLER_STEPS;
        std::vector<Fp> cprod(12 * S * C), q(12 * S * C), tt(12 * S), f1(12 * S);
        Fp12Rows Cp = {cprod.data(), (uint64_t)S * C}, Q = {q.data(), (uint64_t)S * C}, T = {tt.data(), S}, F1 = {f1.data(), S};
        for (uint32_t k = 0; k < S; k++)
            for (uint32_t c = 0; c < C; c++) Cp.st((uint64_t)k * C + c, miller_m1(hp, K, B, k, c));
        for (uint32_t k = 0; k < S; k++) miller_m1b(Cp, Q, T, (uint64_t)k * C, k, C);
        Fp12 fm = miller_m2(Emitter{base, L.off_miller}, K, CoeffLinear{cs.data()}, T, F1, 0);
        for (uint32_t k = 0; k < S; k++)
            for (uint32_t c = 0; c < C; c++) miller_m3(Emitter{base, L.off_miller}, hp, K, B, k, c, F1.ld(k), Q, (uint64_t)k * C + c);
        return chain_final_exp_is_one({base, L.off_final_exp}, {base, L.off_is_one}, fm) ? 1 : 0;
    }
    if (!g_use_team) {
        HostPairs hp = {&ch, &pk};
        Fp12 fm = chain_miller_multi({base, L.off_miller}, K, hp, CoeffLinear{cs.data()});
        return chain_final_exp_is_one({base, L.off_final_exp}, {base, L.off_is_one}, fm) ? 1 : 0;
    }
    TeamHost t;
    t.coeff_sig = CoeffLinear{cs.data()};
    t.coeff_h = CoeffLinear{nullptr};
    t.pair_coeff = &ch;
    t.pair_pk = &pk;
    t.e = {base, L.off_miller};
    team_st(t.slots, TS_XYC, {K_G1_GEN_NEG_Y(), fp_zero()});
    TeamHost::Reg f = team_miller_multi(t, K);
    if (t.e.pos != L.off_final_exp) return -2;
    bool r = team_final_exp_is_one(t, f, Emitter{base, L.off_is_one});
    if (t.e.pos != L.off_is_one) return -3;
    return r ? 1 : 0;
}
struct HostKeys {
    const std::vector<Proj<OpsFp>>* v;
    Proj<OpsFp> ld(uint32_t k) const { return (*v)[k]; }
};
// aggregate_verify circuit for one instance: pks_xy [K][12], bitmap [K]
int hostsim_witness_aggregate(const uint64_t* pks_xy, const uint8_t* bitmap, uint32_t K, const uint8_t* msg, uint32_t msg_len, const uint64_t* sig_xy,
                              uint64_t* out, uint32_t* count_out, blsw_layout_t* Lout) {
    blsw_layout_t L;
    make_layout(msg_len, &L, K);
    if (Lout) *Lout = L;
    if (!out) return 0;
    uint32_t* base = reinterpret_cast<uint32_t*>(out);
    std::vector<Proj<OpsFp>> keys;
    for (uint32_t k = 0; k < K; k++)
        keys.push_back(chain_g1_alloc_only({base, L.off_keys + k * SEG_PK_ALLOC}, load_fp(pks_xy + 12 * k), load_fp(pks_xy + 12 * k + 6)));
    Emitter eb = {base, L.off_bitmap};
    for (uint32_t k = 0; k < K; k++) eb.put_bool(bitmap[k] != 0);
    Emitter em = {base, L.off_msg};
    for (uint32_t i = 0; i < msg_len; i++)
        for (int j = 0; j < 8; j++) em.put_bool((msg[i] >> j) & 1);
    Fp2 sx = {load_fp(sig_xy), load_fp(sig_xy + 6)}, sy = {load_fp(sig_xy + 12), load_fp(sig_xy + 18)};
    g2_alloc_segment(base, L, sx, sy);
    HostKeys hk = {&keys};
    Proj<OpsFp> agg = chain_mapped_aggregate({base, L.off_count}, {base, L.off_agg}, hk, bitmap, K, count_out);
    G1ChainOut g1 = chain_g1_post({base, L.off_pk_not_zero}, {base, L.off_prep_pk}, agg);
    std::vector<uint32_t> bits((L.sha_bits + 31) / 32 + 1, 0);
    BitSink s;
    s.init(bits.data(), 1);
    uint32_t uw[64];
    expand_message_w(s, msg, msg_len, false, uw);
    Emitter ex = {base, L.off_expand};
    for (uint32_t i = 0; i < L.sha_bits; i++) ex.put_bool((bits[i >> 5] >> (i & 31)) & 1);
    Fp2 u0 = {hash_to_field_elem(uw), hash_to_field_elem(uw + 16)};
    Fp2 u1 = {hash_to_field_elem(uw + 32), hash_to_field_elem(uw + 48)};
    Proj<OpsFp2> q0 = chain_map_to_curve({base, L.off_map0}, u0);
    Proj<OpsFp2> q1 = chain_map_to_curve({base, L.off_map1}, u1);
    Proj<OpsFp2> hh = run_cofactor({base, L.off_add}, {base, L.off_cofactor}, q0, q1);
    std::vector<Fp> ch(68 * 4), cs(68 * 4);
    chain_prepare_g2({base, L.off_prep_h}, hh, CoeffLinear{ch.data()});
    bool sinf = fp2_is_zero(sx) && fp2_is_zero(sy);
    Proj<OpsFp2> sp = {sinf ? fp2_zero() : sx, sinf ? fp2_one() : sy, sinf ? fp2_zero() : fp2_one()};
    run_prepare({base, L.off_prep_sig}, sp, CoeffLinear{cs.data()});
    Fp12 fm = chain_miller({base, L.off_miller}, g1.ax, g1.ay, CoeffLinear{cs.data()}, CoeffLinear{ch.data()});
    bool res = chain_final_exp_is_one({base, L.off_final_exp}, {base, L.off_is_one}, fm);
    return res ? 1 : 0;
}
int hostsim_g1_decode(const uint8_t* in, uint64_t* out_xy) {
    Fp x, y;
    int st = g1_decode(in, x, y);
    memcpy(out_xy, x.l, 48);
    memcpy(out_xy + 6, y.l, 48);
    return st;
}
int hostsim_g2_decode(const uint8_t* in, uint64_t* out_xy) {
    Fp2 x, y;
    int st = g2_decode(in, x, y);
    memcpy(out_xy, x.c0.l, 48);
    memcpy(out_xy + 6, x.c1.l, 48);
    memcpy(out_xy + 12, y.c0.l, 48);
    memcpy(out_xy + 18, y.c1.l, 48);
    return st;
}
// signer logic (vsign.hpp, what k_sign runs per lane): sk (32 LE bytes), H(msg) affine -> sig96, pk48; returns the SIGN_* status
int hostsim_sign(const uint8_t* sk32, const uint64_t* h_xy, uint8_t* sig96, uint8_t* pk48) {
    uint32_t k[8];
    int st = sk_from_le32(sk32, k);
    Fp2 x = fp2_zero(), y = fp2_zero();
    Fp px = fp_zero(), py = fp_zero();
    bool sinf = true, pinf = true;
    if (st == SIGN_OK) {
        Fp2 hx = {load_fp(h_xy), load_fp(h_xy + 6)}, hy = {load_fp(h_xy + 12), load_fp(h_xy + 18)};
        Jac2 park[16];
        Jac2 acc = v_g2_mul_gls(ParkHost{park}, Jac2{hx, hy, fp2_one()}, k);
        if (!fp2_is_zero(acc.z)) {
            Fp2 zi = fp2_inv(acc.z), zi2 = fp2_sqr(zi);
            x = fp2_mul(acc.x, zi2);
            y = fp2_mul(acc.y, fp2_mul(zi2, zi));
            sinf = false;
        }
        Jac1v a1 = v1_mul_g1_fixed(k);
        if (!fp_is_zero(a1.z)) {
            Fp zi = fp_inv(a1.z), zi2 = fp_sqr(zi);
            px = fp_mul(a1.x, zi2);
            py = fp_mul(a1.y, fp_mul(zi2, zi));
            pinf = false;
        }
    }
    g2_encode(x, y, sinf, sig96);
    g1_encode(px, py, pinf, pk48);
    return st;
}
// R1CS evaluator (test side): checks <A_i, z> * <B_i, z> = <C_i, z> for every constraint with z = [1] ++ witness, matrices in
// the CSR form blsw_matrices_fill writes. Returns the index of the first unsatisfied constraint, or -1.
// z = [instance (n_inst elements, instance[0] = 1) | witness]: ark-relations' column order. instance == nullptr: z = [1 | witness]
int64_t hostsim_r1cs_check_io(uint64_t n_cons, const uint64_t* const* row_ptr, const uint32_t* const* col, const uint64_t* const* val, const uint64_t* witness,
                              uint64_t n_witness, const uint64_t* instance, uint64_t n_inst) {
    if (!instance) n_inst = 1;
    auto dot = [&](int m, uint64_t i) {
        Fp acc = fp_zero();
        for (uint64_t k = row_ptr[m][i]; k < row_ptr[m][i + 1]; k++) {
            const uint32_t c = col[m][k];
            if (c >= n_inst + n_witness) return fp_from_u32(0xdead);  // out of range: forces a mismatch
            Fp z = c == 0 ? fp_one() : (c < n_inst ? load_fp(instance + (uint64_t)c * 6) : load_fp(witness + (uint64_t)(c - n_inst) * 6));
            acc = fp_add(acc, fp_mul(load_fp(val[m] + k * 6), z));
        }
        return acc;
    };
    for (uint64_t i = 0; i < n_cons; i++)
        if (!fp_eq(fp_mul(dot(0, i), dot(1, i)), dot(2, i))) return (int64_t)i;
    return -1;
}
int64_t hostsim_r1cs_check(uint64_t n_cons, const uint64_t* const* row_ptr, const uint32_t* const* col, const uint64_t* const* val, const uint64_t* witness,
                           uint64_t n_witness) {
    return hostsim_r1cs_check_io(n_cons, row_ptr, col, val, witness, n_witness, nullptr, 1);
}
// value-only hash_to_g2 (vcurve.hpp: what blsw_hash_to_g2_batch / blsw_sign_batch run per lane): expand_message values,
// hash_to_field, SSWU + isogeny x 2, Q0 + Q1, psi-based cofactor clearing; affine result (all zero = identity)
void hostsim_hash_to_g2_values(const uint8_t* msg, uint32_t msg_len, uint64_t* out_xy) {
    uint32_t uw[64];
    expand_message_values(msg, msg_len, uw);
    Fp2 u0 = {hash_to_field_elem(uw), hash_to_field_elem(uw + 16)}, u1 = {hash_to_field_elem(uw + 32), hash_to_field_elem(uw + 48)};
    Proj<OpsFp2> q0 = v_map_to_curve(u0), q1 = v_map_to_curve(u1);
    Jac2 r = {q0.x, q0.y, q0.z};
    if (fp2_is_zero(q0.z)) r = {fp2_one(), fp2_one(), fp2_zero()};
    if (!fp2_is_zero(q1.z)) r = v_add_mixed(r, q1.x, q1.y);
    Jac2 park[3];
    Jac2 acc = v_clear_cofactor(ParkHost{park}, r);
    memset(out_xy, 0, 24 * 8);
    if (fp2_is_zero(acc.z)) return;
    Fp2 zi = fp2_inv(acc.z), zi2 = fp2_sqr(zi);
    Fp2 x = fp2_mul(acc.x, zi2), y = fp2_mul(acc.y, fp2_mul(zi2, zi));
    memcpy(out_xy, x.c0.l, 48);
    memcpy(out_xy + 6, x.c1.l, 48);
    memcpy(out_xy + 12, y.c0.l, 48);
    memcpy(out_xy + 18, y.c1.l, 48);
}
// blsw_verify_batch's per-instance logic (k_decode, the value-only hash, k_vlines, k_verify_team): verdict from compressed bytes; st[2] = decode statuses
int hostsim_verify_values(const uint8_t* pk48, const uint8_t* sig96, const uint8_t* msg, uint32_t msg_len, int32_t* st) {
    Fp px, py;
    Fp2 sx, sy;
    st[0] = g1_decode(pk48, px, py);
    st[1] = g2_decode(sig96, sx, sy);
    uint64_t h_xy[24];
    hostsim_hash_to_g2_values(msg, msg_len, h_xy);
    const Fp2 hx = {load_fp(h_xy), load_fp(h_xy + 6)}, hy = {load_fp(h_xy + 12), load_fp(h_xy + 18)};
    std::vector<Fp> ls(BLSW_VLINE_ROWS), lh(BLSW_VLINE_ROWS);
    vline_chain(sx, sy, K_G1_GEN_X(), K_G1_GEN_NEG_Y(), CoeffLinear{ls.data()});
    vline_chain(hx, hy, px, py, CoeffLinear{lh.data()});
    TeamHost t;
    t.coeff_sig = CoeffLinear{ls.data()};
    t.coeff_h = CoeffLinear{lh.data()};
    t.e = {nullptr, 0};
    TeamHost::Reg f = team_miller_values(t);
    const bool one = team_final_exp_is_one(t, f, Emitter{nullptr, 0});
    return (one && st[0] == 0 && st[1] == 0) ? 1 : 0;
}
// clear_cofactor2 of a given pair (Q0, Q1) of affine points (z = 1; all zero = the identity (0, 0, 0)): the serial chain and the
// chunked one write their "add" + "cofactor" segments (36 + 8979 elements each) and results; returns 1 if everything is equal
int hostsim_cofactor_compare(const uint64_t* q0_xy, const uint64_t* q1_xy, uint64_t* out_serial, uint64_t* out_chunked) {
    auto load = [](const uint64_t* p) {
        Proj<OpsFp2> q = {{load_fp(p), load_fp(p + 6)}, {load_fp(p + 12), load_fp(p + 18)}, fp2_one()};
        if (fp2_is_zero(q.x) && fp2_is_zero(q.y)) q.z = fp2_zero();
        return q;
    };
    Proj<OpsFp2> q0 = load(q0_xy), q1 = load(q1_xy);
    Proj<OpsFp2> a = chain_cofactor({reinterpret_cast<uint32_t*>(out_serial), 0}, {reinterpret_cast<uint32_t*>(out_serial), SEG_ADD}, q0, q1);
    Fp rows[BLSW_COFACTOR_ROWS];
    Emitter e_add = {reinterpret_cast<uint32_t*>(out_chunked), 0}, e = {reinterpret_cast<uint32_t*>(out_chunked), SEG_ADD};
    Proj<OpsFp2> b;
    if (g_cofactor_par == 2) {
        b = run_cofactor(e_add, e, q0, q1);  // cofactor_vf.hpp
    } else {
        const int order[3] = {1, 2, 0};
        for (int k = 0; k < 3; k++) chain_cofactor_chunk(e_add, e, q0, q1, order[k], CoeffLinear{rows});
        b = chain_cofactor_join(e, CoeffLinear{rows});
    }
    bool same = fp_eq(a.x.c0, b.x.c0) && fp_eq(a.x.c1, b.x.c1) && fp_eq(a.y.c0, b.y.c0) && fp_eq(a.y.c1, b.y.c1) && fp_eq(a.z.c0, b.z.c0) && fp_eq(a.z.c1, b.z.c1);
    return same && memcmp(out_serial, out_chunked, (size_t)(SEG_ADD + SEG_COFACTOR) * 48) == 0;
}
// field micro-checks
void hostsim_fp_mul(const uint64_t* a, const uint64_t* b, uint64_t* r) {
    Fp z = fp_mul(load_fp(a), load_fp(b));
    memcpy(r, z.l, 48);
}
void hostsim_fp_mul32(const uint64_t* a, const uint64_t* b, uint64_t* r) {
    Fp z = fp_mul32(load_fp(a), load_fp(b));
    memcpy(r, z.l, 48);
}
void hostsim_fp_inv(const uint64_t* a, uint64_t* r) {
    Fp z = fp_inv(load_fp(a));
    memcpy(r, z.l, 48);
}
void hostsim_fp_inv_fermat(const uint64_t* a, uint64_t* r) {
    Fp z = fp_inv_fermat(load_fp(a));
    memcpy(r, z.l, 48);
}
}
