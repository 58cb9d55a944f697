// C++ host side above the C ABI of blsw.h, with the names of the reference's gadget interface (src/constraints.rs, src/bls.rs) so that a
// caller — and the parity tests — read like the reference's own tests (constraints.rs:318-376). Header only, C++17, HIP runtime API for the
// device buffers; no torch. The reference is a Rust crate: its maintainers bind blsw.h directly; this header is the same
// boundary for a C++ host, and the statement-by-statement mirror of what the Rust side does around `verify`.
//
// One difference in shape, stated once: the reference builds ONE ConstraintSystemRef per (pk, msg, sig) instance and synthesises it on a CPU
// thread; here a ConstraintSystem stands for n independent systems of one shape, generated together on the GPU. Everything else keeps its name:
//   PublicKey::try_from / Signature::try_from          bls.rs:219-242, 316-339 (compressed ZCash encoding; decoded on the device)
//   UInt8::new_witness_vec                              constraints.rs:341
//   ParametersVar / PublicKeyVar / SignatureVar::new_variable(cs, value, AllocationMode)   constraints.rs:194-249
//   BlsSignatureVerifyGadget::verify(&params, &pk, &msg, &sig) -> Boolean                  constraints.rs:90-128
//   BlsSignatureVerifyGadget::aggregate_verify(&params, &keys, &bitmap, &msg, &sig) -> (Boolean, UInt32)   constraints.rs:153-191
//   cs.num_constraints(), cs.num_witness_variables(), Boolean::value()                     constraints.rs:369-373
//   BLS::sign / BLS::verify (the native scheme, batched)                                   bls.rs:411-458
#pragma once
#include <hip/hip_runtime_api.h>

#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "blsw.h"

namespace blsw {

class Error : public std::runtime_error {
   public:
    int code;
    Error(const std::string& what, int c) : std::runtime_error(what + " failed: " + std::to_string(c)), code(c) {}
};
inline void check(int rc, const char* what) {
    if (rc != BLSW_OK) throw Error(what, rc);
}
inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(std::string(what) + " (" + hipGetErrorString(e) + ")", BLSW_ERR_HIP);
}

namespace detail {
class DeviceBytes {  // RAII device allocation
   public:
    DeviceBytes() = default;
    explicit DeviceBytes(size_t bytes) : n_(bytes) { hip_check(hipMalloc(&p_, bytes ? bytes : 1), "hipMalloc"); }
    DeviceBytes(const DeviceBytes&) = delete;
    DeviceBytes& operator=(const DeviceBytes&) = delete;
    DeviceBytes(DeviceBytes&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
    DeviceBytes& operator=(DeviceBytes&& o) noexcept {
        if (this != &o) {
            if (p_) (void)hipFree(p_);
            p_ = o.p_;
            n_ = o.n_;
            o.p_ = nullptr;
        }
        return *this;
    }
    ~DeviceBytes() {
        if (p_) (void)hipFree(p_);
    }
    void* get() const { return p_; }
    size_t size() const { return n_; }
    void upload(const void* src, size_t bytes) { hip_check(hipMemcpy(p_, src, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
    void download(void* dst, size_t bytes, size_t offset = 0) const {
        hip_check(hipMemcpy(dst, static_cast<const char*>(p_) + offset, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    }

   private:
    void* p_ = nullptr;
    size_t n_ = 0;
};
inline std::vector<uint8_t> unhex(std::string s, size_t want) {
    if (s.rfind("0x", 0) == 0) s = s.substr(2);
    if (s.size() != 2 * want) throw Error("hex string of " + std::to_string(want) + " bytes", BLSW_ERR_ARG);
    std::vector<uint8_t> out(want);
    for (size_t i = 0; i < want; i++) {
        auto nib = [&](char c) -> int {
            if (c >= '0' && c <= '9') return c - '0';
            if (c >= 'a' && c <= 'f') return c - 'a' + 10;
            if (c >= 'A' && c <= 'F') return c - 'A' + 10;
            throw Error("hex digit", BLSW_ERR_ARG);
        };
        out[i] = (uint8_t)(nib(s[2 * i]) * 16 + nib(s[2 * i + 1]));
    }
    return out;
}
}  // namespace detail

// ark_r1cs_std::alloc::AllocationMode. PublicKeyVar / SignatureVar: Witness (the reference's circuits) or Input (the point's coordinates become
// instance_assignment[1..]); ParametersVar: Constant or Witness.
enum class AllocationMode { Constant, Input, Witness };

// bls.rs:23-38: Parameters::default() is the standard G1 generator — the only value the engine knows
struct Parameters {};
// bls.rs:219-242 / 316-339: the compressed encodings. try_from checks the text form here; flags, x < p, curve and subgroup membership are
// checked by the device decode when the variable is used, with the outcome in ConstraintSystem::status() (the reference's test harness
// replaces a point that does not decode by the default and expects `false`: tests/tests.rs:244-263 — that rule is applied by verify)
struct PublicKey {
    std::array<uint8_t, 48> bytes;
    static PublicKey try_from(const std::string& hex) {
        PublicKey k;
        auto b = detail::unhex(hex, 48);
        std::memcpy(k.bytes.data(), b.data(), 48);
        return k;
    }
};
struct Signature {
    std::array<uint8_t, 96> bytes;
    static Signature try_from(const std::string& hex) {
        Signature s;
        auto b = detail::unhex(hex, 96);
        std::memcpy(s.bytes.data(), b.data(), 96);
        return s;
    }
};

// n independent constraint systems of one circuit shape (constraints.rs:335: `ConstraintSystem::<Fq>::new_ref()`, once per instance there)
class ConstraintSystem {
   public:
    ConstraintSystem(size_t n, uint32_t msg_len, int device = -1) : n_(n), msg_len_(msg_len), device_(device) {
        if (n == 0) throw Error("ConstraintSystem(n = 0)", BLSW_ERR_ARG);
        check(blsw_layout(msg_len, &layout_), "blsw_layout");
    }
    ~ConstraintSystem() {
        if (engine_) blsw_engine_destroy(engine_);
    }
    ConstraintSystem(const ConstraintSystem&) = delete;
    ConstraintSystem& operator=(const ConstraintSystem&) = delete;

    size_t num_instances() const { return n_; }
    uint32_t msg_len() const { return msg_len_; }
    // cs.num_witness_variables() / num_instance_variables() of every one of the n systems (shape properties; after ParametersVar::new_variable
    // they describe the circuit with that parameter mode)
    uint64_t num_witness_variables() const { return layout_.n_witness; }
    uint64_t num_instance_variables() const { return layout_.n_instance_vars; }
    // cs.num_constraints(): the library synthesises the system symbolically on the host (a few seconds, once per call)
    uint64_t num_constraints() const {
        blsw_matrices_info_t info;
        check(layout_.pk_mode || layout_.sig_mode ? blsw_matrices_info_io(msg_len_, layout_.pk_mode, layout_.sig_mode, &info)
              : layout_.params_mode              ? blsw_matrices_info_params(msg_len_, layout_.params_mode, &info)
                                                 : blsw_matrices_info(msg_len_, layout_.n_keys, 1, &info),
              "blsw_matrices_info");
        return info.n_constraints;
    }
    const blsw_layout_t& layout() const { return layout_; }
    // witness_assignment of system i after verify: n_witness elements of 6 little-endian u64 limbs (Montgomery form: arkworks' in-memory Fq)
    std::vector<uint64_t> witness_assignment(size_t i) const {
        if (!witness_.get() || i >= n_) throw Error("witness_assignment before verify / out of range", BLSW_ERR_ARG);
        std::vector<uint64_t> w((size_t)layout_.n_witness * 6);
        witness_.download(w.data(), w.size() * 8, i * (size_t)layout_.n_witness * 48);
        return w;
    }
    // instance_assignment of system i after verify: n_instance_vars elements (element 0 = one; then the coordinates of the points allocated with
    // AllocationMode::Input, in allocation order), same element encoding as witness_assignment
    std::vector<uint64_t> instance_assignment(size_t i) const {
        if (i >= n_) throw Error("instance_assignment out of range", BLSW_ERR_ARG);
        std::vector<uint64_t> v((size_t)layout_.n_instance_vars * 6);
        if (layout_.n_instance_vars == 1 || !instance_.get()) {  // the constant one (R mod p)
            if (layout_.n_instance_vars != 1) throw Error("instance_assignment before verify", BLSW_ERR_ARG);
            const uint64_t one[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
            std::memcpy(v.data(), one, 48);
            return v;
        }
        instance_.download(v.data(), v.size() * 8, i * v.size() * 8);
        return v;
    }
    // decode statuses (BLSW_ST_*) of (public key, signature) of system i after verify
    std::array<int32_t, 2> status(size_t i) const { return {status_.at(2 * i), status_.at(2 * i + 1)}; }

   private:
    friend class ParametersVar;
    friend class PublicKeyVar;
    friend class SignatureVar;
    friend struct BlsSignatureVerifyGadget;
    size_t n_;
    uint32_t msg_len_;
    int device_;
    blsw_layout_t layout_;
    blsw_engine_t* engine_ = nullptr;
    detail::DeviceBytes workspace_, witness_, instance_, result_, d_status_, pk_xy_, sig_xy_;
    // AllocationMode of the key / the signature: part of the circuit shape (blsw_layout_io)
    void set_io(uint32_t pk_mode, uint32_t sig_mode) {
        if (engine_) throw Error("new_variable after verify", BLSW_ERR_ARG);
        if ((pk_mode || sig_mode) && (layout_.params_mode || layout_.n_keys)) throw Error("AllocationMode::Input: the single-key circuit with Constant parameters", BLSW_ERR_ARG);
        if (pk_mode || sig_mode || layout_.pk_mode || layout_.sig_mode) check(blsw_layout_io(msg_len_, pk_mode, sig_mode, &layout_), "blsw_layout_io");
    }
    std::vector<int32_t> status_;
};

// constraints.rs:341: the message bytes of every system, allocated as witnesses (8 booleans per byte at the head of the vector)
class MessageVar {
   public:
    const std::vector<uint8_t>& bytes() const { return bytes_; }

   private:
    friend class UInt8;
    std::vector<uint8_t> bytes_;  // [n][msg_len]
};
class UInt8 {
   public:
    static MessageVar new_witness_vec(ConstraintSystem& cs, const std::vector<std::vector<uint8_t>>& msgs) {
        if (msgs.size() != cs.num_instances()) throw Error("UInt8::new_witness_vec: one message per system", BLSW_ERR_ARG);
        MessageVar m;
        for (auto& x : msgs) {
            if (x.size() != cs.msg_len()) throw Error("UInt8::new_witness_vec: message length != the circuit's", BLSW_ERR_ARG);
            m.bytes_.insert(m.bytes_.end(), x.begin(), x.end());
        }
        return m;
    }
};

// constraints.rs:194-212. Constant (every circuit of the reference) or Witness; fixes the circuit shape of `cs`
class ParametersVar {
   public:
    static ParametersVar new_variable(ConstraintSystem& cs, const Parameters&, AllocationMode mode) {
        if (mode == AllocationMode::Input) throw Error("ParametersVar: AllocationMode::Input (instance variables are not produced)", BLSW_ERR_ARG);
        if (cs.engine_) throw Error("ParametersVar::new_variable after verify", BLSW_ERR_ARG);
        // the allocation modes together fix the circuit shape, whatever the order the three new_variable calls run in (C++ argument evaluation order)
        if (cs.layout_.pk_mode || cs.layout_.sig_mode) {
            if (mode == AllocationMode::Witness) throw Error("ParametersVar: AllocationMode::Witness together with Input keys / signatures", BLSW_ERR_ARG);
        } else {
            check(blsw_layout_params(cs.msg_len_, mode == AllocationMode::Witness ? 1u : 0u, &cs.layout_), "blsw_layout_params");
        }
        ParametersVar p;
        p.cs_ = &cs;
        return p;
    }

   private:
    friend struct BlsSignatureVerifyGadget;
    ConstraintSystem* cs_ = nullptr;
};
// constraints.rs:214-232 / 234-249: Witness (G1Var / G2Var::new_variable with their in-circuit subgroup checks) or Input (the coordinates are public
// inputs: new_variable_omit_prime_order_check); Constant keys / signatures are a different circuit and not offered
class PublicKeyVar {
   public:
    static PublicKeyVar new_variable(ConstraintSystem& cs, const std::vector<PublicKey>& keys, AllocationMode mode) {
        if (mode == AllocationMode::Constant) throw Error("PublicKeyVar: AllocationMode::Constant is not on the GPU path", BLSW_ERR_ARG);
        if (keys.size() != cs.num_instances()) throw Error("PublicKeyVar::new_variable: one key per system", BLSW_ERR_ARG);
        cs.set_io(mode == AllocationMode::Input ? 1u : 0u, cs.layout_.sig_mode);
        PublicKeyVar v;
        v.keys_ = keys;
        return v;
    }

   private:
    friend struct BlsSignatureVerifyGadget;
    std::vector<PublicKey> keys_;
};
class SignatureVar {
   public:
    static SignatureVar new_variable(ConstraintSystem& cs, const std::vector<Signature>& sigs, AllocationMode mode) {
        if (mode == AllocationMode::Constant) throw Error("SignatureVar: AllocationMode::Constant is not on the GPU path", BLSW_ERR_ARG);
        if (sigs.size() != cs.num_instances()) throw Error("SignatureVar::new_variable: one signature per system", BLSW_ERR_ARG);
        cs.set_io(cs.layout_.pk_mode, mode == AllocationMode::Input ? 1u : 0u);
        SignatureVar v;
        v.sigs_ = sigs;
        return v;
    }

   private:
    friend struct BlsSignatureVerifyGadget;
    std::vector<Signature> sigs_;
};

// Boolean<ConstraintF>: one Boolean of every system — the gadget's output, or a bitmap entry allocated with new_witness (constraints.rs:414-419)
class Boolean {
   public:
    const std::vector<bool>& value() const { return v_; }
    static Boolean new_witness(ConstraintSystem& cs, const std::vector<bool>& values) {
        if (values.size() != cs.num_instances()) throw Error("Boolean::new_witness: one value per system", BLSW_ERR_ARG);
        Boolean b;
        b.v_ = values;
        return b;
    }

   private:
    friend struct BlsSignatureVerifyGadget;
    std::vector<bool> v_;
};
// UInt32<ConstraintF>: the effective public key count of aggregate_verify (constraints.rs:177-189)
class UInt32 {
   public:
    const std::vector<uint32_t>& value() const { return v_; }

   private:
    friend struct BlsSignatureVerifyGadget;
    std::vector<uint32_t> v_;
};

struct BlsSignatureVerifyGadget {
    // constraints.rs:90-128 for the n systems of `cs`: decodes the keys and signatures, generates every witness of every system on the GPU
    // (blsw_engine_submit_bytes on a direct-mode engine) and returns the output Booleans. Synchronous.
    static Boolean verify(const ParametersVar& parameters, const PublicKeyVar& public_key, const MessageVar& message, const SignatureVar& signature) {
        if (!parameters.cs_) throw Error("verify: parameters were not allocated in a ConstraintSystem", BLSW_ERR_ARG);
        ConstraintSystem& cs = *parameters.cs_;
        const size_t n = cs.n_;
        if (public_key.keys_.size() != n || signature.sigs_.size() != n || message.bytes().size() != n * cs.msg_len_)
            throw Error("verify: variables of another ConstraintSystem", BLSW_ERR_ARG);
        // one circuit shape per ConstraintSystem: aggregate_verify has replaced the layout (num_witness_variables would be the aggregate circuit's)
        if (cs.layout_.n_keys) throw Error("verify: this ConstraintSystem was synthesised by aggregate_verify; use a new one", BLSW_ERR_ARG);
        if (!cs.engine_) {
            blsw_engine_options_t opt;
            check(blsw_engine_options_default(&opt), "blsw_engine_options_default");
            opt.device = cs.device_;
            opt.params_mode = cs.layout_.params_mode;
            opt.pk_mode = cs.layout_.pk_mode;
            opt.sig_mode = cs.layout_.sig_mode;
            uint64_t bytes = 0;
            check(blsw_engine_workspace_bytes_ex(n, cs.msg_len_, 1, 1, &opt, &bytes), "blsw_engine_workspace_bytes_ex");
            if (cs.device_ >= 0) hip_check(hipSetDevice(cs.device_), "hipSetDevice");
            cs.workspace_ = detail::DeviceBytes(bytes);
            cs.witness_ = detail::DeviceBytes(n * (size_t)cs.layout_.n_witness * 48);
            cs.instance_ = detail::DeviceBytes(n * (size_t)cs.layout_.n_instance_vars * 48);
            cs.result_ = detail::DeviceBytes(n * 4);
            cs.d_status_ = detail::DeviceBytes(n * 8);
            cs.pk_xy_ = detail::DeviceBytes(n * 96);
            cs.sig_xy_ = detail::DeviceBytes(n * 192);
            check(blsw_engine_create_ex(&cs.engine_, n, cs.msg_len_, 1, 1, &opt, cs.workspace_.get(), bytes), "blsw_engine_create_ex");
        }
        std::vector<uint8_t> pk(n * 48), sg(n * 96);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(&pk[48 * i], public_key.keys_[i].bytes.data(), 48);
            std::memcpy(&sg[96 * i], signature.sigs_[i].bytes.data(), 96);
        }
        detail::DeviceBytes d_pk(pk.size()), d_sg(sg.size()), d_msg(message.bytes().size());
        d_pk.upload(pk.data(), pk.size());
        d_sg.upload(sg.data(), sg.size());
        if (!message.bytes().empty()) d_msg.upload(message.bytes().data(), message.bytes().size());
        const bool io = cs.layout_.pk_mode || cs.layout_.sig_mode;
        if (io) {  // public inputs: decode, then the step that also writes instance_assignment; the fallback rule of tests/tests.rs:244-263 is applied below
            check(blsw_decode_batch(static_cast<const uint8_t*>(d_pk.get()), static_cast<const uint8_t*>(d_sg.get()), n, static_cast<uint64_t*>(cs.pk_xy_.get()),
                                    static_cast<uint64_t*>(cs.sig_xy_.get()), static_cast<int32_t*>(cs.d_status_.get()), nullptr),
                  "blsw_decode_batch");
            check(blsw_engine_submit_io(cs.engine_, static_cast<const uint64_t*>(cs.pk_xy_.get()), static_cast<const uint64_t*>(cs.sig_xy_.get()),
                                        static_cast<const uint8_t*>(d_msg.get()), static_cast<uint64_t*>(cs.instance_.get()), static_cast<uint64_t*>(cs.witness_.get()),
                                        cs.layout_.n_witness, static_cast<int32_t*>(cs.result_.get()), nullptr),
                  "blsw_engine_submit_io");
        } else {
            check(blsw_engine_submit_bytes(cs.engine_, static_cast<const uint8_t*>(d_pk.get()), static_cast<const uint8_t*>(d_sg.get()),
                                           static_cast<const uint8_t*>(d_msg.get()), static_cast<uint64_t*>(cs.pk_xy_.get()), static_cast<uint64_t*>(cs.sig_xy_.get()),
                                           static_cast<int32_t*>(cs.d_status_.get()), static_cast<uint64_t*>(cs.witness_.get()), cs.layout_.n_witness,
                                           static_cast<int32_t*>(cs.result_.get()), nullptr),
                  "blsw_engine_submit_bytes");
        }
        check(blsw_engine_flush(cs.engine_, nullptr), "blsw_engine_flush");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        std::vector<int32_t> r(n);
        cs.result_.download(r.data(), n * 4);
        cs.status_.resize(2 * n);
        cs.d_status_.download(cs.status_.data(), n * 8);
        if (io)
            for (size_t i = 0; i < n; i++)
                if (cs.status_[2 * i] | cs.status_[2 * i + 1]) r[i] = 0;
        Boolean b;
        b.v_.resize(n);
        for (size_t i = 0; i < n; i++) b.v_[i] = r[i] == 1;
        return b;
    }

    // constraints.rs:153-167 for the n systems of `cs`: public_keys[k] / bitmap[k] hold key k / bit k of every system (the reference passes
    // slices of K variables of one system). Returns (result, effective public key count). The circuit is the one of blsw_layout_aggregate:
    // keys, bitmap, msg, sig allocated in the order of constraints.rs:378-441, Constant parameters. A key that FAILS TO DECODE (bad encoding,
    // not on the curve, not in the subgroup) or a signature that does not decode to a non-identity point makes its system false (status()),
    // as in verify. The infinity encoding of a KEY decodes (PublicKey::try_from accepts it, and a key whose bitmap bit is 0 never enters
    // the aggregate, constraints.rs:169-191): such a system returns the gadget's own Boolean. Synchronous; direct-mode batch entry
    // blsw_aggregate_verify_batch.
    static std::pair<Boolean, UInt32> aggregate_verify(const ParametersVar& parameters, const std::vector<PublicKeyVar>& public_keys, const std::vector<Boolean>& bitmap,
                                                       const MessageVar& message, const SignatureVar& signature) {
        if (!parameters.cs_) throw Error("aggregate_verify: parameters were not allocated in a ConstraintSystem", BLSW_ERR_ARG);
        ConstraintSystem& cs = *parameters.cs_;
        const size_t n = cs.n_, K = public_keys.size();
        if (K == 0 || bitmap.size() != K) throw Error("aggregate_verify: public_keys.len() == bitmap.len() > 0", BLSW_ERR_ARG);  // constraints.rs:160
        if (cs.layout_.params_mode || cs.engine_) throw Error("aggregate_verify: Constant parameters, a ConstraintSystem not used by verify", BLSW_ERR_ARG);
        if (signature.sigs_.size() != n || message.bytes().size() != n * cs.msg_len_) throw Error("aggregate_verify: variables of another ConstraintSystem", BLSW_ERR_ARG);
        check(blsw_layout_aggregate(cs.msg_len_, (uint32_t)K, &cs.layout_), "blsw_layout_aggregate");
        if (cs.device_ >= 0) hip_check(hipSetDevice(cs.device_), "hipSetDevice");
        // compressed inputs, system-major: keys [n][K][48], bitmap [n][K], signatures [n][96]
        std::vector<uint8_t> pk(n * K * 48), bm(n * K), sg(n * 96);
        for (size_t k = 0; k < K; k++) {
            if (public_keys[k].keys_.size() != n || bitmap[k].v_.size() != n) throw Error("aggregate_verify: one key and one bit per system", BLSW_ERR_ARG);
            for (size_t i = 0; i < n; i++) {
                std::memcpy(&pk[(i * K + k) * 48], public_keys[k].keys_[i].bytes.data(), 48);
                bm[i * K + k] = bitmap[k].v_[i] ? 1 : 0;
            }
        }
        for (size_t i = 0; i < n; i++) std::memcpy(&sg[96 * i], signature.sigs_[i].bytes.data(), 96);
        // decode (blsw_decode_batch takes as many keys as signatures: the keys with a block of zero bytes beside them, then the signatures likewise)
        const size_t m = n * K;
        detail::DeviceBytes d_pk(m * 48), d_sg(m * 96), d_pk_xy(m * 96), d_sg_xy(m * 192), d_st(m * 8), d_bm(m), d_msg(message.bytes().size());
        hip_check(hipMemset(d_sg.get(), 0, m * 96), "hipMemset");
        d_pk.upload(pk.data(), pk.size());
        check(blsw_decode_batch(static_cast<const uint8_t*>(d_pk.get()), static_cast<const uint8_t*>(d_sg.get()), m, static_cast<uint64_t*>(d_pk_xy.get()),
                                static_cast<uint64_t*>(d_sg_xy.get()), static_cast<int32_t*>(d_st.get()), nullptr),
              "blsw_decode_batch");
        std::vector<int32_t> st_keys(2 * m);
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        d_st.download(st_keys.data(), m * 8);
        detail::DeviceBytes d_pk0(n * 48), d_sig(n * 96), d_pk0_xy(n * 96), d_st2(n * 8);
        cs.sig_xy_ = detail::DeviceBytes(n * 192);
        hip_check(hipMemset(d_pk0.get(), 0, n * 48), "hipMemset");
        d_sig.upload(sg.data(), sg.size());
        check(blsw_decode_batch(static_cast<const uint8_t*>(d_pk0.get()), static_cast<const uint8_t*>(d_sig.get()), n, static_cast<uint64_t*>(d_pk0_xy.get()),
                                static_cast<uint64_t*>(cs.sig_xy_.get()), static_cast<int32_t*>(d_st2.get()), nullptr),
              "blsw_decode_batch");
        std::vector<int32_t> st_sig(2 * n);
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        d_st2.download(st_sig.data(), n * 8);
        d_bm.upload(bm.data(), bm.size());
        if (!message.bytes().empty()) d_msg.upload(message.bytes().data(), message.bytes().size());
        uint64_t bytes = 0;
        check(blsw_aggregate_workspace_bytes(n, cs.msg_len_, (uint32_t)K, &bytes), "blsw_aggregate_workspace_bytes");
        cs.workspace_ = detail::DeviceBytes(bytes);
        cs.witness_ = detail::DeviceBytes(n * (size_t)cs.layout_.n_witness * 48);
        cs.result_ = detail::DeviceBytes(n * 4);
        detail::DeviceBytes d_count(n * 4);
        check(blsw_aggregate_verify_batch(static_cast<const uint64_t*>(d_pk_xy.get()), static_cast<const uint8_t*>(d_bm.get()), (uint32_t)K,
                                          static_cast<const uint64_t*>(cs.sig_xy_.get()), static_cast<const uint8_t*>(d_msg.get()), cs.msg_len_, n,
                                          static_cast<uint64_t*>(cs.witness_.get()), cs.layout_.n_witness, static_cast<int32_t*>(cs.result_.get()),
                                          static_cast<uint32_t*>(d_count.get()), cs.workspace_.get(), bytes, nullptr),
              "blsw_aggregate_verify_batch");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        std::vector<int32_t> r(n);
        cs.result_.download(r.data(), n * 4);
        UInt32 count;
        count.v_.resize(n);
        d_count.download(count.v_.data(), n * 4);
        // status(i): the first key of system i that failed to decode (else OK: a well-formed identity key is not a failure), and the signature's
        cs.status_.assign(2 * n, BLSW_ST_OK);
        Boolean b;
        b.v_.resize(n);
        for (size_t i = 0; i < n; i++) {
            for (size_t k = 0; k < K && cs.status_[2 * i] == BLSW_ST_OK; k++) {
                const int32_t st = st_keys[2 * (i * K + k)];
                if (st != BLSW_ST_IDENTITY) cs.status_[2 * i] = st;
            }
            cs.status_[2 * i + 1] = st_sig[2 * i + 1];
            b.v_[i] = r[i] == 1 && cs.status_[2 * i] == BLSW_ST_OK && cs.status_[2 * i + 1] == BLSW_ST_OK;
        }
        return {b, count};
    }
};

// The native scheme (src/bls.rs:395-458, `impl SignatureScheme for BLS`) for batches: one secret key / message / signature per element.
struct SecretKey {
    std::array<uint8_t, 32> le;  // PrivateKey::try_from(&[u8]) (bls.rs:97-103): little-endian Fr
    static SecretKey try_from(const std::string& hex_be) {  // the fixtures' big-endian hex ("0x..." allowed)
        SecretKey k;
        auto b = detail::unhex(hex_be, 32);
        for (int i = 0; i < 32; i++) k.le[i] = b[31 - i];
        return k;
    }
};
struct BLS {
    struct Signed {
        std::vector<Signature> signatures;    // Signature (compressed), the infinity encoding where status != BLSW_ST_OK
        std::vector<PublicKey> public_keys;   // PublicKey::from(&sk)
        std::vector<int32_t> status;          // BLSW_ST_OK, BLSW_ST_INVALID_SECRET_KEY (sk = 0: Err(InvalidSecretKey), bls.rs:417-419), BLSW_ST_BAD_ENCODING (sk >= r)
    };
    // BLS::sign (bls.rs:411-425) and PublicKey::from(&sk) (bls.rs:183-195): messages [n][msg_len]
    static Signed sign(const Parameters&, const std::vector<SecretKey>& sks, const std::vector<std::vector<uint8_t>>& messages) {
        const size_t n = sks.size();
        if (n == 0 || messages.size() != n) throw Error("BLS::sign: one message per key", BLSW_ERR_ARG);
        const uint32_t msg_len = (uint32_t)messages[0].size();
        std::vector<uint8_t> sk(n * 32), msg(n * (size_t)msg_len);
        for (size_t i = 0; i < n; i++) {
            if (messages[i].size() != msg_len) throw Error("BLS::sign: messages of one length per batch", BLSW_ERR_ARG);
            std::memcpy(&sk[32 * i], sks[i].le.data(), 32);
            if (msg_len) std::memcpy(&msg[(size_t)msg_len * i], messages[i].data(), msg_len);
        }
        uint64_t bytes = 0;
        check(blsw_hash_to_g2_workspace_bytes(n, msg_len, &bytes), "blsw_hash_to_g2_workspace_bytes");
        detail::DeviceBytes d_sk(sk.size()), d_msg(msg.size()), d_sig(n * 96), d_pk(n * 48), d_st(n * 4), ws(bytes);
        d_sk.upload(sk.data(), sk.size());
        if (!msg.empty()) d_msg.upload(msg.data(), msg.size());
        check(blsw_sign_batch(static_cast<const uint8_t*>(d_sk.get()), static_cast<const uint8_t*>(d_msg.get()), msg_len, n, static_cast<uint8_t*>(d_sig.get()), nullptr,
                              static_cast<uint8_t*>(d_pk.get()), nullptr, static_cast<int32_t*>(d_st.get()), ws.get(), bytes, nullptr),
              "blsw_sign_batch");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        std::vector<uint8_t> sg(n * 96), pk(n * 48);
        Signed out;
        out.status.resize(n);
        d_sig.download(sg.data(), sg.size());
        d_pk.download(pk.data(), pk.size());
        d_st.download(out.status.data(), n * 4);
        out.signatures.resize(n);
        out.public_keys.resize(n);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(out.signatures[i].bytes.data(), &sg[96 * i], 96);
            std::memcpy(out.public_keys[i].bytes.data(), &pk[48 * i], 48);
        }
        return out;
    }
    // BLS::verify (bls.rs:427-458) for n (pk, msg, sig) triples: the native pairing check as values (no circuit, no witness tensor).
    // An identity or undecodable key / signature is Err(..) in the reference and `false` here, with the reason in `status` ([n][2], BLSW_ST_*).
    static std::vector<bool> verify(const Parameters&, const std::vector<PublicKey>& pks, const std::vector<std::vector<uint8_t>>& messages,
                                    const std::vector<Signature>& sigs, std::vector<int32_t>* status = nullptr) {
        const size_t n = pks.size();
        if (n == 0 || messages.size() != n || sigs.size() != n) throw Error("BLS::verify: one message and one signature per key", BLSW_ERR_ARG);
        const uint32_t msg_len = (uint32_t)messages[0].size();
        std::vector<uint8_t> pk(n * 48), sg(n * 96), msg(n * (size_t)msg_len);
        for (size_t i = 0; i < n; i++) {
            if (messages[i].size() != msg_len) throw Error("BLS::verify: messages of one length per batch", BLSW_ERR_ARG);
            std::memcpy(&pk[48 * i], pks[i].bytes.data(), 48);
            std::memcpy(&sg[96 * i], sigs[i].bytes.data(), 96);
            if (msg_len) std::memcpy(&msg[(size_t)msg_len * i], messages[i].data(), msg_len);
        }
        uint64_t bytes = 0;
        check(blsw_verify_workspace_bytes(n, msg_len, &bytes), "blsw_verify_workspace_bytes");
        detail::DeviceBytes ws(bytes), d_pk(pk.size()), d_sg(sg.size()), d_msg(msg.size()), d_st(n * 8), d_res(n * 4);
        d_pk.upload(pk.data(), pk.size());
        d_sg.upload(sg.data(), sg.size());
        if (!msg.empty()) d_msg.upload(msg.data(), msg.size());
        // the native algorithm as values (blsw_verify_batch): decode + subgroup checks, hash to G2, projective two-pair Miller loop, final exponentiation
        check(blsw_verify_batch(static_cast<const uint8_t*>(d_pk.get()), static_cast<const uint8_t*>(d_sg.get()), static_cast<const uint8_t*>(d_msg.get()), msg_len, n,
                                static_cast<int32_t*>(d_res.get()), static_cast<int32_t*>(d_st.get()), ws.get(), bytes, nullptr),
              "blsw_verify_batch");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        std::vector<int32_t> r(n);
        d_res.download(r.data(), n * 4);
        if (status) {
            status->resize(2 * n);
            d_st.download(status->data(), n * 8);
        }
        std::vector<bool> out(n);
        for (size_t i = 0; i < n; i++) out[i] = r[i] == 1;
        return out;
    }
};

}  // namespace blsw
