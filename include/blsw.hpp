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
//   cs.num_constraints(), cs.num_witness_variables(), Boolean::value()                     constraints.rs:369-373
#pragma once
#include <hip/hip_runtime_api.h>

#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
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

// ark_r1cs_std::alloc::AllocationMode. Input would put a value into instance_assignment, which the engine does not produce.
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
        check(layout_.params_mode ? blsw_matrices_info_params(msg_len_, layout_.params_mode, &info) : blsw_matrices_info(msg_len_, 0, 1, &info), "blsw_matrices_info");
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
    // decode statuses (BLSW_ST_*) of (public key, signature) of system i after verify
    std::array<int32_t, 2> status(size_t i) const { return {status_.at(2 * i), status_.at(2 * i + 1)}; }

   private:
    friend class ParametersVar;
    friend struct BlsSignatureVerifyGadget;
    size_t n_;
    uint32_t msg_len_;
    int device_;
    blsw_layout_t layout_;
    blsw_engine_t* engine_ = nullptr;
    detail::DeviceBytes workspace_, witness_, result_, d_status_, pk_xy_, sig_xy_;
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
        check(blsw_layout_params(cs.msg_len_, mode == AllocationMode::Witness ? 1u : 0u, &cs.layout_), "blsw_layout_params");
        ParametersVar p;
        p.cs_ = &cs;
        return p;
    }

   private:
    friend struct BlsSignatureVerifyGadget;
    ConstraintSystem* cs_ = nullptr;
};
// constraints.rs:214-232 / 234-249: Witness mode (G1Var / G2Var::new_variable with their in-circuit subgroup checks)
class PublicKeyVar {
   public:
    static PublicKeyVar new_variable(ConstraintSystem& cs, const std::vector<PublicKey>& keys, AllocationMode mode) {
        if (mode != AllocationMode::Witness) throw Error("PublicKeyVar: only AllocationMode::Witness is on the GPU path", BLSW_ERR_ARG);
        if (keys.size() != cs.num_instances()) throw Error("PublicKeyVar::new_variable: one key per system", BLSW_ERR_ARG);
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
        if (mode != AllocationMode::Witness) throw Error("SignatureVar: only AllocationMode::Witness is on the GPU path", BLSW_ERR_ARG);
        if (sigs.size() != cs.num_instances()) throw Error("SignatureVar::new_variable: one signature per system", BLSW_ERR_ARG);
        SignatureVar v;
        v.sigs_ = sigs;
        return v;
    }

   private:
    friend struct BlsSignatureVerifyGadget;
    std::vector<Signature> sigs_;
};

// Boolean<ConstraintF>: the gadget's output of every system
class Boolean {
   public:
    const std::vector<bool>& value() const { return v_; }

   private:
    friend struct BlsSignatureVerifyGadget;
    std::vector<bool> v_;
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
        if (!cs.engine_) {
            blsw_engine_options_t opt;
            check(blsw_engine_options_default(&opt), "blsw_engine_options_default");
            opt.device = cs.device_;
            opt.params_mode = cs.layout_.params_mode;
            uint64_t bytes = 0;
            check(blsw_engine_workspace_bytes_ex(n, cs.msg_len_, 1, 1, &opt, &bytes), "blsw_engine_workspace_bytes_ex");
            if (cs.device_ >= 0) hip_check(hipSetDevice(cs.device_), "hipSetDevice");
            cs.workspace_ = detail::DeviceBytes(bytes);
            cs.witness_ = detail::DeviceBytes(n * (size_t)cs.layout_.n_witness * 48);
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
        check(blsw_engine_submit_bytes(cs.engine_, static_cast<const uint8_t*>(d_pk.get()), static_cast<const uint8_t*>(d_sg.get()),
                                       static_cast<const uint8_t*>(d_msg.get()), static_cast<uint64_t*>(cs.pk_xy_.get()), static_cast<uint64_t*>(cs.sig_xy_.get()),
                                       static_cast<int32_t*>(cs.d_status_.get()), static_cast<uint64_t*>(cs.witness_.get()), cs.layout_.n_witness,
                                       static_cast<int32_t*>(cs.result_.get()), nullptr),
              "blsw_engine_submit_bytes");
        check(blsw_engine_flush(cs.engine_, nullptr), "blsw_engine_flush");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        std::vector<int32_t> r(n);
        cs.result_.download(r.data(), n * 4);
        cs.status_.resize(2 * n);
        cs.d_status_.download(cs.status_.data(), n * 8);
        Boolean b;
        b.v_.resize(n);
        for (size_t i = 0; i < n; i++) b.v_[i] = r[i] == 1;
        return b;
    }
};

}  // namespace blsw
