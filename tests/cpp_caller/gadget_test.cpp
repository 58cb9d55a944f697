// TEST PROGRAM: the reference's gadget test (src/constraints.rs:318-376, `test_verify`) written against include/blsw.hpp — the C++ host side
// above the C ABI. Same inputs, same statements, same assertion; the three messages of the reference's loop are the three systems of one
// ConstraintSystem batch here. Prints key=value pairs for tests/test_gpu_parity.py (witness digest of system 0 against the oracle).
//   gadget_test [constant|witness] [count-constraints]
//   gadget_test sign <file>    every line "<sk hex, big-endian> <msg32 hex>": BLS::sign + PublicKey::from(&sk) for the batch (tests/tests.rs:202-237); prints
//                              "<status> <sig96 hex> <pk48 hex>" per line
//   gadget_test verify <file>  every line "<pk48 hex> <msg32 hex> <sig96 hex>": BLS::verify for the batch (tests/tests.rs:239-268); prints "<0/1> <st_pk> <st_sig>"
//   gadget_test aggregate      the reference's test_aggregate_verify and test_aggregate_verify_neg (constraints.rs:378-521) as the two systems of one batch
#include <cstdio>
#include <cstring>
#include <string>

#include "blsw.hpp"

using namespace blsw;

// position-weighted sum of the assignment's u64 words, mod 2^64 (vectorisable on the checking side)
static uint64_t digest(const std::vector<uint64_t>& w) {
    uint64_t h = 0;
    for (size_t k = 0; k < w.size(); k++) h += w[k] * (2 * (uint64_t)k + 1);
    return h;
}

// constraints.rs:378-441 (bitmap: the first two of 512 keys) and :452-521 (all 512): expected true / false, effective key counts 2 / 512
static int aggregate_tests() {
    ConstraintSystem cs(2, 32);
    const PublicKey pub_key1 = PublicKey::try_from("a491d1b0ecd9bb917989f0e74f0dea0422eac4a873e5e2644f368dffb9a6e20fd6e10c1b77654d067c0618f6e5a7f79a");
    const PublicKey pub_key2 = PublicKey::try_from("b301803f8b5ac4a1133581fc676dfedc60d891dd5fa99028805e5ea5b08d3491af75d0707adab3b70c6a6a580217bf81");
    std::vector<PublicKeyVar> pub_keys;
    pub_keys.push_back(PublicKeyVar::new_variable(cs, {pub_key1, pub_key1}, AllocationMode::Witness));
    for (int i = 1; i < 512; i++) pub_keys.push_back(PublicKeyVar::new_variable(cs, {pub_key2, pub_key2}, AllocationMode::Witness));
    std::vector<Boolean> bitmap;
    bitmap.push_back(Boolean::new_witness(cs, {true, true}));
    bitmap.push_back(Boolean::new_witness(cs, {true, true}));
    for (int i = 2; i < 512; i++) bitmap.push_back(Boolean::new_witness(cs, {false, true}));
    const std::vector<uint8_t> m = detail::unhex("5656565656565656565656565656565656565656565656565656565656565656", 32);
    const MessageVar msg = UInt8::new_witness_vec(cs, {m, m});
    const Signature sig = Signature::try_from(
        "912c3615f69575407db9392eb21fee18fff797eeb2fbe1816366ca2a08ae574d8824dbfafb4c9eaa1cf61b63c6f9b69911f269b664c42947dd1b53ef1081926c1e82bb2a465f927124b08391a5249"
        "036146d6f3f1e17ff5f162f779746d830d1");
    const auto [result, count] = BlsSignatureVerifyGadget::aggregate_verify(ParametersVar::new_variable(cs, Parameters{}, AllocationMode::Constant), pub_keys, bitmap, msg,
                                                                            SignatureVar::new_variable(cs, {sig, sig}, AllocationMode::Witness));
    printf("verification_result_0=%d verification_result_1=%d effective_public_key_count_0=%u effective_public_key_count_1=%u num_witness_variables=%llu", (int)result.value()[0],
           (int)result.value()[1], count.value()[0], count.value()[1], (unsigned long long)cs.num_witness_variables());
    if (result.value()[0] != true || result.value()[1] != false) return 2;
    printf(" digest0=%llu digest1=%llu\n", (unsigned long long)digest(cs.witness_assignment(0)), (unsigned long long)digest(cs.witness_assignment(1)));
    return 0;
}

// a well-formed identity key (PublicKey::try_from accepts the infinity encoding) masked out (system 0) and unmasked (system 1): the C++ mirror
// must return the gadget's own Boolean, not force false; and verify() on a ConstraintSystem that aggregate_verify has synthesised is refused
static int aggregate_identity_tests() {
    ConstraintSystem cs(2, 32);
    const PublicKey pub_key1 = PublicKey::try_from("a491d1b0ecd9bb917989f0e74f0dea0422eac4a873e5e2644f368dffb9a6e20fd6e10c1b77654d067c0618f6e5a7f79a");
    const PublicKey pub_key2 = PublicKey::try_from("b301803f8b5ac4a1133581fc676dfedc60d891dd5fa99028805e5ea5b08d3491af75d0707adab3b70c6a6a580217bf81");
    const PublicKey infinity = PublicKey::try_from("c00000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000");
    std::vector<PublicKeyVar> pub_keys;
    pub_keys.push_back(PublicKeyVar::new_variable(cs, {pub_key1, pub_key1}, AllocationMode::Witness));
    pub_keys.push_back(PublicKeyVar::new_variable(cs, {pub_key2, pub_key2}, AllocationMode::Witness));
    pub_keys.push_back(PublicKeyVar::new_variable(cs, {infinity, infinity}, AllocationMode::Witness));
    std::vector<Boolean> bitmap;
    bitmap.push_back(Boolean::new_witness(cs, {true, true}));
    bitmap.push_back(Boolean::new_witness(cs, {true, true}));
    bitmap.push_back(Boolean::new_witness(cs, {false, true}));
    const std::vector<uint8_t> m = detail::unhex("5656565656565656565656565656565656565656565656565656565656565656", 32);
    const MessageVar msg = UInt8::new_witness_vec(cs, {m, m});
    const Signature sig = Signature::try_from(
        "912c3615f69575407db9392eb21fee18fff797eeb2fbe1816366ca2a08ae574d8824dbfafb4c9eaa1cf61b63c6f9b69911f269b664c42947dd1b53ef1081926c1e82bb2a465f927124b08391a5249"
        "036146d6f3f1e17ff5f162f779746d830d1");
    const ParametersVar params = ParametersVar::new_variable(cs, Parameters{}, AllocationMode::Constant);
    const SignatureVar sig_var = SignatureVar::new_variable(cs, {sig, sig}, AllocationMode::Witness);
    const auto [result, count] = BlsSignatureVerifyGadget::aggregate_verify(params, pub_keys, bitmap, msg, sig_var);
    printf("verification_result_0=%d verification_result_1=%d effective_public_key_count_0=%u effective_public_key_count_1=%u status_pk_0=%d status_pk_1=%d num_witness_variables=%llu",
           (int)result.value()[0], (int)result.value()[1], count.value()[0], count.value()[1], cs.status(0)[0], cs.status(1)[0], (unsigned long long)cs.num_witness_variables());
    printf(" digest0=%llu digest1=%llu", (unsigned long long)digest(cs.witness_assignment(0)), (unsigned long long)digest(cs.witness_assignment(1)));
    int refused = 0;
    try {
        BlsSignatureVerifyGadget::verify(params, pub_keys[0], msg, sig_var);
    } catch (const Error&) {
        refused = 1;
    }
    printf(" verify_after_aggregate_refused=%d\n", refused);
    return refused ? 0 : 3;
}

static std::string hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) {
        s += d[p[i] >> 4];
        s += d[p[i] & 15];
    }
    return s;
}
// the native scheme through the same header: BLS::sign / BLS::verify over the lines of a file
static int native_tests(const char* what, const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) return 20;
    char a[256], b[256], c[256];
    if (!strcmp(what, "sign")) {
        std::vector<SecretKey> sks;
        std::vector<std::vector<uint8_t>> msgs;
        while (fscanf(f, "%255s %255s", a, b) == 2) {
            sks.push_back(SecretKey::try_from(a));
            msgs.push_back(detail::unhex(b, 32));
        }
        fclose(f);
        const BLS::Signed r = BLS::sign(Parameters{}, sks, msgs);
        for (size_t i = 0; i < sks.size(); i++)
            printf("%d %s %s\n", r.status[i], hex(r.signatures[i].bytes.data(), 96).c_str(), hex(r.public_keys[i].bytes.data(), 48).c_str());
        return 0;
    }
    std::vector<PublicKey> pks;
    std::vector<Signature> sigs;
    std::vector<std::vector<uint8_t>> msgs;
    while (fscanf(f, "%255s %255s %255s", a, b, c) == 3) {
        pks.push_back(PublicKey::try_from(a));
        msgs.push_back(detail::unhex(b, 32));
        sigs.push_back(Signature::try_from(c));
    }
    fclose(f);
    std::vector<int32_t> st;
    const std::vector<bool> ok = BLS::verify(Parameters{}, pks, msgs, sigs, &st);
    for (size_t i = 0; i < pks.size(); i++) printf("%d %d %d\n", (int)ok[i], st[2 * i], st[2 * i + 1]);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 2 && (!strcmp(argv[1], "sign") || !strcmp(argv[1], "verify"))) {
        try {
            return native_tests(argv[1], argv[2]);
        } catch (const Error& e) {
            fprintf(stderr, "%s\n", e.what());
            return 10;
        }
    }
    if (argc > 1 && !strcmp(argv[1], "aggregate-identity")) {
        try {
            return aggregate_identity_tests();
        } catch (const Error& e) {
            fprintf(stderr, "%s\n", e.what());
            return 10;
        }
    }
    if (argc > 1 && !strcmp(argv[1], "aggregate")) {
        try {
            return aggregate_tests();
        } catch (const Error& e) {
            fprintf(stderr, "%s\n", e.what());
            return 10;
        }
    }
    const bool params_witness = argc > 1 && !strcmp(argv[1], "witness");
    // "input": the same test with the key and the signature allocated as PUBLIC INPUTS (constraints.rs:214-249 take any AllocationMode)
    const bool io_input = argc > 1 && !strcmp(argv[1], "input");
    const bool count = argc > 2 && !strcmp(argv[2], "count-constraints");
    try {
        // use case from tests/test_cases/verify (constraints.rs:320-324)
        const char* msgs[3] = {
            "5656565656565656565656565656565656565656565656565656565656565656",  // valid
            "5656565656565656565656565656565656565656565656565656565656565657",  // invalid
            "7878787878787878787878787878787878787878787878787878787878787878",  // invalid
        };
        const bool expects[3] = {true, false, false};

        ConstraintSystem cs(3, 32);

        const PublicKey public_key = PublicKey::try_from("a491d1b0ecd9bb917989f0e74f0dea0422eac4a873e5e2644f368dffb9a6e20fd6e10c1b77654d067c0618f6e5a7f79a");
        std::vector<std::vector<uint8_t>> msg_bytes;
        for (const char* m : msgs) msg_bytes.push_back(detail::unhex(m, 32));
        const MessageVar msg = UInt8::new_witness_vec(cs, msg_bytes);
        const Signature sig = Signature::try_from(
            "882730e5d03f6b42c3abc26d3372625034e1d871b65a8a6b900a56dae22da98abbe1b68f85e49fe7652a55ec3d0591c20767677e33e5cbb1207315c41a9ac03be39c2e7668edc043d6"
            "cb1d9fd93033caa8a1c5b0e84bedaeb6c64972503a43eb");

        const Boolean result = BlsSignatureVerifyGadget::verify(
            ParametersVar::new_variable(cs, Parameters{}, params_witness ? AllocationMode::Witness : AllocationMode::Constant),
            PublicKeyVar::new_variable(cs, std::vector<PublicKey>(3, public_key), io_input ? AllocationMode::Input : AllocationMode::Witness), msg,
            SignatureVar::new_variable(cs, std::vector<Signature>(3, sig), io_input ? AllocationMode::Input : AllocationMode::Witness));

        for (int i = 0; i < 3; i++) {
            printf("verification_result_%d=%d ", i, (int)result.value()[i]);
            if (result.value()[i] != expects[i]) {
                fprintf(stderr, "system %d: expected %d\n", i, (int)expects[i]);
                return 2;
            }
        }
        const std::vector<uint64_t> w0 = cs.witness_assignment(0), w2 = cs.witness_assignment(2);
        printf("num_witness_variables=%llu num_instance_variables=%llu status_pk=%d status_sig=%d digest0=%llu digest2=%llu", (unsigned long long)cs.num_witness_variables(),
               (unsigned long long)cs.num_instance_variables(), cs.status(0)[0], cs.status(0)[1], (unsigned long long)digest(w0), (unsigned long long)digest(w2));
        printf(" instance_digest0=%llu", (unsigned long long)digest(cs.instance_assignment(0)));
        if (count) printf(" constraint_size=%llu", (unsigned long long)cs.num_constraints());  // "constraint size" of constraints.rs:369-373
        printf("\n");
        // the argument rules of the mirror
        try {
            (void)ParametersVar::new_variable(cs, Parameters{}, AllocationMode::Input);
            return 3;
        } catch (const Error& e) {
            if (e.code != BLSW_ERR_ARG) return 4;
        }
        try {
            (void)PublicKey::try_from("a491");
            return 5;
        } catch (const Error&) {
        }
    } catch (const Error& e) {
        fprintf(stderr, "%s\n", e.what());
        return 10;
    }
    return 0;
}
