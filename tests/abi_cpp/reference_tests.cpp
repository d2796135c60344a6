// reference_tests.cpp — the reference's own tests, restated in C++ against include/plume.hpp (the C++ host side of libplume_hip.so).
//
// Each test carries the name and the assertions of the Rust test it restates, so that it reads like the reference's test suite:
//   rust-k256/tests/signing.rs:48-64            test_sign_v1, test_sign_v2                     (mock RNG :23-44)
//   rust-k256/tests/verification.rs:25-107      plume_v1_test, plume_v2_test
//   rust-k256/tests/verification.rs:283-294     test_hash_to_curve ("abc")
//   rust-k256/src/lib.rs:177-183                test_encode_pt
//   rust-arkworks/src/tests.rs:126-178          test_keygen, test_sign_and_verify
//   rust-arkworks/src/tests.rs:180-299          test_against_zk_nullifier_sig_{pk, g_r, h, h_r, h_sk, c_and_s}
//   rust-arkworks/src/tests.rs:301-316          test_point_sec1_encoding (the 100 k*G vectors, passed as a text file: `k compressed-hex` per line)
// plus what the reference's tests do not hold (SURVEY.md §8c "gaps"): verify() == false cases, the batch twins, mixed V1 / V2 batches, the
// panics of the signer's invariants, verify_non_zk's Err and false cases.  Everything computes on the GPU through the C ABI.
//
//   g++ -std=c++17 -O1 -Wall -Wextra -I include tests/abi_cpp/reference_tests.cpp -L zk-nullifier-sig_amd -lplume_hip -Wl,-rpath,$PWD/zk-nullifier-sig_amd -o /tmp/reference_tests
//   /tmp/reference_tests [kg_vectors.txt]        exit code 0 and "reference_tests ok" on success; without a GPU: "no device" and exit code 3
#include <cstdio>
#include <fstream>
#include <functional>
#include <sstream>

#include "plume.hpp"

using plume_hip::Bytes;
using plume_hip::Bytes32;

static int fails = 0, checks = 0;
#define ASSERT(cond)                                                                             \
    do {                                                                                         \
        checks++;                                                                                \
        if (!(cond)) { std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); fails++; } \
    } while (0)
#define ASSERT_EQ_HEX(bytes, hex) ASSERT(plume_hip::to_hex((bytes).data(), (bytes).size()) == (hex))
#define ASSERT_THROWS(expr, Exc)                                                                 \
    do {                                                                                         \
        checks++;                                                                                \
        bool thrown = false;                                                                     \
        try { (void)(expr); } catch (const Exc&) { thrown = true; }                              \
        if (!thrown) { std::fprintf(stderr, "FAIL %s:%d: %s did not throw %s\n", __FILE__, __LINE__, #expr, #Exc); fails++; } \
    } while (0)

static Bytes bytes_of(const char* s) { return Bytes(s, s + std::strlen(s)); }

// rust-k256/tests/signing.rs:9-21
static const Bytes message = bytes_of("An example app message string");
static const char* R = "93b9323b629f251b8f3fc2dd11f4672c5544e8230d493eceea98a90bda789808";
static const char* SK = "519b423d715f8b581f4fa8ee59f4771a5b44c8130b4e3eacca54a56dda72b464";
static const char* V1_C = "c6a7fc2c926ddbaf20731a479fb6566f2daa5514baae5223fe3b32edbce83254";
static const char* V1_S = "e69f027d84cb6fe5f761e333d12e975fb190d163e8ea132d7de0bd6079ba28ca";
static const char* V2_C = "3dbfb717705010d4f44a70720c95e74b475bd3a783ab0b9e8a6b3b363434eb96";
static const char* V2_S = "528e8fbb6452f82200797b1a73b2947a92524bd611085a920f1177cb8098136b";

// the mock RNG of rust-k256/tests/signing.rs:23-44: every draw is R
struct Mock {
    void fill_bytes(uint8_t* dest, size_t len) {
        const Bytes r = plume_hip::from_hex(R);
        if (len != r.size() || len != 32) throw std::logic_error("Mock: dest.len() == R.len() == 32");
        std::memcpy(dest, r.data(), len);
    }
};
// thread_rng() stand-in
struct OsRng {
    void fill_bytes(uint8_t* dest, size_t len) {
        std::ifstream f("/dev/urandom", std::ios::binary);
        f.read(reinterpret_cast<char*>(dest), (std::streamsize)len);
        if (!f) throw std::runtime_error("/dev/urandom");
    }
};

// ---------------------------------------------------------------------------------------------- rust-k256/tests/signing.rs
namespace signing {
using namespace plume_rustcrypto;
static void test_sign_v1() {
    const SecretKey sk = SecretKey::from_hex(SK).value();
    Mock rng;
    const PlumeSignature sig = PlumeSignature::sign_v1(sk, message, rng);
    ASSERT(NonZeroScalar::from_hex(V1_C).value() == sig.c);
    ASSERT(NonZeroScalar::from_hex(V1_S).value() == sig.s);
    ASSERT(sig.v1specific.has_value());
}
static void test_sign_v2() {
    const SecretKey sk = SecretKey::from_hex(SK).value();
    Mock rng;
    const PlumeSignature sig = PlumeSignature::sign_v2(sk, message, rng);
    ASSERT(NonZeroScalar::from_hex(V2_C).value() == sig.c);
    ASSERT(NonZeroScalar::from_hex(V2_S).value() == sig.s);
    ASSERT(!sig.v1specific.has_value());
}
// the same through the signer type and the trait's method names (rust-k256/src/randomizedsigner.rs:25-47)
static void test_signer_type() {
    const SecretKey sk = SecretKey::from_hex(SK).value();
    Mock rng;
    const PlumeSigner signer(sk, true);
    const PlumeSignature sig = signer.try_sign_with_rng(rng, message);
    ASSERT_EQ_HEX(sig.c.to_bytes(), V1_C);
    ASSERT(sig.message == message);                                  // the output owns a copy of the message (randomizedsigner.rs:98)
    ASSERT(sig.verify());
}
}  // namespace signing

// ----------------------------------------------------------------------------------------- rust-k256/tests/verification.rs
namespace verification {
using namespace plume_rustcrypto;
// what helpers::test_gen_signals yields for M (rust-k256/tests/verification.rs:159-275), as the arkworks crate's tests pin it (rust-arkworks/src/tests.rs:189-262)
static AffinePoint PK() { return AffinePoint::from_hex("0cec028ee08d09e02672a68310814354f9eabfff0de6dacc1cd3a774496076ae", "eff471fba0409897b6a48e8801ad12f95d0009b753cf8f51c128bf6b0bd27fbd"); }
static AffinePoint NULLIFIER() { return AffinePoint::from_hex("57bc3ed28172ef8adde4b9e0c2cce745fcc5a66473a45c1e626f1d0c67e55830", "6a2f41488d58f33ae46edd2188e111609f9f3ae67ea38fa891d6087fe59ecb73"); }
static AffinePoint G_R() { return AffinePoint::from_hex("9d8ca4350e7e2ad27abc6d2a281365818076662962a28429590e2dc736fe9804", "ff08c30b8afd4e854623c835d9c3aac6bcebe45112472d9b9054816a7670c5a1"); }
static AffinePoint H_R() { return AffinePoint::from_hex("6d017c6f63c59fa7a5b1e9a654e27d2869579f4d152131db270558fccd27b97c", "586c43fb5c99818c564a8f80a88a65f83e3f44d3c6caf5a1a4e290b777ac56ed"); }

static PlumeSignature v1_signature() {
    return PlumeSignature{message, PK(), NULLIFIER(), NonZeroScalar::from_hex(V1_C).value(), NonZeroScalar::from_hex(V1_S).value(), PlumeSignatureV1Fields{G_R(), H_R()}};
}
static PlumeSignature v2_signature() {
    return PlumeSignature{message, PK(), NULLIFIER(), NonZeroScalar::from_hex(V2_C).value(), NonZeroScalar::from_hex(V2_S).value(), std::nullopt};
}
static void plume_v1_test() {
    const PlumeSignature sig = v1_signature();
    ASSERT(sig.pk == SecretKey::from_hex(SK).value().public_key());      // pk: (G * gen_test_scalar_sk()).into()
    const bool verified = sig.verify();
    ASSERT(verified);
}
static void plume_v2_test() { ASSERT(v2_signature().verify()); }
static void test_hash_to_curve() {
    const AffinePoint h = hash_to_secp(bytes_of("abc"));
    ASSERT_EQ_HEX(h.x(), "3377e01eab42db296b512293120c6cee72b6ecf9f9205760bd9ff11fb3cb2c4b");
    ASSERT_EQ_HEX(h.y(), "7f95890f33efebd1044d382a01b1bee0900fb6116f94688d487c6c7b9c8371f6");
}
// rust-k256/src/lib.rs:177-183
static void test_encode_pt() { ASSERT_EQ_HEX(encode_pt(AffinePoint::GENERATOR()), "0279be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798"); }

// the reference holds no `verify() == false` test: every field of the record, changed, must fail (and the V1 / V2 forms must not verify as each other)
static void verify_rejects_changed_fields() {
    PlumeSignature s = v1_signature();
    s.message[0] ^= 1; ASSERT(!s.verify());
    s = v1_signature(); s.c = NonZeroScalar::from_hex(V2_C).value(); ASSERT(!s.verify());
    s = v1_signature(); s.s = NonZeroScalar::from_hex(V2_S).value(); ASSERT(!s.verify());
    s = v1_signature(); s.nullifier = G_R(); ASSERT(!s.verify());
    s = v1_signature(); s.pk = AffinePoint::GENERATOR(); ASSERT(!s.verify());
    s = v1_signature(); std::swap(s.v1specific->r_point, s.v1specific->hashed_to_curve_r); ASSERT(!s.verify());
    s = v1_signature(); s.v1specific->r_point = AffinePoint::IDENTITY(); ASSERT(!s.verify());
    s = v1_signature(); s.v1specific.reset(); ASSERT(!s.verify());                       // V1's c under the V2 hash
    s = v2_signature(); s.v1specific = PlumeSignatureV1Fields{G_R(), H_R()}; ASSERT(!s.verify());
    s = v1_signature(); s.nullifier.xy[63] ^= 1; ASSERT(!s.verify());                   // not a curve point: no AffinePoint in the reference, false here
}
// batch twins: results item by item equal to the single calls, V1 and V2 mixed in one batch, sizes that are not a multiple of anything
static void batch_twins() {
    std::vector<PlumeSignature> sigs;
    std::vector<bool> want;
    for (int i = 0; i < 37; i++) {
        PlumeSignature s = (i % 3 == 0) ? v2_signature() : v1_signature();
        const bool bad = i % 5 == 2;
        if (bad) s.message.push_back((uint8_t)i);
        sigs.push_back(s);
        want.push_back(!bad);
    }
    ASSERT(verify_batch(sigs) == want);
    ASSERT(verify_batch({}).empty());
    // sign_batch: 9 different keys and ragged messages (empty included), nonces from the mock RNG; every signature verifies, alone and as a batch
    std::vector<SecretKey> keys;
    std::vector<Bytes> msgs;
    for (int i = 0; i < 9; i++) {
        Bytes32 k{}; k[31] = (uint8_t)(i + 1); k[0] = (uint8_t)(17 * i);
        keys.push_back(SecretKey::from_bytes(k).value());
        msgs.push_back(Bytes((size_t)(i * 13 % 40), (uint8_t)('a' + i)));
    }
    for (bool v1 : {true, false}) {
        Mock rng;
        const std::vector<PlumeSignature> out = sign_batch(keys, msgs, v1, rng);
        ASSERT(out.size() == keys.size());
        ASSERT(verify_batch(out) == std::vector<bool>(out.size(), true));
        for (size_t i = 0; i < out.size(); i++) {
            ASSERT(out[i].message == msgs[i]);
            ASSERT(out[i].pk == keys[i].public_key());
            ASSERT(out[i].v1specific.has_value() == v1);
            if (v1) ASSERT(out[i].v1specific->r_point == G_R());     // every nonce is R
        }
        // the single-call signer gives the same bytes
        Mock rng2;
        const PlumeSignature one = PlumeSigner(keys[4], v1).sign_with_rng(rng2, msgs[4]);
        ASSERT(one.c == out[4].c && one.s == out[4].s && one.nullifier == out[4].nullifier);
    }
}
// the steps either side of sign / verify: the 33-byte wire format, the aggregate pre-filter, first occurrences of nullifiers, SEC1-DER scalars
static void neighbouring_steps() {
    std::vector<SecretKey> keys;
    std::vector<Bytes> msgs;
    for (int i = 0; i < 40; i++) {
        Bytes32 k{}; k[31] = (uint8_t)(i % 7 + 1); k[5] = 9;                            // 7 distinct signers ...
        keys.push_back(SecretKey::from_bytes(k).value());
        msgs.push_back(Bytes((size_t)(1 + i % 3), (uint8_t)('m' + i % 3)));             // ... x 3 distinct messages: 21 distinct (pk, message) pairs among 40 items
    }
    Mock rng;
    std::vector<PlumeSignature> sigs = sign_batch(keys, msgs, true, rng);
    // SEC1 wire format: the same verdicts as the 64-byte form, and a record that does not decode is `false`
    Bytes pk33, nul33, c, s, r33, h33;
    for (const PlumeSignature& g : sigs) {
        auto a = sec1_record(g.pk), b = sec1_record(g.nullifier), r = sec1_record(g.v1specific->r_point), h = sec1_record(g.v1specific->hashed_to_curve_r);
        pk33.insert(pk33.end(), a.begin(), a.end()); nul33.insert(nul33.end(), b.begin(), b.end()); r33.insert(r33.end(), r.begin(), r.end()); h33.insert(h33.end(), h.begin(), h.end());
        c.insert(c.end(), g.c.to_bytes().begin(), g.c.to_bytes().end()); s.insert(s.end(), g.s.to_bytes().begin(), g.s.to_bytes().end());
    }
    ASSERT(verify_batch_sec1(true, msgs, pk33, nul33, c, s, r33, h33) == std::vector<bool>(sigs.size(), true));
    Bytes bad = nul33; bad[33 * 3] = 4;                                                  // tag 04 in a 33-byte record
    bad[33 * 5] ^= 1;                                                                    // the other root: a different (valid) point
    std::vector<bool> want(sigs.size(), true); want[3] = false; want[5] = false;
    ASSERT(verify_batch_sec1(true, msgs, pk33, bad, c, s, r33, h33) == want);
    // aggregate pre-filter: an honest batch passes, one changed item fails it
    Bytes32 seed; for (size_t i = 0; i < 32; i++) seed[i] = (uint8_t)(i * 7 + 1);
    ASSERT(aggregate_check_v1(sigs, seed));
    std::vector<PlumeSignature> forged = sigs;
    std::swap(forged[11].v1specific->hashed_to_curve_r, forged[12].v1specific->hashed_to_curve_r);   // (every nonce is R here, so the r_points are all equal)
    ASSERT(!aggregate_check_v1(forged, seed));
    ASSERT(aggregate_check_v1(sigs) && !aggregate_check_v1(forged));                               // the library draws the seed
    ASSERT(verify_batch(forged) == [&] { std::vector<bool> w(forged.size(), true); w[11] = w[12] = false; return w; }());
    // first occurrences: item i repeats the (key, message) pair of item i - 21 (7 x 3 pairs, period lcm(7, 3) = 21)
    std::vector<AffinePoint> nuls;
    for (const PlumeSignature& g : sigs) nuls.push_back(g.nullifier);
    std::vector<bool> first = nullifier_first_occurrence(nuls);
    for (size_t i = 0; i < first.size(); i++) ASSERT(first[i] == (i < 21));
    std::vector<bool> live(nuls.size(), true); live[2] = false;                         // item 2 takes no part: its repeat (item 23) becomes the first
    first = nullifier_first_occurrence(nuls, live);
    ASSERT(!first[2] && first[23] && !first[24]);
    // SEC1-DER: the record embeds scalar * G and round-trips
    const auto der = scalars_to_sec1_der({sigs[0].s, keys[3].to_nonzero_scalar()});
    ASSERT(der.size() == 2 && der[1][0] == 0x30 && der[1][1] == 0x6b);
    ASSERT(AffinePoint::from_bytes64(der[1].data() + 45) == keys[3].public_key());
    ASSERT(scalar_from_sec1_der(der[0]).value() == sigs[0].s);
    auto broken = der[0]; broken[2] ^= 1;
    ASSERT(!scalar_from_sec1_der(broken).has_value());
    // a valid scalar with SOMEBODY ELSE'S public key: the reference's from_sec1_der validates the key and returns Err
    auto foreign = der[0];
    std::copy(der[1].begin() + 45, der[1].end(), foreign.begin() + 45);
    ASSERT(!scalar_from_sec1_der(foreign).has_value());
    auto flipped = der[0]; flipped[108] ^= 1;                                                        // ... or a point that is not even on the curve
    ASSERT(!scalar_from_sec1_der(flipped).has_value());
    // the circuit's hash_to_curve inputs: y_pos is the mapped y (definition), the square-root hints are the even roots
    const auto ci = circuit_h2c_inputs(sigs[0].message, sigs[0].pk);
    ASSERT(ci.q0_y_pos == ci.q0_y_mapped && ci.q1_y_pos == ci.q1_y_mapped);
    ASSERT((ci.q0_gx1_sqrt[0] & 1) == 0 && (ci.q0_gx2_sqrt[0] & 1) == 0 && (ci.q1_gx1_sqrt[0] & 1) == 0 && (ci.q1_gx2_sqrt[0] & 1) == 0);
    ASSERT(ci.q0_x_mapped != ci.q1_x_mapped);
}
// invariants of the Rust types at the boundary of this API
static void type_invariants() {
    ASSERT(!NonZeroScalar::from_repr(Bytes32{}).has_value());                                          // zero
    ASSERT(!NonZeroScalar::from_hex("fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141").has_value());   // n
    ASSERT(NonZeroScalar::from_hex("fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364140").has_value());    // n - 1
    ASSERT(!SecretKey::from_hex("ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff").has_value());
    ASSERT(AffinePoint::IDENTITY().to_encoded_point(true) == Bytes{0});
    ASSERT(AffinePoint::GENERATOR().to_encoded_point(false).size() == 65);
}
}  // namespace verification

// ------------------------------------------------------------------------------------------- rust-arkworks/src/tests.rs
namespace arkworks {
using namespace plume_arkworks;
static Fr hex_to_fr(const char* hex) { return Fr::from_hex(hex); }                                    // tests.rs:94-106
static Bytes hardcoded_msg() { return message; }

static void test_keygen() {
    OsRng rng;
    const auto [pk, sk] = keygen(rng);
    // expected_pk = g.mul(sk): an independent route to sk * G -- the signer's comb inside plume_sign_batch (pk_in = NULL)
    const auto sks = plume_rustcrypto::SecretKey::from_bytes(sk.to_bytes_be()).value();
    Mock m;
    ASSERT(plume_rustcrypto::PlumeSignature::sign_v2(sks, Bytes{}, m).pk == pk);
}
static void test_sign_and_verify() {
    OsRng rng;
    const Bytes msg = bytes_of("Message");
    const auto keypair = keygen(rng);
    for (PlumeVersion v : {PlumeVersion::V1, PlumeVersion::V2}) {
        const Signature sig = sign(rng, {keypair.first, keypair.second}, msg, v);
        ASSERT(verify_non_zk(sig, keypair.first, msg, v));
        // a signature of one version under the other's hash; a different message; a different key
        ASSERT(!verify_non_zk(sig, keypair.first, msg, v == PlumeVersion::V1 ? PlumeVersion::V2 : PlumeVersion::V1));
        ASSERT(!verify_non_zk(sig, keypair.first, bytes_of("message"), v));
        ASSERT(!verify_non_zk(sig, Affine::GENERATOR(), msg, v));
        ASSERT_THROWS(verify_non_zk(sig, Affine::IDENTITY(), msg, v), HashToCurveError);     // Err(HashToCurveError), rust-arkworks/src/lib.rs:99-101
    }
}
static Affine hash_to_curve_with_testvalues() {                                                       // tests.rs:166-178
    const Fr sk = hex_to_fr(SK);
    const Affine pk = plume_rustcrypto::SecretKey::from_bytes(sk.to_bytes_be()).value().public_key();
    return plume_arkworks::hash_to_curve(hardcoded_msg(), pk);
}
static void test_against_zk_nullifier_sig_pk() {
    const Affine pk = plume_rustcrypto::SecretKey::from_bytes(hex_to_fr(SK).to_bytes_be()).value().public_key();
    ASSERT_EQ_HEX(pk.x(), "0cec028ee08d09e02672a68310814354f9eabfff0de6dacc1cd3a774496076ae");
    ASSERT_EQ_HEX(pk.y(), "eff471fba0409897b6a48e8801ad12f95d0009b753cf8f51c128bf6b0bd27fbd");
}
static void test_against_zk_nullifier_sig_g_r() {
    const Affine g_r = plume_rustcrypto::SecretKey::from_bytes(hex_to_fr(R).to_bytes_be()).value().public_key();
    ASSERT_EQ_HEX(g_r.x(), "9d8ca4350e7e2ad27abc6d2a281365818076662962a28429590e2dc736fe9804");
    ASSERT_EQ_HEX(g_r.y(), "ff08c30b8afd4e854623c835d9c3aac6bcebe45112472d9b9054816a7670c5a1");
}
static void test_against_zk_nullifier_sig_h() {
    const Affine h = hash_to_curve_with_testvalues();
    ASSERT_EQ_HEX(h.x(), "bcac2d0e12679f23c218889395abcdc01f2affbc49c54d1136a2190db0800b65");
    ASSERT_EQ_HEX(h.y(), "3bcfb339c974c0e757d348081f90a123b0a91a53e32b3752145d87f0cd70966e");
}
// h^r and h^sk are outputs of the signer (hashed_to_curve_r, nullifier): tests.rs:230-264
static void test_against_zk_nullifier_sig_h_r_and_h_sk() {
    const Fr sk = hex_to_fr(SK), r = hex_to_fr(R);
    const Affine pk = plume_rustcrypto::SecretKey::from_bytes(sk.to_bytes_be()).value().public_key();
    const Signature sig = sign_with_r({pk, sk}, hardcoded_msg(), r, PlumeVersion::V1);
    ASSERT_EQ_HEX(sig.second.hashed_to_curve_r.x(), "6d017c6f63c59fa7a5b1e9a654e27d2869579f4d152131db270558fccd27b97c");
    ASSERT_EQ_HEX(sig.second.hashed_to_curve_r.y(), "586c43fb5c99818c564a8f80a88a65f83e3f44d3c6caf5a1a4e290b777ac56ed");
    ASSERT_EQ_HEX(sig.first.nullifier.x(), "57bc3ed28172ef8adde4b9e0c2cce745fcc5a66473a45c1e626f1d0c67e55830");
    ASSERT_EQ_HEX(sig.first.nullifier.y(), "6a2f41488d58f33ae46edd2188e111609f9f3ae67ea38fa891d6087fe59ecb73");
}
static void test_against_zk_nullifier_sig_c_and_s() {
    const Fr r = hex_to_fr(R), sk = hex_to_fr(SK);
    const Affine pk = plume_rustcrypto::SecretKey::from_bytes(sk.to_bytes_be()).value().public_key();
    Signature sig = sign_with_r({pk, sk}, hardcoded_msg(), r, PlumeVersion::V1);
    ASSERT_EQ_HEX(sig.second.digest_private.to_bytes_be(), V1_C);
    ASSERT_EQ_HEX(sig.first.s.to_bytes_be(), V1_S);
    ASSERT(sig.first.variant == PlumeVersion::V1 && sig.second.variant == PlumeVersion::V1);
    ASSERT(verify_non_zk(sig, pk, hardcoded_msg(), PlumeVersion::V1));
    sig.second.zeroize();                                                                              // lib.rs:202-208
    ASSERT(sig.second.digest_private.is_zero() && sig.second.r_point.is_identity() && sig.second.hashed_to_curve_r.is_identity());
    sig = sign_with_r({pk, sk}, hardcoded_msg(), r, PlumeVersion::V2);
    ASSERT_EQ_HEX(sig.second.digest_private.to_bytes_be(), V2_C);
    ASSERT_EQ_HEX(sig.first.s.to_bytes_be(), V2_S);
    ASSERT(verify_non_zk(sig, pk, hardcoded_msg(), PlumeVersion::V2));
    ASSERT_THROWS(sign_with_r({Affine::IDENTITY(), sk}, hardcoded_msg(), r, PlumeVersion::V1), HashToCurveError);
}
// Fr::from_be_bytes_mod_order on the host (marshalling): values around n and a 48-byte input
static void test_fr_reduction() {
    ASSERT(Fr::from_hex("fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141").is_zero());        // n
    ASSERT_EQ_HEX(Fr::from_hex("fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364142").to_bytes_be(), "0000000000000000000000000000000000000000000000000000000000000001");
    ASSERT_EQ_HEX(Fr::from_hex("ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff").to_bytes_be(), "000000000000000000000000000000014551231950b75fc4402da1732fc9bebe");   // 2^256 - 1 - n
    // 2^383 mod n, from Python: pow(2, 383, n)
    Bytes b(48, 0); b[0] = 0x80;
    ASSERT_EQ_HEX(Fr::from_be_bytes_mod_order(b).to_bytes_be(), "a2a8918ca85bafe22016d0b997e4df5f80000000000000000000000000000000");
}
// test_point_sec1_encoding: k * G for the reference's vectors (k = 0 is the identity, `00`)
static void test_point_sec1_encoding(const char* path) {
    std::ifstream f(path);
    ASSERT(f.good());
    std::string line;
    size_t seen = 0;
    Bytes scalars, want;
    std::vector<Bytes> encs;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        unsigned long k; std::string hex;
        if (!(ss >> k >> hex)) continue;
        Bytes32 s{}; for (int i = 0; i < 8; i++) s[(size_t)(31 - i)] = (uint8_t)(k >> (8 * i));
        scalars.insert(scalars.end(), s.begin(), s.end());
        encs.push_back(plume_hip::from_hex(hex));
        seen++;
    }
    ASSERT(seen == 100);
    Bytes der(109 * seen), st(seen);
    plume_hip::check(plume_scalars_to_sec1_der_batch(plume_hip::Engine::shared().ctx(), seen, scalars.data(), der.data(), st.data()), "plume_scalars_to_sec1_der_batch");
    for (size_t i = 0; i < seen; i++) {
        const Affine point = Affine::from_bytes64(&der[109 * i + 45]);                                   // zeros (identity) for k = 0
        const auto enc = sec1_affine(point);
        const Bytes got = enc ? Bytes(enc->begin(), enc->end()) : Bytes{0};                              // `helper` (lib.rs:112-118)
        ASSERT(got == encs[i]);
    }
}
}  // namespace arkworks

int main(int argc, char** argv) {
    try {
        (void)plume_hip::Engine::shared();
    } catch (const plume_hip::Error& e) {
        std::printf("no device: %s (code %d)\n", e.what(), e.code);
        return e.code == PLUME_ERR_NODEV ? 3 : 2;
    }
    struct T { const char* name; std::function<void()> fn; };
    std::vector<T> tests = {
        {"signing::test_sign_v1", signing::test_sign_v1}, {"signing::test_sign_v2", signing::test_sign_v2}, {"signing::test_signer_type", signing::test_signer_type},
        {"verification::plume_v1_test", verification::plume_v1_test}, {"verification::plume_v2_test", verification::plume_v2_test},
        {"verification::test_hash_to_curve", verification::test_hash_to_curve}, {"lib::test_encode_pt", verification::test_encode_pt},
        {"verification::verify_rejects_changed_fields", verification::verify_rejects_changed_fields}, {"verification::batch_twins", verification::batch_twins},
        {"verification::type_invariants", verification::type_invariants}, {"verification::neighbouring_steps", verification::neighbouring_steps},
        {"arkworks::test_keygen", arkworks::test_keygen}, {"arkworks::test_sign_and_verify", arkworks::test_sign_and_verify},
        {"arkworks::test_against_zk_nullifier_sig_pk", arkworks::test_against_zk_nullifier_sig_pk}, {"arkworks::test_against_zk_nullifier_sig_g_r", arkworks::test_against_zk_nullifier_sig_g_r},
        {"arkworks::test_against_zk_nullifier_sig_h", arkworks::test_against_zk_nullifier_sig_h},
        {"arkworks::test_against_zk_nullifier_sig_h_r_and_h_sk", arkworks::test_against_zk_nullifier_sig_h_r_and_h_sk},
        {"arkworks::test_against_zk_nullifier_sig_c_and_s", arkworks::test_against_zk_nullifier_sig_c_and_s}, {"arkworks::test_fr_reduction", arkworks::test_fr_reduction},
    };
    if (argc > 1) tests.push_back({"arkworks::test_point_sec1_encoding", [argv] { arkworks::test_point_sec1_encoding(argv[1]); }});
    for (const T& t : tests) {
        const int before = fails;
        try {
            t.fn();
        } catch (const std::exception& e) {
            std::fprintf(stderr, "FAIL %s: unexpected exception: %s\n", t.name, e.what());
            fails++;
        }
        std::printf("test %s ... %s\n", t.name, fails == before ? "ok" : "FAILED");
    }
    if (fails) { std::printf("reference_tests: %d failure(s) in %d checks\n", fails, checks); return 1; }
    std::printf("reference_tests ok (%zu tests, %d checks)\n", tests.size(), checks);
    return 0;
}
