// plume.hpp — C++17 host side of the MI355X PLUME engine: the reference's Rust API, name for name, above the C ABI of plume_hip.h.
//
// The reference is Rust and this image has no Rust toolchain, so the host code a caller of `plume_rustcrypto` / `plume_arkworks` would
// switch to is written here in C++ (header-only, nothing but the standard library and libplume_hip.so).  Everything cryptographic --
// hash_to_curve, the scalar multiplications, the c-hash, s = r + sk*c -- runs in the HIP kernels behind plume_hip.h; this file marshals
// records into the ABI's structure-of-arrays form and reproduces the reference's types and error behaviour:
//
//   namespace plume_rustcrypto   rust-k256/src/lib.rs:43-156, rust-k256/src/randomizedsigner.rs:25-112
//       DST, AffinePoint, NonZeroScalar, SecretKey, PlumeSignature{message, pk, nullifier, c, s, v1specific}, PlumeSignatureV1Fields,
//       PlumeSignature::verify / sign_v1 / sign_v2, PlumeSigner{secret_key, v1}::try_sign_with_rng / sign_with_rng, hash_to_curve, encode_pt
//       + batch twins: verify_batch, sign_batch, sign_batch_with_nonces;  the steps either side: verify_batch_sec1 (33-byte SEC1 records), aggregate_check_v1,
//         nullifier_first_occurrence, scalars_to_sec1_der / scalar_from_sec1_der, circuit_h2c_inputs
//   namespace plume_arkworks     rust-arkworks/src/lib.rs:60-291, rust-arkworks/src/tests.rs:28-78,119-124
//       Affine, Fr, PlumeVersion, PlumeSignaturePublic / PlumeSignaturePrivate (zeroized on drop), sec1_affine, hash_to_curve, sign_with_r, sign,
//       keygen, verify_non_zk
//   namespace plume_hip          Engine (RAII over plume_ctx, one or several GPUs), Error
//
// Error behaviour:  the signer's `expect(..)` panics (randomizedsigner.rs:61,91,95) -> plume_rustcrypto::Panic;  `signature::Error`
// (randomizedsigner.rs:59) -> plume_rustcrypto::SignatureError;  `HashToCurveError` (rust-arkworks/src/lib.rs:99-101) ->
// plume_arkworks::HashToCurveError;  a failing library call -> plume_hip::Error.  There is no CPU fallback: without a gfx950 device
// Engine's constructor throws plume_hip::Error with code PLUME_ERR_NODEV.
//
// What the Rust types guarantee by construction and this header checks on the host: scalars in [1, n-1] (NonZeroScalar, SecretKey), Fr
// reduced mod n.  What it leaves to the engine: a k256::AffinePoint can only hold a point of the curve; AffinePoint here holds 64 bytes and the
// kernels validate them (an off-curve point makes verify() false and the signer report PLUME_STATUS_BAD_SCALAR).
//
// A single sign / verify is a batch of one (a kernel launch per call: use the batch twins for throughput).
#ifndef PLUME_HPP
#define PLUME_HPP
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "plume_hip.h"

namespace plume_hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc, const char* call) {
    if (rc != PLUME_OK) throw Error(rc, std::string(call) + ": " + plume_last_error());
}

// One context: a single GPU (plume_init) or several (plume_init_multi: the host-pointer calls shard every batch evenly over the devices,
// SURVEY.md §8e).  Used by one host thread at a time; distinct engines are independent.
class Engine {
  public:
    explicit Engine(int device = 0) { check(plume_init(&ctx_, device), "plume_init"); }
    explicit Engine(const std::vector<int>& devices) { check(plume_init_multi(&ctx_, devices.data(), (int)devices.size()), "plume_init_multi"); }
    ~Engine() { if (ctx_) plume_destroy(ctx_); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;
    Engine(Engine&& o) noexcept : ctx_(o.ctx_) { o.ctx_ = nullptr; }
    plume_ctx* ctx() const { return ctx_; }
    int num_shards() const { return plume_num_shards(ctx_); }
    // The signer's schedule (plume_hip.h, plume_set_sign_uniform): 0 = fastest (its instruction trace and table addresses depend on sk and r), 1 = no branch on a digit of
    // them, 2 = and no table address derived from one -- the property k256's constant-time multiplication has (rust-k256/src/randomizedsigner.rs:51-70 multiplies by secrets).
    void set_sign_uniform(int level) { check(plume_set_sign_uniform(ctx_, level), "plume_set_sign_uniform"); }
    int sign_uniform() const { const int l = plume_get_sign_uniform(ctx_); check(l < 0 ? l : 0, "plume_get_sign_uniform"); return l; }   // 1 by default (library 0.4)
    void set_stage_timing(bool on) { check(plume_set_stage_timing(ctx_, on ? 1 : 0), "plume_set_stage_timing"); }   // off by default (library 0.5): needed before last_stage_times
    void set_host_lanes(int lanes) { check(plume_set_host_lanes(ctx_, lanes), "plume_set_host_lanes"); }
    void set_eq1_short(int mode) { check(plume_set_eq1_short(ctx_, mode), "plume_set_eq1_short"); }   // the verifier's first equation where R is given (plume_hip.h): 1 short for large calls (default), 3 short always, 0 long, 2 test
    int eq1_short(size_t* min_items = nullptr) const { const int m = plume_get_eq1_short(ctx_, min_items); check(m < 0 ? m : 0, "plume_get_eq1_short"); return m; }
    const char* last_msm_kernel() const { return plume_last_msm_kernel(ctx_); }                       // measurement hook: the multi-scalar kernel the last verify call launched
    // the engine single-item calls use when none is passed: PLUME_DEVICES="0,1,.." (one multi-device context) or device 0
    static Engine& shared() {
        static std::unique_ptr<Engine> e;
        if (!e) {
            const char* env = std::getenv("PLUME_DEVICES");
            std::vector<int> ids;
            for (const char* p = env; p && *p;) { char* end; long v = std::strtol(p, &end, 10); if (end == p) break; ids.push_back((int)v); p = (*end == ',') ? end + 1 : end; }
            e.reset(ids.size() > 1 ? new Engine(ids) : new Engine(ids.empty() ? 0 : ids[0]));
        }
        return *e;
    }

  private:
    plume_ctx* ctx_ = nullptr;
};

using Bytes = std::vector<uint8_t>;
using Bytes32 = std::array<uint8_t, 32>;
using Bytes64 = std::array<uint8_t, 64>;

// packed messages + (n+1) offsets, the ABI's message form
struct PackedMessages {
    Bytes bytes;
    std::vector<uint64_t> off{0};
    void push(const uint8_t* m, size_t len) { bytes.insert(bytes.end(), m, m + len); off.push_back(bytes.size()); }
    const uint8_t* data() const { static const uint8_t none = 0; return bytes.empty() ? &none : bytes.data(); }
};

// the order of the scalar field (rust-arkworks/src/secp256k1/fields/fr.rs:19), big-endian
inline const Bytes32& order_n() {
    static const Bytes32 n = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE,
                              0xBA, 0xAE, 0xDC, 0xE6, 0xAF, 0x48, 0xA0, 0x3B, 0xBF, 0xD2, 0x5E, 0x8C, 0xD0, 0x36, 0x41, 0x41};
    return n;
}
inline bool is_zero32(const Bytes32& b) { for (uint8_t v : b) if (v) return false; return true; }
inline bool below_n(const Bytes32& b) { return std::memcmp(b.data(), order_n().data(), 32) < 0; }
// big-endian bytes of any length, reduced mod n (Fr::from_be_bytes_mod_order): schoolbook, one byte at a time -- marshalling only, never on a hot path
inline Bytes32 reduce_mod_n(const uint8_t* be, size_t len) {
    // r = (r * 256 + byte) mod n on 5 x 64-bit limbs (r < n < 2^256, so r * 256 + byte < 2^264)
    const Bytes32& nb = order_n();
    uint64_t n[4], r[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) { n[i] = 0; for (int j = 0; j < 8; j++) n[i] = (n[i] << 8) | nb[(size_t)(24 - 8 * i + j)]; }
    for (size_t k = 0; k < len; k++) {
        for (int i = 4; i > 0; i--) r[i] = (r[i] << 8) | (r[i - 1] >> 56);
        r[0] = (r[0] << 8) | be[k];
        // r < 2^264 = 256 * 2^256 < 257 n: subtract n * 2^j for j = 8..0 while it fits
        for (int j = 8; j >= 0; j--) {
            uint64_t m[5] = {0, 0, 0, 0, 0};
            for (int i = 0; i < 4; i++) { m[i] |= n[i] << j; if (j) m[i + 1] |= n[i] >> (64 - j); }
            bool ge = true;
            for (int i = 4; i >= 0; i--) if (r[i] != m[i]) { ge = r[i] > m[i]; break; }
            if (ge) { unsigned __int128 bw = 0; for (int i = 0; i < 5; i++) { unsigned __int128 d = (unsigned __int128)r[i] - m[i] - (uint64_t)bw; r[i] = (uint64_t)d; bw = (d >> 64) & 1; } }
        }
    }
    Bytes32 out;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) out[(size_t)(24 - 8 * i + j)] = (uint8_t)(r[i] >> (56 - 8 * j));
    return out;
}

inline Bytes from_hex(const std::string& h) {
    Bytes b(h.size() / 2);
    auto nib = [](char c) -> int { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; };
    for (size_t i = 0; i < b.size(); i++) {
        const int hi = nib(h[2 * i]), lo = nib(h[2 * i + 1]);
        if (hi < 0 || lo < 0) throw std::invalid_argument("not a hex string");
        b[i] = (uint8_t)(hi * 16 + lo);
    }
    return b;
}
inline std::string to_hex(const uint8_t* b, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s(2 * n, '0');
    for (size_t i = 0; i < n; i++) { s[2 * i] = d[b[i] >> 4]; s[2 * i + 1] = d[b[i] & 15]; }
    return s;
}

}  // namespace plume_hip

// ============================================================================================== plume_rustcrypto shape
namespace plume_rustcrypto {
using plume_hip::Bytes;
using plume_hip::Bytes32;
using plume_hip::Bytes64;
using plume_hip::Engine;

// rust-k256/src/lib.rs:61
inline const std::string DST = "QUUX-V01-CS02-with-secp256k1_XMD:SHA-256_SSWU_RO_";

struct Panic : std::logic_error { using std::logic_error::logic_error; };             // the reference's expect(..) panics
struct SignatureError : std::runtime_error { SignatureError() : std::runtime_error("signature error") {} };   // signature::Error

// k256::AffinePoint as the ABI carries it: x || y big-endian, all-zero = the identity
struct AffinePoint {
    Bytes64 xy{};
    static AffinePoint IDENTITY() { return AffinePoint{}; }
    static AffinePoint GENERATOR() {   // rust-arkworks/src/secp256k1/curves/mod.rs:52-58
        return from_hex("79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798", "483ada7726a3c4655da4fbfc0e1108a8fd17b448a68554199c47d08ffb10d4b8");
    }
    static AffinePoint from_bytes64(const uint8_t* b) { AffinePoint p; std::memcpy(p.xy.data(), b, 64); return p; }
    static AffinePoint from_hex(const std::string& x, const std::string& y) {
        const Bytes bx = plume_hip::from_hex(x), by = plume_hip::from_hex(y);
        if (bx.size() != 32 || by.size() != 32) throw std::invalid_argument("coordinates are 32 bytes each");
        AffinePoint p; std::memcpy(p.xy.data(), bx.data(), 32); std::memcpy(p.xy.data() + 32, by.data(), 32); return p;
    }
    bool is_identity() const { for (uint8_t v : xy) if (v) return false; return true; }
    Bytes32 x() const { Bytes32 r; std::memcpy(r.data(), xy.data(), 32); return r; }
    Bytes32 y() const { Bytes32 r; std::memcpy(r.data(), xy.data() + 32, 32); return r; }
    // ToEncodedPoint::to_encoded_point (SEC1): identity = the single byte 00 (rust-k256/src/utils.rs:23-25, rust-arkworks/src/tests/test_vectors.rs:3-7)
    Bytes to_encoded_point(bool compress) const {
        if (is_identity()) return Bytes{0};
        Bytes out;
        if (compress) { out.push_back((uint8_t)(2 + (xy[63] & 1))); out.insert(out.end(), xy.begin(), xy.begin() + 32); }
        else { out.push_back(4); out.insert(out.end(), xy.begin(), xy.end()); }
        return out;
    }
    bool operator==(const AffinePoint& o) const { return xy == o.xy; }
    bool operator!=(const AffinePoint& o) const { return !(*this == o); }
};
// encode_pt (rust-k256/src/utils.rs:23-25)
inline Bytes encode_pt(const AffinePoint& p) { return p.to_encoded_point(true); }

// k256::NonZeroScalar: an integer of [1, n-1], 32 bytes big-endian
class NonZeroScalar {
  public:
    // NonZeroScalar::from_repr: None for 0 and for values >= n
    static std::optional<NonZeroScalar> from_repr(const Bytes32& b) {
        if (plume_hip::is_zero32(b) || !plume_hip::below_n(b)) return std::nullopt;
        return NonZeroScalar(b);
    }
    static std::optional<NonZeroScalar> from_repr(const Bytes& b) { Bytes32 a; if (b.size() != 32) return std::nullopt; std::memcpy(a.data(), b.data(), 32); return from_repr(a); }
    static std::optional<NonZeroScalar> from_hex(const std::string& h) { return from_repr(plume_hip::from_hex(h)); }
    const Bytes32& to_bytes() const { return v_; }
    std::string to_string() const { return plume_hip::to_hex(v_.data(), 32); }
    bool operator==(const NonZeroScalar& o) const { return v_ == o.v_; }
    bool operator!=(const NonZeroScalar& o) const { return v_ != o.v_; }

  protected:
    explicit NonZeroScalar(const Bytes32& b) : v_(b) {}
    Bytes32 v_;
};

// k256::SecretKey: zeroized when dropped, like the reference's
class SecretKey : public NonZeroScalar {
  public:
    // SecretKey::from_bytes / from_slice: Err for 0 and values >= n
    static std::optional<SecretKey> from_bytes(const Bytes32& b) { auto s = NonZeroScalar::from_repr(b); if (!s) return std::nullopt; return SecretKey(b); }
    static std::optional<SecretKey> from_bytes(const Bytes& b) { Bytes32 a; if (b.size() != 32) return std::nullopt; std::memcpy(a.data(), b.data(), 32); return from_bytes(a); }
    static std::optional<SecretKey> from_hex(const std::string& h) { return from_bytes(plume_hip::from_hex(h)); }
    // SecretKey::random: 32 bytes from the RNG, big-endian, rejection-sampled (pinned by the mock RNG of rust-k256/tests/signing.rs:23-44)
    template <class Rng>
    static SecretKey random(Rng& rng) {
        for (;;) {
            Bytes32 b;
            rng.fill_bytes(b.data(), b.size());
            if (auto k = from_bytes(b)) { volatile uint8_t* w = b.data(); for (size_t i = 0; i < 32; i++) w[i] = 0; return *k; }
        }
    }
    const NonZeroScalar& to_nonzero_scalar() const { return *this; }
    // SecretKey::public_key: sk * G on the GPU (the signer's comb)
    AffinePoint public_key(Engine& eng = Engine::shared()) const {
        uint8_t der[109], st = 0;
        plume_hip::check(plume_scalars_to_sec1_der_batch(eng.ctx(), 1, v_.data(), der, &st), "plume_scalars_to_sec1_der_batch");
        if (st) throw Panic("secret key outside [1, n-1]");
        return AffinePoint::from_bytes64(der + 45);   // 30 6b 02 01 01 04 20 <32> a1 44 03 42 00 04 <x> <y>
    }
    ~SecretKey() { volatile uint8_t* w = v_.data(); for (size_t i = 0; i < 32; i++) w[i] = 0; }
    SecretKey(const SecretKey&) = default;
    SecretKey& operator=(const SecretKey&) = default;

  private:
    explicit SecretKey(const Bytes32& b) : NonZeroScalar(b) {}
};

// rust-k256/src/lib.rs:84-89
struct PlumeSignatureV1Fields {
    AffinePoint r_point;
    AffinePoint hashed_to_curve_r;
};

class PlumeSigner;

// rust-k256/src/lib.rs:67-80
struct PlumeSignature {
    Bytes message;
    AffinePoint pk;
    AffinePoint nullifier;
    NonZeroScalar c;
    NonZeroScalar s;
    std::optional<PlumeSignatureV1Fields> v1specific;

    // PlumeSignature::verify (rust-k256/src/lib.rs:93-145)
    bool verify(Engine& eng = Engine::shared()) const;
    // rust-k256/src/lib.rs:149-156
    template <class Rng> static PlumeSignature sign_v1(const SecretKey& secret_key, const Bytes& msg, Rng& rng, Engine& eng = Engine::shared());
    template <class Rng> static PlumeSignature sign_v2(const SecretKey& secret_key, const Bytes& msg, Rng& rng, Engine& eng = Engine::shared());
};

// batch twin of verify(): ok[i] == sigs[i].verify().  V1 and V2 signatures may be mixed (two calls into the library).
inline std::vector<bool> verify_batch(const std::vector<PlumeSignature>& sigs, Engine& eng = Engine::shared()) {
    std::vector<bool> out(sigs.size(), false);
    for (int ver = 1; ver <= 2; ver++) {
        plume_hip::PackedMessages m;
        Bytes pk, nul, c, s, rp, hr;
        std::vector<size_t> idx;
        for (size_t i = 0; i < sigs.size(); i++) {
            const PlumeSignature& g = sigs[i];
            if ((ver == 1) != g.v1specific.has_value()) continue;
            idx.push_back(i);
            m.push(g.message.data(), g.message.size());
            pk.insert(pk.end(), g.pk.xy.begin(), g.pk.xy.end());
            nul.insert(nul.end(), g.nullifier.xy.begin(), g.nullifier.xy.end());
            c.insert(c.end(), g.c.to_bytes().begin(), g.c.to_bytes().end());
            s.insert(s.end(), g.s.to_bytes().begin(), g.s.to_bytes().end());
            if (ver == 1) {
                rp.insert(rp.end(), g.v1specific->r_point.xy.begin(), g.v1specific->r_point.xy.end());
                hr.insert(hr.end(), g.v1specific->hashed_to_curve_r.xy.begin(), g.v1specific->hashed_to_curve_r.xy.end());
            }
        }
        if (idx.empty()) continue;
        Bytes ok(idx.size());
        plume_hip::check(plume_verify_batch(eng.ctx(), ver, idx.size(), m.data(), m.off.data(), pk.data(), nul.data(), c.data(), s.data(),
                                            ver == 1 ? rp.data() : nullptr, ver == 1 ? hr.data() : nullptr, ok.data()), "plume_verify_batch");
        for (size_t k = 0; k < idx.size(); k++) out[idx[k]] = ok[k] == 1;
    }
    return out;
}
inline bool PlumeSignature::verify(Engine& eng) const { return verify_batch(std::vector<PlumeSignature>{*this}, eng)[0]; }

// rust-k256/src/randomizedsigner.rs:25-41: a borrowed secret key and the variant
class PlumeSigner {
  public:
    PlumeSigner(const SecretKey& secret_key, bool v1_) : v1(v1_), secret_key_(secret_key) {}
    bool v1;
    // RandomizedSigner::try_sign_with_rng (randomizedsigner.rs:43-112)
    template <class Rng>
    PlumeSignature try_sign_with_rng(Rng& rng, const Bytes& msg, Engine& eng = Engine::shared()) const;
    template <class Rng>
    PlumeSignature sign_with_rng(Rng& rng, const Bytes& msg, Engine& eng = Engine::shared()) const { return try_sign_with_rng(rng, msg, eng); }

  private:
    const SecretKey& secret_key_;
};

// status byte of the signer -> the reference's panics, in the order the reference would hit them (randomizedsigner.rs:61,91,95)
inline void raise_for_status(uint8_t st, const AffinePoint& nullifier) {
    if (st & PLUME_STATUS_BAD_SCALAR) throw Panic("secret key or nonce outside [1, n-1] (no SecretKey / NonZeroScalar holds it)");
    if ((st & PLUME_STATUS_IDENTITY) && nullifier.is_identity()) throw Panic("something is drammatically wrong if the input hashed to the identity");
    if (st & PLUME_STATUS_C_NOT_CANONICAL) throw Panic("it should be impossible to get the hash equal to zero");
    if (st & PLUME_STATUS_IDENTITY) throw Panic("something is terribly wrong if the nonce is equal to negated product of the secret and the hash");
}

// batch twin of the signer with the nonces supplied (the RNG stays on the host): one signature per (key, message, nonce).  Throws Panic at the first
// item the reference's signer would panic on.
inline std::vector<PlumeSignature> sign_batch_with_nonces(const std::vector<SecretKey>& keys, const std::vector<Bytes>& msgs, bool v1, const std::vector<SecretKey>& nonces,
                                                          Engine& eng = Engine::shared()) {
    const size_t n = keys.size();
    if (msgs.size() != n || nonces.size() != n) throw std::invalid_argument("keys, msgs and nonces must have one entry per signature");
    plume_hip::PackedMessages m;
    Bytes sk(32 * n), r(32 * n), pk(64 * n), nul(64 * n), c(32 * n), s(32 * n), rp(64 * n), hr(64 * n), st(n);
    for (size_t i = 0; i < n; i++) {
        m.push(msgs[i].data(), msgs[i].size());
        std::memcpy(&sk[32 * i], keys[i].to_bytes().data(), 32);
        std::memcpy(&r[32 * i], nonces[i].to_bytes().data(), 32);
    }
    std::vector<PlumeSignature> out;
    if (n == 0) return out;
    const int rc = plume_sign_batch(eng.ctx(), v1 ? 1 : 2, n, m.data(), m.off.data(), sk.data(), r.data(), nullptr, pk.data(), nul.data(), c.data(), s.data(), rp.data(), hr.data(), st.data());
    { volatile uint8_t* w = sk.data(); for (size_t i = 0; i < sk.size(); i++) w[i] = 0; }
    { volatile uint8_t* w = r.data(); for (size_t i = 0; i < r.size(); i++) w[i] = 0; }
    plume_hip::check(rc, "plume_sign_batch");
    out.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const AffinePoint nl = AffinePoint::from_bytes64(&nul[64 * i]);
        raise_for_status(st[i], nl);
        Bytes32 cb, sb;
        std::memcpy(cb.data(), &c[32 * i], 32);
        std::memcpy(sb.data(), &s[32 * i], 32);
        auto cs = NonZeroScalar::from_repr(cb), ss = NonZeroScalar::from_repr(sb);
        if (!cs) throw Panic("it should be impossible to get the hash equal to zero");
        if (!ss) throw Panic("something is terribly wrong if the nonce is equal to negated product of the secret and the hash");
        PlumeSignature g{msgs[i], AffinePoint::from_bytes64(&pk[64 * i]), nl, *cs, *ss, std::nullopt};
        if (v1) g.v1specific = PlumeSignatureV1Fields{AffinePoint::from_bytes64(&rp[64 * i]), AffinePoint::from_bytes64(&hr[64 * i])};
        out.push_back(std::move(g));
    }
    return out;
}
// ... and with the nonces drawn as the reference draws them: SecretKey::random(rng) per signature, in order (randomizedsigner.rs:49)
template <class Rng>
std::vector<PlumeSignature> sign_batch(const std::vector<SecretKey>& keys, const std::vector<Bytes>& msgs, bool v1, Rng& rng, Engine& eng = Engine::shared()) {
    std::vector<SecretKey> nonces;
    nonces.reserve(keys.size());
    for (size_t i = 0; i < keys.size(); i++) nonces.push_back(SecretKey::random(rng));
    return sign_batch_with_nonces(keys, msgs, v1, nonces, eng);
}

template <class Rng>
PlumeSignature PlumeSigner::try_sign_with_rng(Rng& rng, const Bytes& msg, Engine& eng) const {
    const SecretKey r_scalar = SecretKey::random(rng);                                         // randomizedsigner.rs:49
    return std::move(sign_batch_with_nonces({secret_key_}, {msg}, v1, {r_scalar}, eng)[0]);
}
template <class Rng>
PlumeSignature PlumeSignature::sign_v1(const SecretKey& secret_key, const Bytes& msg, Rng& rng, Engine& eng) { return PlumeSigner(secret_key, true).sign_with_rng(rng, msg, eng); }
template <class Rng>
PlumeSignature PlumeSignature::sign_v2(const SecretKey& secret_key, const Bytes& msg, Rng& rng, Engine& eng) { return PlumeSigner(secret_key, false).sign_with_rng(rng, msg, eng); }

// ---- the steps either side of sign / verify, as the Rust binding has them (bindings/rust/plume-hip) ----
// The serde / wasm wire format (javascript/src/lib.rs:95-118,147-184): points as 33-byte SEC1-compressed records (02|03 || x; a record starting with 00 is the
// identity), decompressed and validated on the GPU; a record that would fail AffinePoint::from_encoded_point gives false.  Arrays hold n records each; r_point33 /
// hashed_to_curve_r33 are used for v1 only.
inline std::vector<bool> verify_batch_sec1(bool v1, const std::vector<Bytes>& msgs, const Bytes& pk33, const Bytes& nullifier33, const Bytes& c, const Bytes& s, const Bytes& r_point33,
                                           const Bytes& hashed_to_curve_r33, Engine& eng = Engine::shared()) {
    const size_t n = msgs.size();
    if (pk33.size() != 33 * n || nullifier33.size() != 33 * n || c.size() != 32 * n || s.size() != 32 * n || (v1 && (r_point33.size() != 33 * n || hashed_to_curve_r33.size() != 33 * n)))
        throw std::invalid_argument("verify_batch_sec1: arrays must hold one record per message");
    plume_hip::PackedMessages m;
    for (const Bytes& x : msgs) m.push(x.data(), x.size());
    Bytes ok(n);
    if (n) plume_hip::check(plume_verify_batch_sec1(eng.ctx(), v1 ? 1 : 2, n, m.data(), m.off.data(), pk33.data(), nullifier33.data(), c.data(), s.data(), v1 ? r_point33.data() : nullptr,
                                                    v1 ? hashed_to_curve_r33.data() : nullptr, ok.data()), "plume_verify_batch_sec1");
    return std::vector<bool>(ok.begin(), ok.end());
}
// AffinePoint::to_encoded_point(true) as the fixed 33-byte record of that wire format (identity: 00 followed by zeros)
inline std::array<uint8_t, 33> sec1_record(const AffinePoint& p) {
    std::array<uint8_t, 33> r{};
    if (!p.is_identity()) { r[0] = (uint8_t)(2 + (p.xy[63] & 1)); std::memcpy(r.data() + 1, p.xy.data(), 32); }
    return r;
}
// Aggregate pre-filter (no reference counterpart; plume_hip.h plume_aggregate_check): true iff every V1 signature of the batch would verify() -- up to a false-accept
// probability of 2^-126 over `seed`, 32 fresh random bytes the signers could not predict.  All-or-nothing: on false, verify_batch finds the culprits.
// `seed == nullptr`: the library draws the 32 bytes from the OS generator for this call -- what a caller without fresh randomness of its own should do.
inline bool aggregate_check_v1(const std::vector<PlumeSignature>& sigs, const Bytes32* seed, Engine& eng = Engine::shared()) {
    plume_hip::PackedMessages m;
    Bytes pk, nul, c, s, rp, hr;
    for (const PlumeSignature& g : sigs) {
        if (!g.v1specific) throw std::invalid_argument("the aggregate check needs the V1 fields r_point / hashed_to_curve_r");
        m.push(g.message.data(), g.message.size());
        pk.insert(pk.end(), g.pk.xy.begin(), g.pk.xy.end());
        nul.insert(nul.end(), g.nullifier.xy.begin(), g.nullifier.xy.end());
        c.insert(c.end(), g.c.to_bytes().begin(), g.c.to_bytes().end());
        s.insert(s.end(), g.s.to_bytes().begin(), g.s.to_bytes().end());
        rp.insert(rp.end(), g.v1specific->r_point.xy.begin(), g.v1specific->r_point.xy.end());
        hr.insert(hr.end(), g.v1specific->hashed_to_curve_r.xy.begin(), g.v1specific->hashed_to_curve_r.xy.end());
    }
    uint8_t rec[PLUME_AGG_RESULT_BYTES];
    static const uint8_t none[64] = {0};
    const bool e = sigs.empty();
    plume_hip::check(plume_aggregate_check(eng.ctx(), 1, 0, sigs.size(), m.data(), m.off.data(), e ? none : pk.data(), e ? none : nul.data(), e ? none : c.data(), e ? none : s.data(),
                                           e ? none : rp.data(), e ? none : hr.data(), seed ? seed->data() : nullptr, nullptr, rec), "plume_aggregate_check");
    return rec[0] == 1;
}
inline bool aggregate_check_v1(const std::vector<PlumeSignature>& sigs, const Bytes32& seed, Engine& eng = Engine::shared()) { return aggregate_check_v1(sigs, &seed, eng); }
inline bool aggregate_check_v1(const std::vector<PlumeSignature>& sigs, Engine& eng = Engine::shared()) { return aggregate_check_v1(sigs, static_cast<const Bytes32*>(nullptr), eng); }
// The application step after verification (reference README.md:5: one nullifier per (pk, message)): first[i] is true iff item i is live and no live item with the
// same nullifier comes before it.  `live` may be empty (all items take part), e.g. pass verify_batch's result.
inline std::vector<bool> nullifier_first_occurrence(const std::vector<AffinePoint>& nullifiers, const std::vector<bool>& live = {}, Engine& eng = Engine::shared()) {
    const size_t n = nullifiers.size();
    if (!live.empty() && live.size() != n) throw std::invalid_argument("live must be empty or hold one flag per nullifier");
    Bytes nul(64 * n), lv(live.begin(), live.end()), first(n);
    for (size_t i = 0; i < n; i++) std::memcpy(&nul[64 * i], nullifiers[i].xy.data(), 64);
    if (n) plume_hip::check(plume_nullifier_first_occurrence(eng.ctx(), n, nul.data(), live.empty() ? nullptr : lv.data(), nullptr, first.data(), nullptr), "plume_nullifier_first_occurrence");
    return std::vector<bool>(first.begin(), first.end());
}
// SecretKey::from(scalar).to_sec1_der() for a batch -- the encoding the wasm wrapper uses for `s` and `digest_private` (javascript/src/lib.rs:98-110): the 109-byte
// SEC1 ECPrivateKey with publicKey = scalar * G, the generator multiplications on the GPU
inline std::vector<std::array<uint8_t, 109>> scalars_to_sec1_der(const std::vector<NonZeroScalar>& scalars, Engine& eng = Engine::shared()) {
    const size_t n = scalars.size();
    Bytes flat(32 * n), der(109 * n), st(n);
    for (size_t i = 0; i < n; i++) std::memcpy(&flat[32 * i], scalars[i].to_bytes().data(), 32);
    if (n) plume_hip::check(plume_scalars_to_sec1_der_batch(eng.ctx(), n, flat.data(), der.data(), st.data()), "plume_scalars_to_sec1_der_batch");
    std::vector<std::array<uint8_t, 109>> out(n);
    for (size_t i = 0; i < n; i++) std::memcpy(out[i].data(), &der[109 * i], 109);
    return out;
}
// SecretKey::from_sec1_der for that fixed form: nullopt where the reference returns Err -- a record of another shape, a scalar outside [1, n-1], or an embedded
// public key that is not scalar * G (elliptic-curve's TryFrom<EcPrivateKey> validates it; here the GPU recomputes it)
inline std::optional<NonZeroScalar> scalar_from_sec1_der(const std::array<uint8_t, 109>& der, Engine& eng = Engine::shared()) {
    Bytes32 k; uint8_t ok = 0;
    plume_hip::check(plume_sec1_der_to_scalars_checked(eng.ctx(), 1, der.data(), k.data(), &ok), "plume_sec1_der_to_scalars_checked");
    if (!ok) return std::nullopt;
    return NonZeroScalar::from_repr(k);
}

// hash_to_curve(m, pk) (rust-k256/src/utils.rs:11-20): RFC 9380 hash_to_curve over m || SEC1c(pk)
inline AffinePoint hash_to_curve(const Bytes& m, const AffinePoint& pk, Engine& eng = Engine::shared()) {
    plume_hip::PackedMessages pm;
    pm.push(m.data(), m.size());
    AffinePoint h;
    plume_hip::check(plume_hash_to_curve_batch(eng.ctx(), 1, pm.data(), pm.off.data(), pk.xy.data(), h.xy.data()), "plume_hash_to_curve_batch");
    return h;
}
// Secp256k1::hash_from_bytes::<ExpandMsgXmd<Sha256>>(&[s], &[DST]) over the raw bytes (rust-k256/tests/verification.rs:148-156 `hash_to_secp`)
inline AffinePoint hash_to_secp(const Bytes& s, Engine& eng = Engine::shared()) {
    plume_hip::PackedMessages pm;
    pm.push(s.data(), s.size());
    AffinePoint h;
    plume_hip::check(plume_hash_to_curve_batch(eng.ctx(), 1, pm.data(), pm.off.data(), nullptr, h.xy.data()), "plume_hash_to_curve_batch");
    return h;
}

// The circom verifier's hash_to_curve inputs for one signature (circuits/circom/verify_nullifier.circom:14-31), every value as the circuit's four 64-bit little-endian
// registers (circuits/circom/utils.ts:11-17): q{0,1}_x_mapped / q{0,1}_y_mapped from plume_h2c_intermediates_batch (pinned to the reference's vectors) and
// q{0,1}_gx1_sqrt / gx2_sqrt / y_pos from plume_h2c_hints_batch (UNPINNED definitions: include/plume_hip.h).
struct CircuitH2cInputs {
    using Reg = std::array<uint64_t, 4>;
    Reg q0_gx1_sqrt, q0_gx2_sqrt, q0_y_pos, q0_x_mapped, q0_y_mapped, q1_gx1_sqrt, q1_gx2_sqrt, q1_y_pos, q1_x_mapped, q1_y_mapped;
};
inline CircuitH2cInputs circuit_h2c_inputs(const Bytes& m, const AffinePoint& pk, Engine& eng = Engine::shared()) {
    plume_hip::PackedMessages pm;
    pm.push(m.data(), m.size());
    uint64_t hints[24], mapped[16];
    plume_hip::check(plume_h2c_hints_batch(eng.ctx(), 1, pm.data(), pm.off.data(), pk.xy.data(), 1, reinterpret_cast<uint8_t*>(hints)), "plume_h2c_hints_batch");
    plume_hip::check(plume_h2c_intermediates_batch(eng.ctx(), 1, pm.data(), pm.off.data(), pk.xy.data(), 1, nullptr, reinterpret_cast<uint8_t*>(mapped), nullptr, nullptr),
                     "plume_h2c_intermediates_batch");
    auto reg = [](const uint64_t* p) { return CircuitH2cInputs::Reg{p[0], p[1], p[2], p[3]}; };
    CircuitH2cInputs o;
    o.q0_gx1_sqrt = reg(hints); o.q0_gx2_sqrt = reg(hints + 4); o.q0_y_pos = reg(hints + 8);
    o.q1_gx1_sqrt = reg(hints + 12); o.q1_gx2_sqrt = reg(hints + 16); o.q1_y_pos = reg(hints + 20);
    o.q0_x_mapped = reg(mapped); o.q0_y_mapped = reg(mapped + 4); o.q1_x_mapped = reg(mapped + 8); o.q1_y_mapped = reg(mapped + 12);
    return o;
}

}  // namespace plume_rustcrypto

// ================================================================================================ plume_arkworks shape
namespace plume_arkworks {
using plume_hip::Bytes;
using plume_hip::Bytes32;
using plume_hip::Engine;
using Affine = plume_rustcrypto::AffinePoint;          // secp256k1::Affine

struct HashToCurveError : std::runtime_error { using std::runtime_error::runtime_error; };

// secp256k1::Fr: an element of the scalar field, reduced, zero included
class Fr {
  public:
    Fr() : v_{} {}
    static Fr from_be_bytes_mod_order(const uint8_t* b, size_t len) { Fr f; f.v_ = plume_hip::reduce_mod_n(b, len); return f; }
    static Fr from_be_bytes_mod_order(const Bytes& b) { return from_be_bytes_mod_order(b.data(), b.size()); }
    static Fr from_hex(const std::string& h) { return from_be_bytes_mod_order(plume_hip::from_hex(h)); }   // tests.rs:94-106 hex_to_fr
    // Fr::rand: 48 random bytes reduced (bias < 2^-128), the width hash_to_field uses
    template <class Rng> static Fr rand(Rng& rng) { uint8_t b[48]; rng.fill_bytes(b, sizeof b); Fr f = from_be_bytes_mod_order(b, sizeof b); volatile uint8_t* w = b; for (size_t i = 0; i < sizeof b; i++) w[i] = 0; return f; }
    const Bytes32& to_bytes_be() const { return v_; }
    bool is_zero() const { return plume_hip::is_zero32(v_); }
    void zeroize() { volatile uint8_t* w = v_.data(); for (size_t i = 0; i < 32; i++) w[i] = 0; }
    bool operator==(const Fr& o) const { return v_ == o.v_; }
    bool operator!=(const Fr& o) const { return v_ != o.v_; }

  private:
    Bytes32 v_;
};

enum class PlumeVersion { V1 = 1, V2 = 2 };             // rust-arkworks/src/lib.rs:66-69
using PublicKey = Affine;                                // lib.rs:216
using SecretKeyMaterial = Fr;                            // lib.rs:218

// sec1_affine (rust-arkworks/src/lib.rs:76-88): None for the identity
inline std::optional<std::array<uint8_t, 33>> sec1_affine(const Affine& a) {
    if (a.is_identity()) return std::nullopt;
    std::array<uint8_t, 33> w;
    w[0] = (uint8_t)(2 + (a.xy[63] & 1));
    std::memcpy(w.data() + 1, a.xy.data(), 32);
    return w;
}

// rust-arkworks/src/lib.rs:185-191
struct PlumeSignaturePublic {
    Bytes message;
    Fr s;
    Affine nullifier;
    std::optional<PlumeVersion> variant;
};
// rust-arkworks/src/lib.rs:194-214: the witness; zeroized on drop like the reference's
struct PlumeSignaturePrivate {
    Affine hashed_to_curve_r;
    Affine r_point;
    Fr digest_private;
    PlumeVersion variant;
    void zeroize() { digest_private.zeroize(); hashed_to_curve_r = Affine{}; r_point = Affine{}; }
    ~PlumeSignaturePrivate() { zeroize(); }
    PlumeSignaturePrivate(const Affine& hr, const Affine& rp, const Fr& d, PlumeVersion v) : hashed_to_curve_r(hr), r_point(rp), digest_private(d), variant(v) {}
    PlumeSignaturePrivate(const PlumeSignaturePrivate&) = default;
    PlumeSignaturePrivate& operator=(const PlumeSignaturePrivate&) = default;
};
using Signature = std::pair<PlumeSignaturePublic, PlumeSignaturePrivate>;

// hash_to_curve(message, pk) (rust-arkworks/src/lib.rs:90-106): Err for pk = identity
inline Affine hash_to_curve(const Bytes& message, const Affine& pk, Engine& eng = Engine::shared()) {
    if (pk.is_identity()) throw HashToCurveError("`pk` shouldn't be the identity element");
    return plume_rustcrypto::hash_to_curve(message, pk, eng);
}

// keygen (rust-arkworks/src/tests.rs:119-124): sk = Fr::rand, pk = sk * G (on the GPU)
template <class Rng>
std::pair<PublicKey, SecretKeyMaterial> keygen(Rng& rng, Engine& eng = Engine::shared()) {
    for (;;) {
        const Fr sk = Fr::rand(rng);
        if (sk.is_zero()) continue;
        uint8_t der[109], st = 0;
        plume_hip::check(plume_scalars_to_sec1_der_batch(eng.ctx(), 1, sk.to_bytes_be().data(), der, &st), "plume_scalars_to_sec1_der_batch");
        return {Affine::from_bytes64(der + 45), sk};
    }
}

// batch twin of sign_with_r: signature i from (pk_i, sk_i), message i, r_i.  pk is supplied, not recomputed; c is reduced mod n and never rejected.
inline std::vector<Signature> sign_with_r_batch(const std::vector<std::pair<PublicKey, SecretKeyMaterial>>& keypairs, const std::vector<Bytes>& messages, const std::vector<Fr>& r_scalars,
                                                PlumeVersion version, Engine& eng = Engine::shared()) {
    const size_t n = keypairs.size();
    if (messages.size() != n || r_scalars.size() != n) throw std::invalid_argument("keypairs, messages and r_scalars must have one entry per signature");
    plume_hip::PackedMessages m;
    Bytes sk(32 * n), r(32 * n), pkin(64 * n), nul(64 * n), c(32 * n), s(32 * n), rp(64 * n), hr(64 * n), st(n);
    for (size_t i = 0; i < n; i++) {
        if (keypairs[i].first.is_identity()) throw HashToCurveError("`pk` shouldn't be the identity element");   // lib.rs:99-101
        m.push(messages[i].data(), messages[i].size());
        std::memcpy(&pkin[64 * i], keypairs[i].first.xy.data(), 64);
        std::memcpy(&sk[32 * i], keypairs[i].second.to_bytes_be().data(), 32);
        std::memcpy(&r[32 * i], r_scalars[i].to_bytes_be().data(), 32);
    }
    std::vector<Signature> out;
    if (n == 0) return out;
    const int rc = plume_sign_batch(eng.ctx(), (int)version, n, m.data(), m.off.data(), sk.data(), r.data(), pkin.data(), nullptr, nul.data(), c.data(), s.data(), rp.data(), hr.data(), st.data());
    { volatile uint8_t* w = sk.data(); for (size_t i = 0; i < sk.size(); i++) w[i] = 0; }
    { volatile uint8_t* w = r.data(); for (size_t i = 0; i < r.size(); i++) w[i] = 0; }
    plume_hip::check(rc, "plume_sign_batch");
    out.reserve(n);
    for (size_t i = 0; i < n; i++) {
        // Fr holds zero, so a zero sk / r is a value here; the engine's k256-shaped status bits are not errors of this API, except an off-curve pk
        if ((st[i] & PLUME_STATUS_BAD_SCALAR) && !keypairs[i].second.is_zero() && !r_scalars[i].is_zero()) throw HashToCurveError("`pk` is not a point of the curve");
        out.emplace_back(PlumeSignaturePublic{messages[i], Fr::from_be_bytes_mod_order(&s[32 * i], 32), Affine::from_bytes64(&nul[64 * i]), version},
                         PlumeSignaturePrivate(Affine::from_bytes64(&hr[64 * i]), Affine::from_bytes64(&rp[64 * i]), Fr::from_be_bytes_mod_order(&c[32 * i], 32), version));
    }
    return out;
}
// sign_with_r (rust-arkworks/src/lib.rs:229-278)
inline Signature sign_with_r(const std::pair<const PublicKey&, const SecretKeyMaterial&>& keypair, const Bytes& message, const Fr& r_scalar, PlumeVersion version,
                             Engine& eng = Engine::shared()) {
    return std::move(sign_with_r_batch({{keypair.first, keypair.second}}, {message}, {r_scalar}, version, eng)[0]);
}
// sign (rust-arkworks/src/lib.rs:281-291): r = Fr::rand(rng)
template <class Rng>
Signature sign(Rng& rng, const std::pair<const PublicKey&, const SecretKeyMaterial&>& keypair, const Bytes& message, PlumeVersion version, Engine& eng = Engine::shared()) {
    return sign_with_r(keypair, message, Fr::rand(rng), version, eng);
}

// verify_non_zk (rust-arkworks/src/tests.rs:28-78), batched: c' from the GIVEN r_point / hashed_to_curve_r, both equations for V1 and V2, c' == digest_private.
// Result i: 1 = Ok(true), 0 = Ok(false), 2 = Err(HashToCurveError).
inline Bytes verify_non_zk_batch(const std::vector<Signature>& sigs, const std::vector<PublicKey>& pks, const std::vector<Bytes>& messages, PlumeVersion version,
                                 Engine& eng = Engine::shared()) {
    const size_t n = sigs.size();
    if (pks.size() != n || messages.size() != n) throw std::invalid_argument("sigs, pks and messages must have one entry per signature");
    plume_hip::PackedMessages m;
    Bytes pk(64 * n), nul(64 * n), s(32 * n), rp(64 * n), hr(64 * n), d(32 * n), ok(n);
    for (size_t i = 0; i < n; i++) {
        m.push(messages[i].data(), messages[i].size());
        std::memcpy(&pk[64 * i], pks[i].xy.data(), 64);
        std::memcpy(&nul[64 * i], sigs[i].first.nullifier.xy.data(), 64);
        std::memcpy(&s[32 * i], sigs[i].first.s.to_bytes_be().data(), 32);
        std::memcpy(&rp[64 * i], sigs[i].second.r_point.xy.data(), 64);
        std::memcpy(&hr[64 * i], sigs[i].second.hashed_to_curve_r.xy.data(), 64);
        std::memcpy(&d[32 * i], sigs[i].second.digest_private.to_bytes_be().data(), 32);
    }
    if (n) plume_hip::check(plume_verify_non_zk_batch(eng.ctx(), (int)version, n, m.data(), m.off.data(), pk.data(), nul.data(), s.data(), rp.data(), hr.data(), d.data(), ok.data()),
                            "plume_verify_non_zk_batch");
    return ok;
}
// the reference's signature takes `pp` (Parameters{g_point}); the generator is fixed in the engine
inline bool verify_non_zk(const Signature& sig, const PublicKey& pk, const Bytes& message, PlumeVersion version, Engine& eng = Engine::shared()) {
    const uint8_t r = verify_non_zk_batch({sig}, {pk}, {message}, version, eng)[0];
    if (r == 2) throw HashToCurveError("`pk` shouldn't be the identity element");
    return r == 1;
}

}  // namespace plume_arkworks
#endif  // PLUME_HPP
