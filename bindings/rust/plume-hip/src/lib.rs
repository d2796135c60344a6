//! Batch PLUME on AMD MI355X through `libplume_hip.so` (C ABI: `include/plume_hip.h`) behind the `plume_rustcrypto` surface.
//!
//! NOT COMPILED in the build image (it has no Rust toolchain): this crate is the binding `INTEGRATION.md` describes, kept complete so that a
//! maintainer with `cargo` only has to build it.  The same FFI surface IS exercised from a non-Python caller by `tests/abi_c/abi_smoke.c`
//! (plain C, same symbols, same argument order), and from Python by `zk_nullifier_sig_amd` (ctypes).
//!
//! Shape: the reference's own types and method names (`rust-k256/src/lib.rs:67-156`, `rust-k256/src/randomizedsigner.rs:25-47`) with a batch
//! twin for every entry point.  Inside `rust-k256` the record would be `crate::PlumeSignature`; as a standalone crate it is mirrored here.
//!
//! ```ignore
//! let engine = HipEngine::new(0)?;                                  // or HipEngine::new_multi(&[0, 1, 2, 3, 4, 5, 6, 7])?
//! let oks: Vec<bool> = engine.verify_batch(&sigs)?;                 // sigs[i].verify()
//! let sigs = engine.sign_batch(&keys, &msgs, true, &mut OsRng)?;    // PlumeSigner::new(&keys[i], true).sign_with_rng(rng, msgs[i])
//! ```
use k256::elliptic_curve::sec1::{FromEncodedPoint, ToEncodedPoint};
use k256::elliptic_curve::rand_core::CryptoRngCore;
use k256::{AffinePoint, EncodedPoint, FieldBytes, NonZeroScalar, SecretKey};
use std::os::raw::{c_char, c_int, c_void};

/// `plume_rustcrypto::PlumeSignatureV1Fields` (rust-k256/src/lib.rs:84-89)
#[derive(Clone, Debug, PartialEq)]
pub struct PlumeSignatureV1Fields {
    pub r_point: AffinePoint,
    pub hashed_to_curve_r: AffinePoint,
}
/// `plume_rustcrypto::PlumeSignature` (rust-k256/src/lib.rs:67-80)
#[derive(Clone, Debug, PartialEq)]
pub struct PlumeSignature {
    pub message: Vec<u8>,
    pub pk: AffinePoint,
    pub nullifier: AffinePoint,
    pub c: NonZeroScalar,
    pub s: NonZeroScalar,
    pub v1specific: Option<PlumeSignatureV1Fields>,
}

/// What the reference signals by `panic!` / `Err` inside `try_sign_with_rng` (randomizedsigner.rs:59-61,90-95), per item of a batch.
#[derive(Clone, Debug, PartialEq, Eq)]
pub enum SignError {
    /// "something is drammatically wrong if the input hashed to the identity" (randomizedsigner.rs:61)
    HashedToIdentity,
    /// "it should be impossible to get the hash equal to zero" — the digest is 0 or >= n (randomizedsigner.rs:90-91)
    ChallengeNotCanonical,
    /// "the nonce is equal to negated product of the secret and the hash" — s == 0 (randomizedsigner.rs:95)
    ZeroResponse,
    /// a scalar outside [1, n-1] reached the library (cannot happen through `SecretKey` / `NonZeroScalar`)
    BadScalar,
}

#[derive(Debug)]
pub struct HipError(pub String);
impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "plume_hip: {}", self.0) }
}
impl std::error::Error for HipError {}

// ------------------------------------------------------------------------------------------------------ FFI (include/plume_hip.h)
#[repr(C)]
pub struct plume_ctx { _private: [u8; 0] }

pub const PLUME_STATUS_C_NOT_CANONICAL: u8 = 1;
pub const PLUME_STATUS_BAD_SCALAR: u8 = 2;
pub const PLUME_STATUS_IDENTITY: u8 = 4;

#[link(name = "plume_hip")]
extern "C" {
    fn plume_init(out: *mut *mut plume_ctx, device_id: c_int) -> c_int;
    fn plume_init_multi(out: *mut *mut plume_ctx, device_ids: *const c_int, n_devices: c_int) -> c_int;
    fn plume_num_shards(ctx: *const plume_ctx) -> c_int;
    fn plume_destroy(ctx: *mut plume_ctx);
    fn plume_last_error() -> *const c_char;
    fn plume_host_register(p: *mut c_void, bytes: usize) -> c_int;
    fn plume_host_unregister(p: *mut c_void) -> c_int;
    fn plume_verify_batch(ctx: *mut plume_ctx, version: c_int, n: usize, msgs: *const u8, msg_off: *const u64,
        pk: *const u8, nullifier: *const u8, c: *const u8, s: *const u8, r_point: *const u8, hashed_to_curve_r: *const u8, ok: *mut u8) -> c_int;
    fn plume_verify_batch_sec1(ctx: *mut plume_ctx, version: c_int, n: usize, msgs: *const u8, msg_off: *const u64,
        pk33: *const u8, nullifier33: *const u8, c: *const u8, s: *const u8, r_point33: *const u8, hashed_to_curve_r33: *const u8, ok: *mut u8) -> c_int;
    fn plume_sign_batch(ctx: *mut plume_ctx, version: c_int, n: usize, msgs: *const u8, msg_off: *const u64, sk: *const u8, r: *const u8, pk_in: *const u8,
        pk: *mut u8, nullifier: *mut u8, c: *mut u8, s: *mut u8, r_point: *mut u8, hashed_to_curve_r: *mut u8, status: *mut u8) -> c_int;
    fn plume_scalars_to_sec1_der_batch(ctx: *mut plume_ctx, n: usize, scalars: *const u8, der109: *mut u8, status: *mut u8) -> c_int;
    fn plume_sec1_der_to_scalars_checked(ctx: *mut plume_ctx, n: usize, der109: *const u8, scalars: *mut u8, ok: *mut u8) -> c_int;
    fn plume_h2c_hints_batch(ctx: *mut plume_ctx, n: usize, msgs: *const u8, msg_off: *const u64, pk: *const u8, registers: c_int, hints: *mut u8) -> c_int;
    fn plume_set_sub_batches(ctx: *mut plume_ctx, sub_batches: c_int) -> c_int;
    fn plume_set_in_flight(ctx: *mut plume_ctx, batches: c_int) -> c_int;
    fn plume_set_sign_uniform(ctx: *mut plume_ctx, level: c_int) -> c_int;
    fn plume_get_sign_uniform(ctx: *const plume_ctx) -> c_int;
    fn plume_set_host_lanes(ctx: *mut plume_ctx, lanes: c_int) -> c_int;
    fn plume_set_stage_timing(ctx: *mut plume_ctx, on: c_int) -> c_int;
    fn plume_set_eq1_short(ctx: *mut plume_ctx, mode: c_int) -> c_int;
    fn plume_get_eq1_short(ctx: *const plume_ctx, min_items: *mut usize) -> c_int;
    fn plume_last_msm_kernel(ctx: *const plume_ctx) -> *const c_char;
    fn plume_shard_numa_node(ctx: *const plume_ctx, shard: c_int) -> c_int;
    fn plume_aggregate_check(ctx: *mut plume_ctx, version: c_int, mode: c_int, n: usize, msgs: *const u8, msg_off: *const u64, pk: *const u8, nullifier: *const u8, c: *const u8,
                             s: *const u8, r_point: *const u8, hashed_to_curve_r: *const u8, seed: *const u8, hash_ok: *mut u8, result: *mut u8) -> c_int;
}

fn last_error() -> HipError { HipError(unsafe { std::ffi::CStr::from_ptr(plume_last_error()) }.to_string_lossy().into_owned()) }

/// One context (`plume_ctx`): a GPU, or several with the batch sharded by the library.  One caller thread at a time (plume_hip.h "Threading").
pub struct HipEngine(*mut plume_ctx);
unsafe impl Send for HipEngine {}
impl Drop for HipEngine { fn drop(&mut self) { unsafe { plume_destroy(self.0) } } }

/// SoA staging of a slice of signatures in the ABI's formats (64-byte affine points, all-zero = identity; 32-byte big-endian scalars)
struct Packed { msgs: Vec<u8>, off: Vec<u64>, pk: Vec<u8>, nul: Vec<u8>, c: Vec<u8>, s: Vec<u8>, rp: Vec<u8>, hr: Vec<u8>, v1: bool }

fn put_point(dst: &mut [u8], p: &AffinePoint) {
    let e = p.to_encoded_point(false);               // 04 || x || y, or 00 for the identity
    if let (Some(x), Some(y)) = (e.x(), e.y()) { dst[..32].copy_from_slice(x); dst[32..64].copy_from_slice(y); }
    // identity: the 64 bytes stay zero
}
fn get_point(src: &[u8]) -> AffinePoint {
    if src[..64].iter().all(|b| *b == 0) { return AffinePoint::IDENTITY; }
    let e = EncodedPoint::from_affine_coordinates(FieldBytes::from_slice(&src[..32]), FieldBytes::from_slice(&src[32..64]), false);
    Option::from(AffinePoint::from_encoded_point(&e)).expect("the library only emits curve points")
}
fn get_scalar(src: &[u8]) -> Option<NonZeroScalar> { Option::from(NonZeroScalar::from_repr(*FieldBytes::from_slice(&src[..32]))) }

fn pack(sigs: &[PlumeSignature]) -> Packed {
    let n = sigs.len();
    let v1 = sigs.first().map_or(false, |s| s.v1specific.is_some());
    let mut p = Packed { msgs: Vec::new(), off: Vec::with_capacity(n + 1), pk: vec![0; 64 * n], nul: vec![0; 64 * n], c: vec![0; 32 * n], s: vec![0; 32 * n],
                         rp: vec![0; if v1 { 64 * n } else { 0 }], hr: vec![0; if v1 { 64 * n } else { 0 }], v1 };
    p.off.push(0);
    for (i, sig) in sigs.iter().enumerate() {
        assert_eq!(sig.v1specific.is_some(), v1, "a batch is all V1 or all V2");
        p.msgs.extend_from_slice(&sig.message);
        p.off.push(p.msgs.len() as u64);
        put_point(&mut p.pk[64 * i..], &sig.pk);
        put_point(&mut p.nul[64 * i..], &sig.nullifier);
        p.c[32 * i..32 * i + 32].copy_from_slice(&sig.c.to_bytes());
        p.s[32 * i..32 * i + 32].copy_from_slice(&sig.s.to_bytes());
        if let Some(v) = &sig.v1specific { put_point(&mut p.rp[64 * i..], &v.r_point); put_point(&mut p.hr[64 * i..], &v.hashed_to_curve_r); }
    }
    p.msgs.push(0); // keeps the pointer non-null for n = 0 / empty messages
    p
}

impl HipEngine {
    /// `plume_init`: one GPU
    pub fn new(device: i32) -> Result<Self, HipError> {
        let mut p = std::ptr::null_mut();
        match unsafe { plume_init(&mut p, device) } { 0 => Ok(Self(p)), _ => Err(last_error()) }
    }
    /// `plume_init_multi`: the host-pointer calls split every batch evenly and contiguously over `devices` (one worker thread, streams and staging
    /// buffers per device inside the library; results land in disjoint slices; no collective)
    pub fn new_multi(devices: &[i32]) -> Result<Self, HipError> {
        let mut p = std::ptr::null_mut();
        match unsafe { plume_init_multi(&mut p, devices.as_ptr(), devices.len() as c_int) } { 0 => Ok(Self(p)), _ => Err(last_error()) }
    }
    pub fn num_shards(&self) -> usize { unsafe { plume_num_shards(self.0) as usize } }

    /// Batch twin of `PlumeSignature::verify` (rust-k256/src/lib.rs:93-145): `result[i] == sigs[i].verify()`.  All V1 or all V2.
    pub fn verify_batch(&self, sigs: &[PlumeSignature]) -> Result<Vec<bool>, HipError> {
        let p = pack(sigs);
        let mut ok = vec![0u8; sigs.len()];
        let null = std::ptr::null();
        let rc = unsafe { plume_verify_batch(self.0, if p.v1 { 1 } else { 2 }, sigs.len(), p.msgs.as_ptr(), p.off.as_ptr(), p.pk.as_ptr(), p.nul.as_ptr(), p.c.as_ptr(),
                                             p.s.as_ptr(), if p.v1 { p.rp.as_ptr() } else { null }, if p.v1 { p.hr.as_ptr() } else { null }, ok.as_mut_ptr()) };
        if rc != 0 { return Err(last_error()); }
        Ok(ok.into_iter().map(|b| b == 1).collect())
    }

    /// Aggregate pre-filter (no reference counterpart; include/plume_hip.h `plume_aggregate_check`): `Ok(true)` iff every V1 signature of the batch would
    /// `verify()` — up to a false-accept probability of 2^-126 over `seed`, which must be 32 fresh random bytes the signers could not predict.  All-or-nothing:
    /// on `Ok(false)` call `verify_batch` to find the culprits.  Per item the challenge hash is checked exactly; the two group equations (lib.rs:101,109,117,122)
    /// only in one random linear combination (a single 5n-point multi-scalar multiplication on the GPU).
    /// `seed = None` lets the library draw the 32 bytes from the OS generator (prefer it over anything constant, reused or known to the signers).
    pub fn aggregate_check_v1(&self, sigs: &[PlumeSignature], seed: Option<&[u8; 32]>) -> Result<bool, HipError> {
        let p = pack(sigs);
        assert!(p.v1 || sigs.is_empty(), "the aggregate check needs the V1 fields r_point / hashed_to_curve_r");
        let mut rec = [0u8; 72];
        let rc = unsafe { plume_aggregate_check(self.0, 1, 0, sigs.len(), p.msgs.as_ptr(), p.off.as_ptr(), p.pk.as_ptr(), p.nul.as_ptr(), p.c.as_ptr(), p.s.as_ptr(),
                                                p.rp.as_ptr(), p.hr.as_ptr(), seed.map_or(std::ptr::null(), |s| s.as_ptr()), std::ptr::null_mut(), rec.as_mut_ptr()) };
        if rc != 0 { return Err(last_error()); }
        Ok(rec[0] == 1)
    }

    /// The wire format of the serde / wasm layer (javascript/src/lib.rs:95-118,147-184): points as 33-byte SEC1-compressed records, decompressed and
    /// validated on the GPU; a record that would fail `AffinePoint::from_encoded_point` gives `false`.  Arrays are `n` records each.
    #[allow(clippy::too_many_arguments)]
    pub fn verify_batch_sec1(&self, v1: bool, msgs: &[&[u8]], pk33: &[u8], nullifier33: &[u8], c: &[u8], s: &[u8], r_point33: &[u8], hashed_to_curve_r33: &[u8])
        -> Result<Vec<bool>, HipError> {
        let n = msgs.len();
        // the library reads 33 n / 32 n bytes through these pointers: a short slice must never reach it (this is a safe fn)
        let lens_ok = pk33.len() == 33 * n && nullifier33.len() == 33 * n && c.len() == 32 * n && s.len() == 32 * n
            && (!v1 || (r_point33.len() == 33 * n && hashed_to_curve_r33.len() == 33 * n));
        if !lens_ok { return Err(HipError(format!("verify_batch_sec1: array lengths do not match {n} items (33 n bytes per point array, 32 n per scalar array)"))); }
        let (mut buf, mut off) = (Vec::new(), vec![0u64]);
        for m in msgs { buf.extend_from_slice(m); off.push(buf.len() as u64); }
        buf.push(0);
        let mut ok = vec![0u8; n];
        let null = std::ptr::null();
        let rc = unsafe { plume_verify_batch_sec1(self.0, if v1 { 1 } else { 2 }, n, buf.as_ptr(), off.as_ptr(), pk33.as_ptr(), nullifier33.as_ptr(), c.as_ptr(), s.as_ptr(),
                                                  if v1 { r_point33.as_ptr() } else { null }, if v1 { hashed_to_curve_r33.as_ptr() } else { null }, ok.as_mut_ptr()) };
        if rc != 0 { return Err(last_error()); }
        Ok(ok.into_iter().map(|b| b == 1).collect())
    }

    /// Batch twin of `PlumeSigner::new(&keys[i], v1).try_sign_with_rng(rng, msgs[i])` (randomizedsigner.rs:43-112).  The nonces are drawn on the host
    /// exactly as the reference draws them — `SecretKey::random(rng)`, one per item, in item order (randomizedsigner.rs:49) — and wiped after the call;
    /// an item on which the reference would panic comes back as `Err(SignError)` instead of unwinding.
    pub fn sign_batch(&self, keys: &[SecretKey], msgs: &[&[u8]], v1: bool, rng: &mut impl CryptoRngCore)
        -> Result<Vec<Result<PlumeSignature, SignError>>, HipError> {
        let nonces: Vec<SecretKey> = keys.iter().map(|_| SecretKey::random(rng)).collect();
        self.sign_batch_with_nonces(keys, msgs, v1, &nonces)
    }

    /// Same with the nonces supplied (the mock RNG of rust-k256/tests/signing.rs:23-44; `plume_arkworks::sign_with_r`).
    pub fn sign_batch_with_nonces(&self, keys: &[SecretKey], msgs: &[&[u8]], v1: bool, nonces: &[SecretKey])
        -> Result<Vec<Result<PlumeSignature, SignError>>, HipError> {
        let n = keys.len();
        assert!(msgs.len() == n && nonces.len() == n);
        let (mut buf, mut off) = (Vec::new(), vec![0u64]);
        for m in msgs { buf.extend_from_slice(m); off.push(buf.len() as u64); }
        buf.push(0);
        let (mut sk, mut r) = (vec![0u8; 32 * n], vec![0u8; 32 * n]);
        for i in 0..n { sk[32 * i..32 * i + 32].copy_from_slice(&keys[i].to_bytes()); r[32 * i..32 * i + 32].copy_from_slice(&nonces[i].to_bytes()); }
        let (mut pk, mut nul, mut rp, mut hr) = (vec![0u8; 64 * n], vec![0u8; 64 * n], vec![0u8; 64 * n], vec![0u8; 64 * n]);
        let (mut c, mut s, mut status) = (vec![0u8; 32 * n], vec![0u8; 32 * n], vec![0u8; n]);
        let rc = unsafe { plume_sign_batch(self.0, if v1 { 1 } else { 2 }, n, buf.as_ptr(), off.as_ptr(), sk.as_ptr(), r.as_ptr(), std::ptr::null(), pk.as_mut_ptr(),
                                           nul.as_mut_ptr(), c.as_mut_ptr(), s.as_mut_ptr(), rp.as_mut_ptr(), hr.as_mut_ptr(), status.as_mut_ptr()) };
        sk.iter_mut().for_each(|b| *b = 0);   // the reference zeroizes its secrets (SecretKey: ZeroizeOnDrop); so does the library on the device
        r.iter_mut().for_each(|b| *b = 0);
        if rc != 0 { return Err(last_error()); }
        Ok((0..n).map(|i| {
            let st = status[i];
            let nullifier = get_point(&nul[64 * i..]);
            if st & PLUME_STATUS_BAD_SCALAR != 0 { return Err(SignError::BadScalar); }
            if st & PLUME_STATUS_IDENTITY != 0 && nullifier == AffinePoint::IDENTITY { return Err(SignError::HashedToIdentity); }     // :61
            if st & PLUME_STATUS_C_NOT_CANONICAL != 0 { return Err(SignError::ChallengeNotCanonical); }                           // :91
            if st & PLUME_STATUS_IDENTITY != 0 { return Err(SignError::ZeroResponse); }                                             // :95
            Ok(PlumeSignature {
                message: msgs[i].to_vec(), pk: get_point(&pk[64 * i..]), nullifier,
                c: get_scalar(&c[32 * i..]).ok_or(SignError::ChallengeNotCanonical)?, s: get_scalar(&s[32 * i..]).ok_or(SignError::ZeroResponse)?,
                v1specific: if v1 { Some(PlumeSignatureV1Fields { r_point: get_point(&rp[64 * i..]), hashed_to_curve_r: get_point(&hr[64 * i..]) }) } else { None },
            })
        }).collect())
    }

    /// `SecretKey::from(scalar).to_sec1_der()` for a batch — the encoding the wasm wrapper uses for `s` and `digest_private`
    /// (javascript/src/lib.rs:98-110): SEC1 `ECPrivateKey { version 1, privateKey, publicKey = scalar * G (uncompressed) }`, 109 bytes.  The public-key
    /// field costs one generator multiplication per scalar: the GPU's doubling-free comb does them.
    pub fn scalars_to_sec1_der(&self, scalars: &[NonZeroScalar]) -> Result<Vec<[u8; 109]>, HipError> {
        let n = scalars.len();
        let mut flat = vec![0u8; 32 * n];
        for (i, k) in scalars.iter().enumerate() { flat[32 * i..32 * i + 32].copy_from_slice(&k.to_bytes()); }
        let (mut der, mut status) = (vec![0u8; 109 * n], vec![0u8; n]);
        let rc = unsafe { plume_scalars_to_sec1_der_batch(self.0, n, flat.as_ptr(), der.as_mut_ptr(), status.as_mut_ptr()) };
        if rc != 0 { return Err(last_error()); }
        Ok(der.chunks_exact(109).map(|ch| { let mut a = [0u8; 109]; a.copy_from_slice(ch); a }).collect())
    }

    /// `SecretKey::from_sec1_der` for the 109-byte records of `scalars_to_sec1_der`, with the reference's semantics: `None` for a record of another shape, a scalar outside
    /// [1, n-1], or an embedded public key that is not scalar * G (elliptic-curve's `TryFrom<EcPrivateKey>` validates it; here the GPU recomputes it).
    pub fn scalars_from_sec1_der(&self, der: &[[u8; 109]]) -> Result<Vec<Option<NonZeroScalar>>, HipError> {
        let n = der.len();
        let flat: Vec<u8> = der.iter().flat_map(|d| d.iter().copied()).collect();
        let (mut sc, mut ok) = (vec![0u8; 32 * n], vec![0u8; n]);
        let rc = unsafe { plume_sec1_der_to_scalars_checked(self.0, n, flat.as_ptr(), sc.as_mut_ptr(), ok.as_mut_ptr()) };
        if rc != 0 { return Err(last_error()); }
        Ok((0..n).map(|i| if ok[i] == 1 { get_scalar(&sc[32 * i..]) } else { None }).collect())
    }

    /// The circuit's square-root hints for `h2c(msg || SEC1c(pk))` (circuits/circom/verify_nullifier.circom:21-23,27-29) as 4 x 64-bit little-endian registers each:
    /// `[q0_gx1_sqrt, q0_gx2_sqrt, q0_y_pos, q1_gx1_sqrt, q1_gx2_sqrt, q1_y_pos]` per item.  UNPINNED: include/plume_hip.h defines them (the reference's generator
    /// of these inputs is not in its tree).
    pub fn h2c_hints(&self, msgs: &[&[u8]], pks: &[AffinePoint]) -> Result<Vec<[[u64; 4]; 6]>, HipError> {
        let n = msgs.len();
        assert!(pks.len() == n);
        let (mut buf, mut off) = (Vec::new(), vec![0u64]);
        for m in msgs { buf.extend_from_slice(m); off.push(buf.len() as u64); }
        buf.push(0);
        let mut pk = vec![0u8; 64 * n];
        for (i, p) in pks.iter().enumerate() { put_point(&mut pk[64 * i..], p); }
        let mut out = vec![0u8; 192 * n];
        let rc = unsafe { plume_h2c_hints_batch(self.0, n, buf.as_ptr(), off.as_ptr(), pk.as_ptr(), 1, out.as_mut_ptr()) };
        if rc != 0 { return Err(last_error()); }
        Ok(out.chunks_exact(192).map(|it| {
            let mut r = [[0u64; 4]; 6];
            for k in 0..6 { for j in 0..4 { r[k][j] = u64::from_le_bytes(it[32 * k + 8 * j..32 * k + 8 * j + 8].try_into().unwrap()); } }
            r
        }).collect())
    }

    /// Device-resident calls only (`plume_set_sub_batches`): 1 = strictly serial launch order (the default and, on the MI355X, the fastest: DESIGN.md section 6)
    pub fn set_sub_batches(&self, k: i32) -> Result<(), HipError> { if unsafe { plume_set_sub_batches(self.0, k) } == 0 { Ok(()) } else { Err(last_error()) } }
    /// Batches in flight (`plume_set_in_flight`): with 2, device-resident calls issued on different streams run side by side (two lanes of the context); default 1
    pub fn set_in_flight(&self, k: i32) -> Result<(), HipError> { if unsafe { plume_set_in_flight(self.0, k) } == 0 { Ok(()) } else { Err(last_error()) } }
    /// The verifier's first equation where `r_point` is given (`plume_set_eq1_short`): 1 = the short form for calls of at least `eq1_short()?.1` items (default), 3 = the short
    /// form whatever the size, 0 = the long form always, 2 = test mode (every item through the fallback).  Verdicts do not depend on it.
    pub fn set_eq1_short(&self, mode: i32) -> Result<(), HipError> { if unsafe { plume_set_eq1_short(self.0, mode as c_int) } == 0 { Ok(()) } else { Err(last_error()) } }
    /// `(mode, min_items)` in force (`plume_get_eq1_short`).
    pub fn eq1_short(&self) -> Result<(i32, usize), HipError> { let mut m: usize = 0; let r = unsafe { plume_get_eq1_short(self.0, &mut m) }; if r >= 0 { Ok((r as i32, m)) } else { Err(last_error()) } }
    /// Measurement hook (`plume_last_msm_kernel`): the multi-scalar kernel the last verify call on this context launched.
    pub fn last_msm_kernel(&self) -> Option<String> { let p = unsafe { plume_last_msm_kernel(self.0) }; if p.is_null() { None } else { Some(unsafe { std::ffi::CStr::from_ptr(p) }.to_string_lossy().into_owned()) } }
    /// The level this context signs at (`plume_get_sign_uniform`).
    pub fn sign_uniform(&self) -> Result<i32, HipError> { let l = unsafe { plume_get_sign_uniform(self.0) }; if l >= 0 { Ok(l as i32) } else { Err(last_error()) } }
    /// per-stage timing events inside the device pipelines: off by default (library 0.5); turn on before a call whose `last_stage_times` are wanted
    pub fn set_stage_timing(&self, on: bool) -> Result<(), HipError> { if unsafe { plume_set_stage_timing(self.0, on as c_int) } == 0 { Ok(()) } else { Err(last_error()) } }
    /// Host-pointer calls: 1 = every piece on the context itself, 2 (default) = pieces alternate between the context and a second lane.
    pub fn set_host_lanes(&self, lanes: i32) -> Result<(), HipError> { if unsafe { plume_set_host_lanes(self.0, lanes as c_int) } == 0 { Ok(()) } else { Err(last_error()) } }
    /// The signer's schedule (`plume_set_sign_uniform`).  k256's multiplication is constant-time, so the library's default is level 1 (no branch on a digit of `sk` or `r` in
    /// the kernels that walk them; table rows still gathered at digit-dependent addresses).  0 = fastest, not uniform; 2 = no secret-dependent address either (every row of a
    /// window's table is read and one kept by masked selects, as k256 does).  Outputs are unchanged at every level.
    pub fn set_sign_uniform(&self, level: i32) -> Result<(), HipError> { if unsafe { plume_set_sign_uniform(self.0, level as c_int) } == 0 { Ok(()) } else { Err(last_error()) } }
    /// The NUMA node shard `d`'s worker thread bound itself to (`None`: not bound) — allocate / pin the caller arrays of that shard's slice there
    pub fn shard_numa_node(&self, d: usize) -> Option<i32> { let v = unsafe { plume_shard_numa_node(self.0, d as c_int) }; if v >= 0 { Some(v) } else { None } }

    /// Page-lock a long-lived buffer once (`plume_host_register`): the copy engines then read / write it directly and the library's
    /// upload / compute / download pipeline overlaps fully.  Pair with `unpin`.
    pub fn pin(buf: &mut [u8]) -> Result<(), HipError> { if unsafe { plume_host_register(buf.as_mut_ptr() as *mut c_void, buf.len()) } == 0 { Ok(()) } else { Err(last_error()) } }
    pub fn unpin(buf: &mut [u8]) -> Result<(), HipError> { if unsafe { plume_host_unregister(buf.as_mut_ptr() as *mut c_void) } == 0 { Ok(()) } else { Err(last_error()) } }
}

// ------------------------------------------------------------------------------------------------ the reference's single-item surface
impl PlumeSignature {
    /// `PlumeSignature::verify` (rust-k256/src/lib.rs:93-145) on the GPU — a batch of one; use `HipEngine::verify_batch` for throughput.
    pub fn verify(&self, engine: &HipEngine) -> bool { engine.verify_batch(std::slice::from_ref(self)).map(|v| v[0]).unwrap_or(false) }
    /// `PlumeSignature::sign_v1` (rust-k256/src/lib.rs:149-151; the doc comments of sign_v1 / sign_v2 are swapped there, the behaviour is this)
    pub fn sign_v1(engine: &HipEngine, secret_key: &SecretKey, msg: &[u8], rng: &mut impl CryptoRngCore) -> Self { PlumeSigner::new(secret_key, true).sign_with_rng(engine, rng, msg) }
    /// `PlumeSignature::sign_v2` (rust-k256/src/lib.rs:154-156)
    pub fn sign_v2(engine: &HipEngine, secret_key: &SecretKey, msg: &[u8], rng: &mut impl CryptoRngCore) -> Self { PlumeSigner::new(secret_key, false).sign_with_rng(engine, rng, msg) }
}

/// `plume_rustcrypto::randomizedsigner::PlumeSigner` (randomizedsigner.rs:25-41)
pub struct PlumeSigner<'signing> {
    secret_key: &'signing SecretKey,
    pub v1: bool,
}
impl<'signing> PlumeSigner<'signing> {
    pub fn new(secret_key: &'signing SecretKey, v1: bool) -> Self { PlumeSigner { secret_key, v1 } }
    /// `RandomizedSigner::try_sign_with_rng` (randomizedsigner.rs:43-112).  `Err` only where the reference returns `signature::Error` (h2c failure,
    /// unreachable with this DST); the reference's `expect`s panic here too, with its messages.
    pub fn try_sign_with_rng(&self, engine: &HipEngine, rng: &mut impl CryptoRngCore, msg: &[u8]) -> Result<PlumeSignature, HipError> {
        let mut out = engine.sign_batch(std::slice::from_ref(self.secret_key), &[msg], self.v1, rng)?;
        match out.remove(0) {
            Ok(sig) => Ok(sig),
            Err(SignError::HashedToIdentity) => panic!("something is drammatically wrong if the input hashed to the identity"),
            Err(SignError::ChallengeNotCanonical) => panic!("it should be impossible to get the hash equal to zero"),
            Err(SignError::ZeroResponse) => panic!("something is terribly wrong if the nonce is equal to negated product of the secret and the hash"),
            Err(SignError::BadScalar) => unreachable!("SecretKey is in [1, n-1] by construction"),
        }
    }
    pub fn sign_with_rng(&self, engine: &HipEngine, rng: &mut impl CryptoRngCore, msg: &[u8]) -> PlumeSignature {
        self.try_sign_with_rng(engine, rng, msg).expect("libplume_hip call failed")
    }
}

#[cfg(test)]
mod tests {
    //! rust-k256/tests/signing.rs:9-64 against the GPU path (needs a gfx950 device and PLUME_HIP_LIB_DIR at build time)
    use super::*;
    use k256::elliptic_curve::rand_core::{CryptoRng, Error, RngCore};
    const R: [u8; 32] = hex_literal::hex!("93b9323b629f251b8f3fc2dd11f4672c5544e8230d493eceea98a90bda789808");
    const SK: [u8; 32] = hex_literal::hex!("519b423d715f8b581f4fa8ee59f4771a5b44c8130b4e3eacca54a56dda72b464");
    struct Mock;
    impl CryptoRng for Mock {}
    impl RngCore for Mock {
        fn next_u32(&mut self) -> u32 { unimplemented!() }
        fn next_u64(&mut self) -> u64 { unimplemented!() }
        fn fill_bytes(&mut self, dest: &mut [u8]) { dest.copy_from_slice(&R) }
        fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), Error> { self.fill_bytes(dest); Ok(()) }
    }
    #[test]
    fn fixed_vector() {
        let engine = HipEngine::new(0).unwrap();
        let sk = SecretKey::from_slice(&SK).unwrap();
        let v1 = PlumeSignature::sign_v1(&engine, &sk, b"An example app message string", &mut Mock);
        assert_eq!(v1.c.to_bytes().as_slice(), hex_literal::hex!("c6a7fc2c926ddbaf20731a479fb6566f2daa5514baae5223fe3b32edbce83254"));
        assert_eq!(v1.s.to_bytes().as_slice(), hex_literal::hex!("e69f027d84cb6fe5f761e333d12e975fb190d163e8ea132d7de0bd6079ba28ca"));
        let v2 = PlumeSignature::sign_v2(&engine, &sk, b"An example app message string", &mut Mock);
        assert_eq!(v2.c.to_bytes().as_slice(), hex_literal::hex!("3dbfb717705010d4f44a70720c95e74b475bd3a783ab0b9e8a6b3b363434eb96"));
        assert_eq!(v2.s.to_bytes().as_slice(), hex_literal::hex!("528e8fbb6452f82200797b1a73b2947a92524bd611085a920f1177cb8098136b"));
        assert!(v1.verify(&engine) && v2.verify(&engine));
    }
}
