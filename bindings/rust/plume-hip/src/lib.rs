//! Batch PLUME on AMD MI355X through libplume_hip.so (C ABI: include/plume_hip.h).
// UNTESTED (no Rust toolchain in the build image).  In-tree use inside rust-k256 would import crate::{..}; as a standalone crate the
// signature record is mirrored here with the reference's field names (rust-k256/src/lib.rs:67-89).
use k256::{AffinePoint, NonZeroScalar};
pub struct PlumeSignatureV1Fields { pub r_point: AffinePoint, pub hashed_to_curve_r: AffinePoint }
pub struct PlumeSignature { pub message: Vec<u8>, pub pk: AffinePoint, pub nullifier: AffinePoint, pub c: NonZeroScalar, pub s: NonZeroScalar, pub v1specific: Option<PlumeSignatureV1Fields> }
use k256::elliptic_curve::sec1::ToEncodedPoint;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct plume_ctx { _private: [u8; 0] }

#[link(name = "plume_hip")]
extern "C" {
    fn plume_init(out: *mut *mut plume_ctx, device_id: c_int) -> c_int;
    fn plume_destroy(ctx: *mut plume_ctx);
    fn plume_last_error() -> *const c_char;
    fn plume_verify_batch(ctx: *mut plume_ctx, version: c_int, n: usize,
        msgs: *const u8, msg_off: *const u64,
        pk: *const u8, nullifier: *const u8, c: *const u8, s: *const u8,
        r_point: *const u8, hashed_to_curve_r: *const u8, ok: *mut u8) -> c_int;
    fn plume_sign_batch(ctx: *mut plume_ctx, version: c_int, n: usize,
        msgs: *const u8, msg_off: *const u64, sk: *const u8, r: *const u8, pk_in: *const u8,
        pk: *mut u8, nullifier: *mut u8, c: *mut u8, s: *mut u8,
        r_point: *mut u8, hashed_to_curve_r: *mut u8, status: *mut u8) -> c_int;
}

pub struct HipEngine(*mut plume_ctx);
impl HipEngine {
    pub fn new(device: i32) -> Result<Self, String> {
        let mut p = std::ptr::null_mut();
        match unsafe { plume_init(&mut p, device) } { 0 => Ok(Self(p)), _ => Err(last_error()) }
    }
    /// `PlumeSignature::verify` for a homogeneous (all V1 or all V2) slice; result[i] == sigs[i].verify()
    pub fn verify_batch(&self, sigs: &[PlumeSignature]) -> Result<Vec<bool>, String> {
        let n = sigs.len();
        let v1 = sigs.first().map_or(false, |s| s.v1specific.is_some());
        let (mut msgs, mut off) = (Vec::new(), Vec::with_capacity(n + 1));
        let (mut pk, mut nul, mut c, mut s) = (vec![0u8; 64 * n], vec![0u8; 64 * n], vec![0u8; 32 * n], vec![0u8; 32 * n]);
        let (mut rp, mut hr) = (vec![0u8; if v1 { 64 * n } else { 0 }], vec![0u8; if v1 { 64 * n } else { 0 }]);
        off.push(0u64);
        for (i, sig) in sigs.iter().enumerate() {
            assert_eq!(sig.v1specific.is_some(), v1, "mixed V1/V2 batch");
            msgs.extend_from_slice(&sig.message); off.push(msgs.len() as u64);
            put_point(&mut pk[64 * i..], &sig.pk); put_point(&mut nul[64 * i..], &sig.nullifier);
            c[32 * i..32 * i + 32].copy_from_slice(&sig.c.to_bytes()); s[32 * i..32 * i + 32].copy_from_slice(&sig.s.to_bytes());
            if let Some(PlumeSignatureV1Fields { r_point, hashed_to_curve_r }) = &sig.v1specific {
                put_point(&mut rp[64 * i..], r_point); put_point(&mut hr[64 * i..], hashed_to_curve_r);
            }
        }
        msgs.push(0); // keep the pointer non-null for n = 0 / empty messages
        let mut ok = vec![0u8; n];
        let rc = unsafe { plume_verify_batch(self.0, if v1 { 1 } else { 2 }, n, msgs.as_ptr(), off.as_ptr(), pk.as_ptr(), nul.as_ptr(),
            c.as_ptr(), s.as_ptr(), if v1 { rp.as_ptr() } else { std::ptr::null() }, if v1 { hr.as_ptr() } else { std::ptr::null() }, ok.as_mut_ptr()) };
        if rc != 0 { return Err(last_error()); }
        Ok(ok.into_iter().map(|b| b == 1).collect())
    }
    // sign_batch(&[SecretKey], &[&[u8]], nonces: &[[u8; 32]], v1: bool) -> Vec<Result<PlumeSignature, ..>> is built the same way;
    // status bit 1 / 4 map to the `expect` panics of randomizedsigner.rs:61,91,95 (return Err instead of unwinding).
}
impl Drop for HipEngine { fn drop(&mut self) { unsafe { plume_destroy(self.0) } } }
unsafe impl Send for HipEngine {}   // one caller thread at a time (plume_hip.h "Threading")

fn put_point(dst: &mut [u8], p: &AffinePoint) {
    let e = p.to_encoded_point(false);               // 04 || x || y, or 00 for the identity
    if let (Some(x), Some(y)) = (e.x(), e.y()) { dst[..32].copy_from_slice(x); dst[32..64].copy_from_slice(y); }
    // identity: leave the 64 bytes zero
}
fn last_error() -> String { unsafe { std::ffi::CStr::from_ptr(plume_last_error()) }.to_string_lossy().into_owned() }
