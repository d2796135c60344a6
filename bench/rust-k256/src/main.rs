//! UNTESTED (no Rust toolchain in the build image).  Times plume_rustcrypto's own sign_v1 + verify on one core.
use plume_rustcrypto::{PlumeSignature, SecretKey};
use rand_core::OsRng;
use sha2::{Digest, Sha256};
use std::time::Instant;

fn blk(tag: &str, i: u64) -> [u8; 32] {
    // SHA256(tag || LE64(seed) || LE64(i)), seed = 0x504C554D45
    let mut h = Sha256::new();
    h.update(tag.as_bytes());
    h.update(0x504C554D45u64.to_le_bytes());
    h.update(i.to_le_bytes());
    h.finalize().into()
}

fn main() {
    let n: u64 = std::env::args().nth(1).and_then(|s| s.parse().ok()).unwrap_or(1024);
    let sigs: Vec<PlumeSignature> = (0..n)
        .map(|i| {
            // keys: any valid scalar; the generator's exact mod-(n-1) reduction is irrelevant for timing
            let sk = SecretKey::from_slice(&blk("sk", i)).unwrap_or_else(|_| SecretKey::random(&mut OsRng));
            PlumeSignature::sign_v1(&sk, &blk("msg", i), &mut OsRng)
        })
        .collect();
    let t0 = Instant::now();
    let ok = sigs.iter().filter(|s| s.verify()).count();
    let dt = t0.elapsed().as_secs_f64();
    assert_eq!(ok as u64, n);
    println!("{{\"impl\": \"plume_rustcrypto (k256)\", \"threads\": 1, \"n\": {n}, \"verifies_per_s\": {:.1}}}", n as f64 / dt);
}
