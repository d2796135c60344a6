// UNTESTED.  PLUME_HIP_LIB_DIR = directory holding libplume_hip.so (zk-nullifier-sig_amd/ in this repository)
fn main() {
    if let Ok(dir) = std::env::var("PLUME_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=plume_hip");
}
