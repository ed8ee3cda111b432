// Links libndfft_mi355x.so (built by `make -C ndrustfft_amd/csrc`).
fn main() {
    let dir = std::env::var("NDFFT_LIB_DIR").unwrap_or_else(|_| "../ndrustfft_amd/csrc".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=ndfft_mi355x");
    println!("cargo:rerun-if-env-changed=NDFFT_LIB_DIR");
}
