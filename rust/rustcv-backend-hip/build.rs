// Links the prebuilt librustcv_hip.so (make -C rustcv_amd/csrc).  Same role as rustcv-camera/build.rs:9-31,
// which compiles the Objective-C bridge with the cc crate; here the library is built by hipcc.
fn main() {
    let dir = std::env::var("RUSTCV_HIP_LIB_DIR").unwrap_or_else(|_| "../../rustcv_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rustcv_hip");
    println!("cargo:rerun-if-env-changed=RUSTCV_HIP_LIB_DIR");
}
