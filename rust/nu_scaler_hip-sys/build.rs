// Tell cargo where libnuscaler_hip.so lives.  NUSCALER_HIP_LIB_DIR = the directory that holds it (the
// `nu_scaler_amd/lib` of a built checkout of the HIP repository, or an install prefix's lib directory).
// The library is built with `make -C nu_scaler_amd/csrc` (hipcc --offload-arch=gfx950) and needs only libamdhip64
// at run time.  Same shape as the reference's own FFI crates (nu_scaler_core/fsr3-sys/build.rs: an environment
// variable names the SDK directory), minus bindgen: src/lib.rs is generated from include/nuscaler_hip.h by
// tools/gen_rust_sys.py of the HIP repository and committed.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=NUSCALER_HIP_LIB_DIR");
    let dir = env::var("NUSCALER_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from("/opt/nuscaler_hip/lib"));
    if !dir.join("libnuscaler_hip.so").exists() {
        println!(
            "cargo:warning=libnuscaler_hip.so not found in {} (set NUSCALER_HIP_LIB_DIR); linking will fail",
            dir.display()
        );
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=nuscaler_hip");
    // let the final binary find the library next to where it was linked from
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
