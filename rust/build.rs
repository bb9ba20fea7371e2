// Links against libblu_hip.so, built by `make -C blu_amd/csrc` (hipcc, gfx950) into blu_amd/.
// BLU_HIP_LIB_DIR overrides the directory.  (Uncompiled here: no Rust toolchain in the build image.)
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("BLU_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("blu_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=blu_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=BLU_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/blu_hip.h");
}
