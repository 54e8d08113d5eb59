// Links libmarlin_hip.so (built in-tree by `make -C marlin_amd/csrc`).
// MARLIN_HIP_LIB_DIR overrides the default ../marlin_amd.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("MARLIN_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("marlin_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    let lib = if env::var("CARGO_FEATURE_BN254").is_ok() { "marlin_hip_bn254" } else { "marlin_hip" };
    println!("cargo:rustc-link-lib=dylib={}", lib);
    println!("cargo:rerun-if-env-changed=MARLIN_HIP_LIB_DIR");
}
