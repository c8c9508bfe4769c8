// Links libneedle_capi.so.  NEEDLE_CAPI_LIB_DIR points at the directory that holds it
// (needle_amd/lib in this repository after `make -C needle_amd/csrc`).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("NEEDLE_CAPI_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../needle_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=needle_capi");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=NEEDLE_CAPI_LIB_DIR");
}
