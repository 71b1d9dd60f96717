// Links libfourier.so (SONAME libfourier.so.0, the name fourier-ffi's CMake gives its library:
// fourier-ffi/CMakeLists.txt:38-65).  FOURIER_HIP_LIB_DIR = directory holding it (default: the in-tree build,
// ../../fourier_amd/lib).  HIP itself is a dependency of libfourier.so, not of this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=FOURIER_HIP_LIB_DIR");
    if env::var_os("CARGO_FEATURE_HIP").is_none() {
        return;
    }
    let dir = env::var_os("FOURIER_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").unwrap()).join("../../fourier_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=fourier");
    // let `cargo test` / `cargo run` find the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
