//! Locates `libmantagpu.so` (built by `make -C manta_rs_amd/csrc`, hipcc, gfx950) and tells cargo to link it.
//! Search order: `MANTAGPU_LIB_DIR`, then `<this repo>/manta_rs_amd/lib`. ROCm's `libamdhip64.so` is found through the
//! rpath `/opt/rocm/lib` embedded in the shared library.
use std::{env, path::PathBuf};

fn main() {
    println!("cargo:rerun-if-env-changed=MANTAGPU_LIB_DIR");
    let dir = env::var_os("MANTAGPU_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").expect("cargo sets CARGO_MANIFEST_DIR"))
            .join("../../manta_rs_amd/lib")
    });
    let lib = dir.join("libmantagpu.so");
    assert!(
        lib.exists(),
        "{} not found: build it with `make -C manta_rs_amd/csrc` or set MANTAGPU_LIB_DIR",
        lib.display()
    );
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=mantagpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-changed={}", lib.display());
    println!("cargo:rerun-if-changed=../../include/mantagpu.h");
}
