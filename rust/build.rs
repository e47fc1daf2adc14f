// cc + bindgen build script of the shim.
//   1. the engine library: if ../stable-diffusion-xl-burn_amd/lib/libsdxl_mi355.so is missing it is built with the repo's own
//      recipe (hipcc --offload-arch=gfx950 over csrc/*.hip, csrc/*.cpp -- stable-diffusion-xl-burn_amd/build.py); SDXL_MI355_LIB_DIR
//      points at a prebuilt one;
//   2. bindgen over include/sdxl_mi355.h -> $OUT_DIR/bindings.rs (every symbol the shim calls is declared there);
//   3. a five-line C translation unit compiled with `cc` that pins the header's enum values at compile time, so an ABI
//      drift between header and shim is a build error rather than a wrong dtype at run time.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..");
    let include = root.join("include");
    let lib_dir = env::var("SDXL_MI355_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| root.join("stable-diffusion-xl-burn_amd").join("lib"));
    if !lib_dir.join("libsdxl_mi355.so").exists() {
        let st = Command::new("python3")
            .arg(root.join("stable-diffusion-xl-burn_amd").join("build.py"))
            .status()
            .expect("python3 stable-diffusion-xl-burn_amd/build.py (needs hipcc)");
        assert!(st.success(), "building libsdxl_mi355.so failed");
    }
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=sdxl_mi355");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    // the HIP runtime for the host<->device copies of the generic tensor bridge (hipMalloc / hipMemcpy / hipFree)
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rerun-if-changed={}", include.join("sdxl_mi355.h").display());

    let bindings = bindgen::Builder::default()
        .header(include.join("sdxl_mi355.h").to_str().unwrap())
        .allowlist_function("sdxl_.*")
        .allowlist_type("sdxl_.*")
        .allowlist_var("SDXL_.*")
        .generate()
        .expect("bindgen over include/sdxl_mi355.h");
    bindings
        .write_to_file(PathBuf::from(env::var("OUT_DIR").unwrap()).join("bindings.rs"))
        .unwrap();

    cc::Build::new()
        .include(&include)
        .file("src/abi_check.c")
        .compile("sdxl_mi355_abi_check");
}
