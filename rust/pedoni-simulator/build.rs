// Links pedoni-simulator against libpedoni_hip.so (UNVERIFIED: never compiled, see
// src/models/sfm_hip.rs).  PEDONI_HIP_LIB_DIR = the directory holding the library
// (<repo>/pedoni_amd/lib after `python -m pedoni_amd.build`).
fn main() {
    println!("cargo:rerun-if-env-changed=PEDONI_HIP_LIB_DIR");
    if let Ok(dir) = std::env::var("PEDONI_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=pedoni_hip");
}
