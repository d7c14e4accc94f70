// Links libls_amd.so (built by `python constraint-solver_amd/build.py`).
fn main() {
    let dir = std::env::var("LS_AMD_LIB_DIR").expect("set LS_AMD_LIB_DIR to <repo>/constraint-solver_amd/lib");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=ls_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=LS_AMD_LIB_DIR");
}
