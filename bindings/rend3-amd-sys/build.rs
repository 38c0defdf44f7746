// Points the linker at the in-tree library (rend3_amd/librend3_amd.so, built by rend3_amd/build.py with hipcc).
fn main() {
    let dir = std::env::var("REND3_AMD_LIB_DIR").unwrap_or_else(|_| format!("{}/../../rend3_amd", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rend3_amd");
    println!("cargo:rerun-if-env-changed=REND3_AMD_LIB_DIR");
}
