// Where libtakzero_hip.so lives: TAKZERO_HIP_LIB_DIR, or <repo>/takzero_amd (python -m takzero_amd.build puts it there).
fn main() {
    let dir = std::env::var("TAKZERO_HIP_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../../takzero_amd", std::env::var("CARGO_MANIFEST_DIR").unwrap()));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=takzero_hip");
    println!("cargo:rerun-if-env-changed=TAKZERO_HIP_LIB_DIR");
}
