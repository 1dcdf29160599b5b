// Links libpt_amd.so. PT_AMD_LIB_DIR names the directory that holds it (default: ../../path_tracing_amd).
fn main() {
    let dir = std::env::var("PT_AMD_LIB_DIR").unwrap_or_else(|_| {
        let manifest = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{manifest}/../../path_tracing_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=pt_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=PT_AMD_LIB_DIR");
}
