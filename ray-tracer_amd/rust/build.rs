// Link against the in-tree librt_mi355x.so (built by `make -C ray-tracer_amd/csrc`).
fn main() {
    let dir = std::env::var("RT_MI355X_LIB_DIR").unwrap_or_else(|_| {
        format!("{}/../lib", std::env::var("CARGO_MANIFEST_DIR").unwrap())
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rt_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
