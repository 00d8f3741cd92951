// RT_MI355X_LIB_DIR = directory holding librt_hip.so (raytracer-rs_amd/lib of this repository after `make`).
fn main() {
    let dir = std::env::var("RT_MI355X_LIB_DIR").unwrap_or_else(|_| "../../../raytracer-rs_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rerun-if-env-changed=RT_MI355X_LIB_DIR");
}
