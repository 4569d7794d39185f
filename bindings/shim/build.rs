// libdbhip.so is built by `make -C databend_amd/csrc` (hipcc --offload-arch=gfx950); DBHIP_LIB_DIR names the directory that holds it.
fn main() {
    let dir = std::env::var("DBHIP_LIB_DIR").expect("DBHIP_LIB_DIR");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=dbhip");
    println!("cargo:rerun-if-env-changed=DBHIP_LIB_DIR");
}
