// Links the prebuilt HIP library.  Build it first with `make -C renderer_amd/csrc` (hipcc,
// --offload-arch=gfx950) and point RND_AMD_LIB_DIR at the directory holding librnd_amd.so.
// The declarations in src/lib.rs are written by hand, so neither bindgen nor libclang is needed.
fn main() {
    let dir = std::env::var("RND_AMD_LIB_DIR").unwrap_or_else(|_| "../../../renderer_amd/csrc".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rnd_amd");
    println!("cargo:rerun-if-env-changed=RND_AMD_LIB_DIR");
}
