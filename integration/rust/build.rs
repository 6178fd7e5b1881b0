// build.rs of the reference crate once it links the MI355X render path.
fn main() {
    let dir = std::env::var("RT_HIP_DIR").expect("set RT_HIP_DIR to the directory that holds librt_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RT_HIP_DIR");
}
