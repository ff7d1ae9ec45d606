// Builds ldpc_amd/csrc with hipcc through `make` and links the resulting
// libldpc_hip.so.  (With the `cc` crate: cc::Build::new().compiler("hipcc")
// .flag("--offload-arch=gfx950").files([...]) gives the same library; bindgen on
// include/ldpc_hip.h regenerates the extern block of src/lib.rs.)
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let root = PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("../..");
    let status = Command::new("make")
        .arg("-C")
        .arg(root.join("ldpc_amd/csrc"))
        .status()
        .expect("make failed to start");
    assert!(status.success(), "building libldpc_hip.so failed");
    println!("cargo:rustc-link-search=native={}", root.join("ldpc_amd").display());
    println!("cargo:rustc-link-lib=dylib=ldpc_hip");
    println!("cargo:rerun-if-changed={}", root.join("include/ldpc_hip.h").display());
}
