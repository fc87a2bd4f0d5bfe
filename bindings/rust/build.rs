// build.rs of the patched crate: link librnamc.so (the C ABI of include/rnamc.h).
// RNAMC_LIB_DIR names the directory that holds librnamc.so (default: the in-tree build of
// this repository, rna_algos_amd/).  The ROCm runtime is a dependency of librnamc.so itself.
use std::env;
use std::path::PathBuf;

fn main() {
  let dir = env::var("RNAMC_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
    PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../rna_algos_amd")
  });
  println!("cargo:rerun-if-env-changed=RNAMC_LIB_DIR");
  println!("cargo:rustc-link-search=native={}", dir.display());
  println!("cargo:rustc-link-lib=dylib=rnamc");
  // so that the binaries find it without LD_LIBRARY_PATH
  println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
