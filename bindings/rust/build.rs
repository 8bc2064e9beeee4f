// UNVERIFIED SOURCE (no rustc in the build image).
// Link against libpetal_mi355x.so; PETAL_MI355X_LIB_DIR = directory that holds it (petal-neighbors_amd/ of the repo).
fn main() {
    let dir = std::env::var("PETAL_MI355X_LIB_DIR").unwrap_or_else(|_| "../../petal-neighbors_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=petal_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=PETAL_MI355X_LIB_DIR");
}
