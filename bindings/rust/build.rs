fn main() {
    // AUDIOMATCH_AMD_LIB_DIR = directory that holds libaudiomatch_amd.so (audio-matcher_amd/)
    let dir = std::env::var("AUDIOMATCH_AMD_LIB_DIR").unwrap_or_else(|_| "../../audio-matcher_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=audiomatch_amd");
    println!("cargo:rerun-if-env-changed=AUDIOMATCH_AMD_LIB_DIR");
}
