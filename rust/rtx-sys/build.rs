// Builds librtx.so from the HIP sources of this repository (or links a prebuilt one named by RTX_LIB_DIR).
use std::{env, path::PathBuf, process::Command};

fn main() {
    println!("cargo:rerun-if-env-changed=RTX_LIB_DIR");
    if let Ok(dir) = env::var("RTX_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=rtx");
        return;
    }
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../..");
    let src = root.join("ray-tracer_amd/csrc/rtx.hip");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    for f in ["ray-tracer_amd/csrc/rtx.hip", "ray-tracer_amd/csrc/rtx_device.hpp", "ray-tracer_amd/csrc/accel.hpp", "include/rtx.h"] {
        println!("cargo:rerun-if-changed={}", root.join(f).display());
    }
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let status = Command::new(hipcc)
        .args(["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-fPIC", "-shared"])
        .arg(format!("-I{}", root.join("include").display()))
        .arg(&src)
        .arg("-o")
        .arg(out.join("librtx.so"))
        .status()
        .expect("hipcc not found: set HIPCC or RTX_LIB_DIR");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rtx");
}
