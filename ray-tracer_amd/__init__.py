"""MI355X (gfx950) path-tracing backend for magmast/ray-tracer.

Layout
  csrc/      HIP kernels and the C ABI declared in include/rtx.h        (librtx.so)
  host/      C++ host layer: the reference's constructors + the flattener   (librtx_host.so)
  scene.py   constructor-level scene description (names follow the reference's Rust types)
  scenes.py  seeded restatements of the reference's nine scenes + `default`
  capi.py    ctypes plumbing shared by the native libraries
  render.py  Camera / Scene front end over the native libraries
  parallel.py  multi-GPU sample splitting (one process per GPU, RCCL reduce)

The directory name contains a hyphen; import it with importlib.import_module("ray-tracer_amd").
Importing the package loads no native code; `render` does, and fails loudly when the HIP library is absent.
"""
