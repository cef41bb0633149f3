#!/bin/bash
# Builds the host layer and the library's host side with AddressSanitizer + UBSan (host code only; device code is
# compiled as usual and never runs) and runs the dump driver on the CPU.
set -e
cd "$(dirname "$0")/../.."
OUT=scratch/sanitize/dump_driver
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $SAN \
  -Wno-unused-function -Wno-unused-value -Wno-unused-result -Iinclude -c ray-tracer_amd/csrc/rtx.hip -o /tmp/rtx_san.o
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 $SAN -Iinclude -pthread /tmp/rtx_san.o \
  -x c++ ray-tracer_amd/host/ray_tracing.cpp ray-tracer_amd/host/host_capi.cpp scratch/sanitize/dump_driver.cpp -o $OUT
ASAN_OPTIONS=detect_leaks=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $OUT
