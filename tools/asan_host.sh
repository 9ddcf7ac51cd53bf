#!/bin/bash
# Host-side AddressSanitizer build of the C-ABI layer (frankenz_hip.hip: argument checks, label / dictionary
# table construction, chunking and dispatch logic) -- device code is NOT instrumented (-fno-gpu-sanitize; the GPU
# pool refuses sanitizer runs), so this covers what runs on the CPU: the library loads, exports every symbol of
# include/frankenz_hip.h, and the error paths of the CPU-side ABI tests run clean under ASan.
#   ./tools/asan_host.sh          (needs the regular build's fz_inst_b*.o next to the sources)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/fz_asan}
mkdir -p $OUT
cd $ROOT/frankenz_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
    -fsanitize=address -fno-gpu-sanitize -shared-libsan -c frankenz_hip.hip -o $OUT/frankenz_hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libsan -o $OUT/libfrankenz_hip_asan.so \
    $OUT/frankenz_hip.o fz_inst_b4.o fz_inst_b5.o fz_inst_b6.o fz_inst_b7.o fz_inst_b8.o fz_inst_b12.o fz_inst_b16.o fz_inst_b24.o fz_inst_b32.o
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
cd $ROOT
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 FRANKENZ_HIP_LIB=$OUT/libfrankenz_hip_asan.so python -m pytest tests/test_abi.py -q -m "not gpu"
