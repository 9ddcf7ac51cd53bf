#!/bin/bash
# development build: only the 5-band unit (mask-free variants: -DFZ_DEV_FAST) + stubs for the other band counts, so the
# library is small and builds in ~1.5 min.  NOT a release build: run __graft_entry__.build(force=True) before measuring
# for the record or committing results.
#   tools/devbuild.sh [-o name.so] [extra hipcc flags for the 5-band unit]
#   FZ_DEV_MAIN=1 forces the main unit (k-NN, planes, ABI) to be rebuilt, FZ_DEV_MAINFLAGS='-DX' adds flags to it
set -e
cd "$(dirname "$0")/../frankenz_amd/csrc"
OUT=libfrankenz_hip.so
if [ "$1" = "-o" ]; then OUT=$2; shift 2; fi
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off"
B5=/tmp/fz_dev_b5_${OUT%.so}.o          # (per output name: variants build side by side)
/opt/rocm/bin/hipcc $F -DFZ_BT=5 -DFZ_DEV_FAST "$@" -c fz_inst.hip -o $B5 &
if [ ! -f /tmp/fz_dev_main.o ] || [ frankenz_hip.hip -nt /tmp/fz_dev_main.o ] || [ fz_ctx.h -nt /tmp/fz_dev_main.o ] || [ -n "$FZ_DEV_MAIN" ]; then
  /opt/rocm/bin/hipcc $F $FZ_DEV_MAINFLAGS -mllvm -amdgpu-mfma-vgpr-form -c frankenz_hip.hip -o /tmp/fz_dev_main.o &
fi
if [ ! -f /tmp/fz_dev_stubs.o ] || [ fz_ctx.h -nt /tmp/fz_dev_stubs.o ]; then
  /opt/rocm/bin/hipcc $F -c ../../tools/dev_stubs.hip -o /tmp/fz_dev_stubs.o &
fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT /tmp/fz_dev_main.o $B5 /tmp/fz_dev_stubs.o
ls -la $OUT
