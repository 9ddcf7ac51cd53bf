#!/bin/bash
# development build of the MAIN unit only (ABI, k-NN, planes: ~20 s), linked with the band-count units of the last full build:
#   tools/mainbuild.sh [-o libfz_x.so] [-DMACRO ...]      (default output: libfrankenz_hip.so, with the compiler's resource report)
set -e
cd "$(dirname "$0")/../frankenz_amd/csrc"
OUT=libfrankenz_hip.so
if [ "$1" = "-o" ]; then OUT=$2; shift 2; fi
# the band-count units share headers with the main unit (fz_ctx.h holds the context's layout): linking stale ones against a changed
# header gives a library that reads fields at old offsets (round 5 lost an hour to exactly that)
for h in fz_ctx.h fz_device.h fz_kernels.h fz_launch.h fz_hist.h fz_nolist.h fz_plane.h fz_modec.h fz_knn.h fz_fastmath.h fz_tables.h fz_inst.hip ../../include/frankenz_hip.h; do
  if [ "$h" -nt fz_inst_b5.o ]; then echo "mainbuild: $h is newer than the band-count units -- run __graft_entry__.build(force=True)"; exit 1; fi
done
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
if [ "$OUT" = libfrankenz_hip.so ]; then
  /opt/rocm/bin/hipcc $F -Rpass-analysis=kernel-resource-usage "$@" -c "$PWD/frankenz_hip.hip" -o frankenz_hip.o 2> kernel_resources.txt || { tail -30 kernel_resources.txt; exit 1; }
  O=frankenz_hip.o
  /opt/rocm/bin/hipcc $F "$@" --cuda-device-only -S frankenz_hip.hip -o kernel_isa_plane_rows.s.all 2>/dev/null
  (cd ../.. && python3 -c "
import __graft_entry__ as g, sys
g._keep_plane_isa()
p = g.check_hand_scheduled()
if p: sys.exit('mainbuild: the hand-scheduled kernels failed their checks (run __graft_entry__.build(force=True) for the fallback):\n  ' + '\n  '.join(p))")
else
  O=/tmp/fz_main_$$.o
  /opt/rocm/bin/hipcc $F "$@" -c frankenz_hip.hip -o $O
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $O fz_inst_b*.o
grep -A10 "k_plane_rows\|k_knn_mfma" kernel_resources.txt | grep "Function Name\|VGPRs:\|ScratchSize" | sed 's/.*remark: //' | paste - - - | sed 's/\[-Rpass[^]]*\]//g' | cut -c1-160
ls -la $OUT
