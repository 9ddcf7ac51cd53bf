#!/bin/bash
# device code of one translation unit's object, disassembled: tools/disasm.sh OBJ.o OUT.s  (then tools/isa_loops.py OUT.s KERNEL-SUBSTRING)
set -e
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn --mcpu=gfx950 $T/dev.co > "$2"
rm -rf $T
