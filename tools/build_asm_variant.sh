#!/bin/bash
# Build tools/_ablate/lib_<name>.so with the device code of ONE kernel unit passed through an assembly filter (dev tool):
#   tools/build_asm_variant.sh <name> <unit, e.g. nus_k_lanczos_x2> <filter command taking in.s out.s> ...
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift; shift
CS=nu_scaler_amd/csrc; L=/opt/rocm/lib/llvm/bin; T=tools/_ablate/asm_$name
make -s -j8 -C $CS >/dev/null
mkdir -p $T
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -x hip -I$CS -Iinclude"
/opt/rocm/bin/hipcc $F --cuda-device-only -S -o $T/dev.s $CS/$unit.hip 2>/dev/null
"$@" $T/dev.s $T/dev_f.s
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev_f.s -o $T/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
/opt/rocm/bin/hipcc $F --cuda-host-only -c -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -o $T/host.o $CS/$unit.hip 2>/dev/null
OTHERS=$(ls $CS/build/*.o | grep -v $unit.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ablate/lib_$name.so $T/host.o $OTHERS
echo "built tools/_ablate/lib_$name.so"
