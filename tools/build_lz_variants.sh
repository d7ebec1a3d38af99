#!/bin/bash
# Build A/B variants of one kernel unit (UNIT, default nus_k_lanczos_x2) into tools/_ablate/lib_<name>.so (timing only; dev tool).
#   [UNIT=nus_k_flow] tools/build_lz_variants.sh name1="-DFLAG=1 -DX=2" name2="" old=@<git-rev>   ("@rev": the unit as of that commit)
set -e
cd "$(dirname "$0")/.."
CS=nu_scaler_amd/csrc
make -s -j8 -C $CS >/dev/null
mkdir -p tools/_ablate
UNIT=${UNIT:-nus_k_lanczos_x2}
OTHERS=$(ls $CS/build/*.o | grep -v $UNIT.o)
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  src=$CS/$UNIT.hip
  if [[ "$flags" == @* ]]; then
    git show "${flags#@}:$CS/$UNIT.hip" > tools/_ablate/src_$name.hip
    src=tools/_ablate/src_$name.hip; flags=""
  fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -x hip -I$CS -Iinclude -DNUS_DEV_BUILD $flags \
      -c -o tools/_ablate/lz_$name.o $src &
done
wait
for spec in "$@"; do
  name="${spec%%=*}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ablate/lib_$name.so tools/_ablate/lz_$name.o $OTHERS
  echo "built tools/_ablate/lib_$name.so"
done
