#!/bin/bash
# Build a complete A/B library with the same -D flags on EVERY kernel unit: tools/build_full_variant.sh name "-DFLAG=1 ..."  ->  tools/_ablate/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
CS=nu_scaler_amd/csrc
make -s -j8 -C $CS >/dev/null
name=$1; flags=$2
d=tools/_ablate/full_$name; mkdir -p $d
for f in $CS/nus_k_*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -x hip -I$CS -Iinclude -DNUS_DEV_BUILD $flags -c -o $d/$(basename $f .hip).o $f &
done
wait
HOSTO=$(ls $CS/build/*.o | grep -v "nus_k_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ablate/lib_$name.so $d/*.o $HOSTO
echo "built tools/_ablate/lib_$name.so"
