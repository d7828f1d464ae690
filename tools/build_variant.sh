#!/bin/bash
# Builds a variant of the product library for same-box A/B timing through DXMI_LIB:
#   tools/build_variant.sh <name> <source.hip> [-DFLAG ...]   ->  diffusion-by-maxentirl_amd/dxmi_hip/libdxmi_<name>.so
# Only <source.hip> is recompiled (with the extra flags); every other object is the product build's.  Variant libraries are
# scratch: delete them before the round ends (they would ship to the GPU box beside the product library).
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../diffusion-by-maxentirl_amd/csrc"
make -j8 > /dev/null
obj=/tmp/dxmi_variant_${name}_${src%.hip}.o
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -c "$src" -o "$obj"
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs "$obj" -o ../dxmi_hip/libdxmi_${name}.so
echo "built dxmi_hip/libdxmi_${name}.so"
