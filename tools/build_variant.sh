#!/bin/bash
# Development variant of the library: the shipped objects with ONE instantiation unit rebuilt under extra flags.
#   tools/build_variant.sh <tag> <unit, e.g. inst_splitk> <flags...>   ->  flute_amd/csrc/libflute_amd_<tag>.so   (load with FLUTE_AMD_LIB=<path>)
set -e
tag=$1; unit=$2; shift 2
cd "$(dirname "$0")/../flute_amd/csrc"
make -j8 >/dev/null
mkdir -p build_abl
pre=""
case "$unit" in inst_oneshot_*) pre="-mllvm -amdgpu-kernarg-preload-count=14";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC $pre "$@" -c $unit.hip -o build_abl/${unit}_$tag.o
objs=$(ls build/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libflute_amd_$tag.so $objs build_abl/${unit}_$tag.o
echo "built flute_amd/csrc/libflute_amd_$tag.so"
