#!/bin/bash
# Ablation builds of the split-K block kernel: libflute_amd_abl<N>.so = the shipped objects with inst_splitk.o rebuilt
# under -DFLUTE_SK_ABLATE=<N> (bits: 1 no activation requests in the loop, 2 no weight requests, 4 no MFMA, 8 no lookups,
# 16 no fragment reads, 32 no barriers, 128 no seam - also in the skinny kernel).  Results are WRONG by construction; only the timings mean something.
#   tools/splitk_ablate.sh 1 2 4 ...   then   FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_abl4.so python tools/splitk_lab.py time
set -e
cd "$(dirname "$0")/../flute_amd/csrc"
make -j16 >/dev/null
for n in "$@"; do
  mkdir -p build_abl
  if [ "$n" = spin0 ]; then   # not an ablation: kXwgSpinLimit = 0, every owner that is not last abandons its share (xwg.h's fallback path)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -DFLUTE_XWG_SPIN_LIMIT=0 -c inst_splitk.hip -o build_abl/inst_splitk_$n.o &
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -mllvm -amdgpu-kernarg-preload-count=14 -DFLUTE_XWG_SPIN_LIMIT=0 -c inst_oneshot_skinny_b4.hip -o build_abl/inst_skinny_$n.o &
    continue
  fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -DFLUTE_SK_ABLATE=$n -c inst_splitk.hip -o build_abl/inst_splitk_$n.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -mllvm -amdgpu-kernarg-preload-count=14 -DFLUTE_SK_ABLATE=$n -c inst_oneshot_skinny_b4.hip -o build_abl/inst_skinny_$n.o &
done
wait
for n in "$@"; do
  objs=$(ls build/*.o | grep -v "inst_splitk.o\|inst_oneshot_skinny_b4.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libflute_amd_abl$n.so $objs build_abl/inst_splitk_$n.o build_abl/inst_skinny_$n.o
done
