#!/bin/bash
# registers / scratch / occupancy of every kernel of one instantiation unit:  tools/kernel_resources.sh inst_splitk [extra hipcc flags]
unit=$1; shift
pre=""; case "$unit" in inst_oneshot_*) pre="-mllvm -amdgpu-kernarg-preload-count=14";; esac
cd "$(dirname "$0")/../flute_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC $pre "$@" -c $unit.hip -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - |
  sed 's/Function Name: _ZN9flute_amd//; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/' | tr '\t' ' '
rm -f /tmp/kr_$$.o
