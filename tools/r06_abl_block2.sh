#!/bin/bash
# ablation builds of the 2- / 4-bit block kernel (tools/build_variant.sh b2abl<N> inst_block_b4 -DFLUTE_B2_ABLATE=<N>): M = 4096 / 2048 / 8192 on 4096^2
for n in "$@"; do
  if [ "$n" = shipped ]; then R06_CASE=block2 python tools/r06_lab.py; else R06_CASE=block2 FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_b2abl$n.so python tools/r06_lab.py; fi
done 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('tag'), r['M'], r.get('dtype'), r['plan']['family'], r['plan']['m_tiles'], r['plan']['grid'], r.get('us'), r.get('error'))"
