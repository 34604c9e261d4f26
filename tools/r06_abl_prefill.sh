#!/bin/bash
# ablation builds of the split-K block kernel (tools/build_variant.sh abl<N> inst_splitk -DFLUTE_SK_ABLATE=<N>) on throughput-bound launches:
# 128 x 128 tiles, M = 1024 / 2048 / 4096 on 4096^2
for n in "$@"; do
  if [ "$n" = shipped ]; then R06_CASE=prefill python tools/r06_lab.py; else R06_CASE=prefill FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_abl$n.so python tools/r06_lab.py; fi
done 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('tag'), r['M'], r['plan']['family'], r['plan']['kw'], r['plan']['m_tiles'], r['plan']['grid'], r.get('us'), r.get('error'))"
