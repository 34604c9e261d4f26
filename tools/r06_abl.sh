#!/bin/bash
# ablation builds (tools/build_variant.sh abl<N> inst_splitk -DFLUTE_SK_ABLATE=<N>) timed on M = 256 x 4096^2
for n in "$@"; do
  if [ "$n" = shipped ]; then R06_CASE=abl python tools/r06_lab.py; else R06_CASE=abl FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_abl$n.so python tools/r06_lab.py; fi
done 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('tag'), r['plan']['kw'], r['plan']['splitk'], r.get('us'), r.get('error'))"
