#!/bin/bash
# ablation builds of the persistent MFMA decode kernel (tools/build_variant.sh pmabl<N> inst_oneshot_persistm_b4 -DFLUTE_PM_ABLATE=<N>)
for n in "$@"; do
  if [ "$n" = shipped ]; then R06_CASE=persistm_abl python tools/r06_lab.py; else R06_CASE=persistm_abl FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_pmabl$n.so python tools/r06_lab.py; fi
done 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('tag'), r['M'], r['N'], r['K'], r['plan']['grid'], r.get('us'), r.get('error'))"
