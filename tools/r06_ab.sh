#!/bin/bash
# A/B of library variants on ONE box: tools/r06_ab.sh <case> <reps> <variant>...   (variant: shipped | <tag> of libflute_amd_<tag>.so)
c=$1; reps=$2; shift 2
for r in $(seq $reps); do for v in "$@"; do
  if [ "$v" = shipped ]; then R06_CASE=$c python tools/r06_lab.py; else R06_CASE=$c FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_$v.so python tools/r06_lab.py; fi
done; done 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r.get('tag'), r['M'], r['N'], r['K'], r['dtype'], 'fam', r.get('plan', {}).get('family'), 'rt', r.get('plan', {}).get('m_tiles'), 'kw', r.get('plan', {}).get('kw'), 'sk', r.get('plan', {}).get('splitk'), 'grid', r.get('plan', {}).get('grid'), r.get('us'), r.get('error'))"
