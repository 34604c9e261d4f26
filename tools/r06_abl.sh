#!/bin/bash
# Ablation builds (tools/build_variant.sh <tag><N> <unit> -D<MACRO>=<N>) timed in one go:  tools/r06_abl.sh <kernel> shipped <N> <N> ...
#   m256      split-K block kernel, M = 256 on 4096^2          (abl<N>:    inst_splitk                    -DFLUTE_SK_ABLATE=<N>; lab case abl)
#   prefill   split-K block kernel, 128 x 128 tiles, M >= 1024 (abl<N>:    the same builds;                                       lab case prefill)
#   block2    2- / 4-bit block kernel, M = 2048 .. 8192        (b2abl<N>:  inst_block_b4                  -DFLUTE_B2_ABLATE=<N>; lab case block2)
#   persistm  persistent MFMA decode kernel, M = 4 / 16        (pmabl<N>:  inst_oneshot_persistm_b4_f16   -DFLUTE_PM_ABLATE=<N>; lab case persistm_abl)
kernel=$1; shift
case "$kernel" in
  m256)     lab=abl;          tag=abl;;
  prefill)  lab=prefill;      tag=abl;;
  block2)   lab=block2;       tag=b2abl;;
  persistm) lab=persistm_abl; tag=pmabl;;
  *) echo "usage: $0 m256|prefill|block2|persistm shipped <N> ..."; exit 2;;
esac
for n in "$@"; do
  if [ "$n" = shipped ]; then R06_CASE=$lab python tools/r06_lab.py; else R06_CASE=$lab FLUTE_AMD_LIB=flute_amd/csrc/libflute_amd_$tag$n.so python tools/r06_lab.py; fi
done 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); p = r.get('plan', {}); print(r.get('tag'), r.get('M'), r.get('N'), r.get('K'), r.get('dtype'), p.get('family'), p.get('kw'), p.get('splitk'), p.get('m_tiles'), p.get('grid'), r.get('us'), r.get('error'))"
