#!/bin/bash
# A/B of NTT variants on one box by the bench's own NTT section: tools/ab_ntt_quick.sh <variant>... (alternates with the working build)
for i in 1 2; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export FHELIN_LIB=$GRAFT_REPO_ROOT/tmp_variants/$v/libfhelin_amd.so; else unset FHELIN_LIB; fi
    python bench.py --workload ntt --no-cpu-baseline --ntt-steps 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-working}', d['value'], d['roofline']['frac'])"
  done
done
