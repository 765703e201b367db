#!/bin/bash
# A/B of NTT kernel variants on the SAME box: per-kernel averages from rocprofv3 for the default build and each variant
# usage (GPU box): tools/ab_ntt.sh <outdir> <variant-name>...
out="$1"; shift
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
run() { # name libpath
  FHELIN_LIB="$2" rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/$1" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload ntt --no-cpu-baseline > "$GRAFT_REPO_ROOT/$out/$1.json" 2> "$GRAFT_REPO_ROOT/$out/$1.err"
  f=$(find "$GRAFT_REPO_ROOT/$out/$1" -name '*kernel_stats.csv' | head -1)
  echo "== $1"; cut -c1-100 "$GRAFT_REPO_ROOT/$out/$1.json" | grep -o '"value": [0-9.]*'
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ntt' in r['Name']:
        print('  %-60s calls %5s avg %8.1f us' % (r['Name'][31:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
}
run default ""
for v in "$@"; do run "$v" "$GRAFT_REPO_ROOT/tmp_variants/$v/libfhelin_amd.so"; done
run default2 ""
