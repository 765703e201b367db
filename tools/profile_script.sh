#!/bin/bash
# rocprofv3 kernel stats of an arbitrary repo script (GPU box): tools/profile_script.sh <outdir> <script> [args...]
out="$GRAFT_REPO_ROOT/$1"; shift
script="$GRAFT_REPO_ROOT/$1"; shift
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw" -o p -- python3 "$script" "$@" > "$out/run.out" 2> "$out/run.err" || { tail -5 "$out/run.err"; exit 1; }
cp "$(find "$out/raw" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
rm -rf "$out/raw"
cat "$out/run.out"
python3 - "$out/kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:18]:
    print('%-52s calls %6s total %8.2f ms avg %8.1f us %5.1f%%' % (r['Name'].replace('fhelin::','').replace('(anonymous namespace)::','')[:52], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
print('total kernel ms', tot/1e6)
PY
