#!/bin/bash
# counter passes (one rocprofv3 --pmc run per group, no tracing options) for a repo script (GPU box):
#   tools/profile_pmc.sh <outdir> "<group1>;<group2>;..." <script> [args...]      group = space-separated counter names
out="$GRAFT_REPO_ROOT/$1"; shift
groups="$1"; shift
script="$GRAFT_REPO_ROOT/$1"; shift
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
i=0
IFS=';' read -ra G <<< "$groups"
for g in "${G[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $g --output-format csv -d "$out/pass$i" -o p -- python3 "$script" "$@" > "$out/run$i.out" 2> "$out/run$i.err" || { tail -5 "$out/run$i.err"; exit 1; }
done
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$out" > "$out/summary.json"
rm -rf "$out"/pass*
cat "$out/summary.json"
