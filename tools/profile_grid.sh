#!/bin/bash
# rocprofv3 kernel trace of a repo script, summarised per kernel and launch shape (GPU box):
#   tools/profile_grid.sh <outdir> <name-filter,comma-separated> <script> [args...]
out="$GRAFT_REPO_ROOT/$1"; shift
filt="$1"; shift
script="$GRAFT_REPO_ROOT/$1"; shift
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/raw" -o p -- python3 "$script" "$@" > "$out/run.out" 2> "$out/run.err" || { tail -5 "$out/run.err"; exit 1; }
tr=$(find "$out/raw" -name '*kernel_trace.csv' | head -1)
python3 "$GRAFT_REPO_ROOT/tools/trace_by_grid.py" "$tr" ${filt//,/ } > "$out/by_grid.txt"
rm -rf "$out/raw"
cat "$out/by_grid.txt"
