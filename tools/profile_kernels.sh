#!/bin/bash
# kernel durations + HBM traffic counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes) for a fixed-shape
# command: tools/profile_kernels.sh <outdir> <python-script-and-args...>      (GPU box)
out="$GRAFT_REPO_ROOT/$1"; shift
script="$GRAFT_REPO_ROOT/$1"; shift
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- python3 "$script" "$@" > "$out/run_stats.json" 2> "$out/run_stats.err" || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o p -- python3 "$script" "$@" > "$out/run_fetch.json" 2> "$out/run_fetch.err" || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o p -- python3 "$script" "$@" > "$out/run_write.json" 2> "$out/run_write.err" || exit 1
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$out" > "$out/summary.json"
cat "$out/run_stats.json" | cut -c1-400
rm -rf "$out/stats" "$out/fetch" "$out/write"
