#!/bin/bash
# rocprofv3 kernel trace + stats of the timed forward passes (GPU box): tools/profile_forward.sh <outdir> <log_n> [steps]
# BENCH_ARGS="--batch 4": extra arguments for bench.py (the window then covers steps x samples-per-step)
out="$1"; logn="$2"; steps="${3:-3}"
mkdir -p "$GRAFT_REPO_ROOT/$out"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/raw" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" \
   --forward-only --steps "$steps" --warmup 1 --log-n "$logn" $BENCH_ARGS > "$GRAFT_REPO_ROOT/$out/bench.json" 2> "$GRAFT_REPO_ROOT/$out/bench.err"
st=$(find "$GRAFT_REPO_ROOT/$out/raw" -name '*kernel_stats.csv' | head -1)
tr=$(find "$GRAFT_REPO_ROOT/$out/raw" -name '*kernel_trace.csv' | head -1)
cp "$st" "$GRAFT_REPO_ROOT/$out/kernel_stats.csv"
ms=$(python3 -c "import json,sys; print(json.loads(open('$GRAFT_REPO_ROOT/$out/bench.json').read().strip().splitlines()[-1])['value'])")
spp=$(python3 -c "import json,sys; print(json.loads(open('$GRAFT_REPO_ROOT/$out/bench.json').read().strip().splitlines()[-1]).get('samples_per_pass',1) * json.loads(open('$GRAFT_REPO_ROOT/$out/bench.json').read().strip().splitlines()[-1]).get('passes_per_step',1))")
win=$(python3 -c "print($ms*$steps*$spp/1000.0)")
python3 "$GRAFT_REPO_ROOT/tools/trace_summary.py" "$tr" "$win" > "$GRAFT_REPO_ROOT/$out/trace_summary.txt"
TRACE_WINDOW_S="$win" python3 "$GRAFT_REPO_ROOT/tools/trace_by_grid.py" "$tr" ntt_ > "$GRAFT_REPO_ROOT/$out/ntt_by_grid.txt"
TRACE_WINDOW_S="$win" python3 "$GRAFT_REPO_ROOT/tools/trace_by_grid.py" "$tr" > "$GRAFT_REPO_ROOT/$out/all_by_grid.txt"
cat "$GRAFT_REPO_ROOT/$out/bench.json" | cut -c1-400; cat "$GRAFT_REPO_ROOT/$out/trace_summary.txt"
rm -rf "$GRAFT_REPO_ROOT/$out/raw"
