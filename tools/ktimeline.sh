#!/bin/bash
# rocprofv3 --kernel-trace of a bench.py command, turned into the timeline of its last step (tools/step_timeline.py).
# usage: gpurun -- 'bash tools/ktimeline.sh gpurun_out/<dir> [bench.py arguments]'
set -u
ROOT=$(pwd)
OUT=$ROOT/$1
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o k -- python3 "$ROOT/bench.py" --from-reads 0 "$@" \
    > "$OUT/tl_bench.json" 2> "$OUT/kt.err"
TRACE=$(find "$OUT/kt" -name '*kernel_trace.csv' | head -1)
MS=$(python3 -c "import json,sys; print(json.loads(open('$OUT/tl_bench.json').read().strip().splitlines()[-1])['ms_per_step'])")
python3 "$ROOT/tools/step_timeline.py" "$TRACE" "$MS" 0.25 > "$OUT/timeline.txt"
rm -rf "$OUT/kt"
tail -60 "$OUT/timeline.txt"
