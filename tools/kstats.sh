#!/bin/bash
# rocprofv3 --kernel-trace --stats of a bench.py command; keeps the per-kernel summary and the JSON line.
# usage: gpurun -- 'bash tools/kstats.sh gpurun_out/<dir> [bench.py arguments]'
set -u
ROOT=$(pwd)
OUT=$ROOT/$1
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o k -- python3 "$ROOT/bench.py" "$@" \
    > "$OUT/stats_bench.json" 2> "$OUT/kt.err"
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/kt"
head -40 "$OUT/kernel_stats.csv" | cut -c1-200
