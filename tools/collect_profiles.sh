#!/bin/bash
# Collects, on a GPU box, everything profiles/ holds for one round:
#   kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench.py command
#   stats_bench.json   the JSON line bench.py printed under that profiler run
#   pmc.json           per-kernel FETCH_SIZE / WRITE_SIZE / SQ counters (separate --pmc passes, one step each)
#   bench_100k.json    the JSON line of an unprofiled default run (with the CPU baseline)
# usage (from the repo root):  gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh gpurun_out/final [bench.py arguments]'
set -u
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/final}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="${@:2}"   # extra bench.py arguments (e.g. --config 4)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o k -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-sample 0 --blasrmc-reads 0 $ARGS \
    > "$OUT/stats_bench.json" 2> "$OUT/kt.err"
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
# (PMC passes only on request, PMC=1, each under its own time limit: rocprofv3 --pmc hangs inside torch's generator
#  for the configs[2] set-up on this image; tools/pmc_config1.sh collects them on configs[1])
i=0
for ctr in ${PMC:+"FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"}; do
  i=$((i + 1))
  timeout 600 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc$i" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 $ARGS \
      > /dev/null 2> "$OUT/pmc$i.err"
done
[ -n "${PMC:-}" ] && python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc.json" $(find "$OUT"/pmc* -name '*counter_collection.csv') > "$OUT/pmc_top.txt"
rm -rf "$OUT"/kt "$OUT"/pmc[0-9]
cd "$ROOT" && python3 bench.py $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
