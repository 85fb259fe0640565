#!/bin/bash
# rocprofv3 of tools/anchor_bench.py (no torch in that process): the kernel summary, then FETCH_SIZE / WRITE_SIZE per kernel
# usage: gpurun -- 'bash tools/anchor_profile.sh gpurun_out/anchor_prof --genome-mb 3100 --reads 20000 --families 4000'
ROOT=$(pwd)
OUT=$ROOT/$1
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o k -- python3 "$ROOT/tools/anchor_bench.py" "$@" --cpu-reads 0 --steps 3 \
    > "$OUT/stats_bench.json" 2> "$OUT/kt.err"
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  timeout 900 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc$i" -o p -- python3 "$ROOT/tools/anchor_bench.py" "$@" --cpu-reads 0 --steps 1 \
      > /dev/null 2> "$OUT/pmc$i.err"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc.json" $(find "$OUT"/pmc* -name '*counter_collection.csv') > "$OUT/pmc_top.txt"
rm -rf "$OUT"/kt "$OUT"/pmc[0-9]
head -12 "$OUT/kernel_stats.csv" | cut -c1-170
head -8 "$OUT/pmc_top.txt" | cut -c1-300
