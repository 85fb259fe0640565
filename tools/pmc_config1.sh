#!/bin/bash
# PMC passes (separate runs, --pmc only) of the configs[1] workload: per-kernel HBM traffic and SQ counters
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/r02_pmc_c1}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i + 1))
  rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc$i" -o p -- python3 "$ROOT/bench.py" --config 1 --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 \
      > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc.json" $(find "$OUT"/pmc* -name '*counter_collection.csv') > "$OUT/pmc_top.txt"
rm -rf "$OUT"/pmc[0-9]
cat "$OUT/pmc_top.txt" | cut -c1-400
