#!/bin/bash
# PMC passes (separate runs, --pmc only) of one bench.py workload, tagged with the library's build id.
#   usage (repo root):  gpurun --timeout 2400 -- 'bash tools/pmc_config.sh gpurun_out/r03_pmc_c2 --config 2'
# rocprofv3 --pmc dies inside torch's generator for the configs[2] set-up on this image, so the inputs are generated
# and saved by an unprofiled run first and the profiled runs load them (bench.py --save-workload / --load-workload).
# The saved workload (4 GB) lives in a private directory made by mktemp (unpredictable name, mode 700, removed at the
# end), not at a fixed path in /tmp and not under gpurun_out/ (which is copied back).  Each pass has its own time limit.
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/pmc}
ARGS="${@:2}"
mkdir -p "$OUT"
WL=$(mktemp -d /tmp/blasr_wl.XXXXXX)
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 $ARGS --save-workload "$WL/wl.pt" > "$OUT/plain.json" 2> "$OUT/plain.err"
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i + 1))
  timeout 700 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc$i" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 \
      $ARGS --load-workload "$WL/wl.pt" > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err"
  echo "pass $i ($ctr): exit $?" >> "$OUT/passes.txt"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc.json" --bench-line "$OUT/plain.json" $(find "$OUT"/pmc* -name '*counter_collection.csv') > "$OUT/pmc_top.txt" 2>> "$OUT/passes.txt"
rm -rf "$OUT"/pmc[0-9] "$WL"
cat "$OUT/passes.txt"; cut -c1-300 "$OUT/pmc_top.txt" | head -30
