#!/bin/bash
# PMC passes (separate runs, --pmc only) of the configs[2] workload.  rocprofv3 --pmc dies inside torch's generator
# for this set-up on this image, so the inputs are generated and written to /tmp by an unprofiled run first and the
# profiled runs load them (bench.py --save-workload / --load-workload).  Each pass has its own time limit.
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/r02_pmc_c2}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 --save-workload /tmp/wl_c2.pt > "$OUT/plain.json" 2> "$OUT/plain.err"
ls -la /tmp/wl_c2.pt >> "$OUT/plain.err"
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i + 1))
  timeout 700 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc$i" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 \
      --load-workload /tmp/wl_c2.pt > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err"
  echo "pass $i ($ctr): exit $?" >> "$OUT/passes.txt"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc.json" $(find "$OUT"/pmc* -name '*counter_collection.csv') > "$OUT/pmc_top.txt" 2>> "$OUT/passes.txt"
rm -rf "$OUT"/pmc[0-9] /tmp/wl_c2.pt
cat "$OUT/passes.txt"; cut -c1-300 "$OUT/pmc_top.txt" | head -30
