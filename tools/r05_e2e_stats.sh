#!/bin/bash
# kernel statistics of configs[2]'s end_to_end leg (anchoring + intervals + the path), one pass under rocprofv3
OUT=$(pwd)/gpurun_out/${1:-r05_e2e}
ROOT=$(pwd)
mkdir -p $OUT
python -m pytest tests/test_intervals_gpu.py tests/test_m4_noisy_end_to_end_gpu.py tests/test_rbao_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
BLASR_HIP_IV_TRACE=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks" -o k -- python3 "$ROOT/bench.py" --config 2 --steps 1 --warmup 0 --from-reads 2 \
    --other-configs 0 --blasrmc-reads 0 --cpu-sample 0 --detail-out "$OUT/e2e_detail.json" > "$OUT/e2e.log" 2> "$OUT/e2e.err"
cp "$(find "$OUT/ks" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_end_to_end.csv"; rm -rf "$OUT/ks"
cd $ROOT
grep "\[iv\]" $OUT/e2e.err | tail -7
python3 - <<P
import csv, json
rows=list(csv.DictReader(open("$OUT/kernel_stats_end_to_end.csv")))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('anchor_','iv_','rocprim','ROCPRIM')):
        print("%-100s calls %5s total %8.2f ms"%(n[:100], r['Calls'], float(r['TotalDurationNs'])/1e6))
e=json.load(open("$OUT/e2e_detail.json"))["end_to_end"]
print(e["ms_per_step"], e["stage_ms_per_step"], e["anchor_stage_ms"])
P
