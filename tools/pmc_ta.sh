#!/bin/bash
# Where the memory path of a kernel is busy: texture addresser / L1 (TA, TCP) against the time the GPU is active, per kernel,
# for one bench.py workload.  usage (repo root): gpurun -- 'bash tools/pmc_ta.sh gpurun_out/r04_ta --config 2'
ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/ta}; ARGS="${@:2}"
mkdir -p "$OUT"; WL=$(mktemp -d /tmp/blasr_wl.XXXXXX); cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 --from-reads 0 --other-configs 0 $ARGS --save-workload "$WL/wl.pt" > "$OUT/plain.json" 2> "$OUT/plain.err"
i=0
for ctr in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i + 1))
  timeout 900 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/p$i" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 \
    --cpu-sample 0 --blasrmc-reads 0 --from-reads 0 --other-configs 0 $ARGS --load-workload "$WL/wl.pt" > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err"
  echo "pass $i exit $?"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        per[n][r["Counter_Name"]] += float(r["Counter_Value"])
res = {n: dict(c) for n, c in per.items() if "blasr_hip" in n}
json.dump(res, open(out + "/ta.json", "w"), indent=1, sort_keys=True)
for n in sorted(res, key=lambda n: -res[n].get("GRBM_GUI_ACTIVE", 0))[:12]:
    print(n[:50], {k: round(v / 1e6, 1) for k, v in sorted(res[n].items())})
PY
rm -rf "$OUT"/p[0-9] "$WL"
tail -3 "$OUT"/pmc1.err
