#!/bin/bash
# Read requests of the L2 towards memory by size (32 / 64 / 128 bytes) per kernel, for one bench.py workload: settles what
# FETCH_SIZE means for a kernel's access pattern (the guide doubles it for 16-byte-per-lane streaming reads only).
#   usage (repo root): gpurun -- 'bash tools/pmc_rdreq.sh gpurun_out/r04_rdreq --config 2'
ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/rdreq}; ARGS="${@:2}"
mkdir -p "$OUT"; WL=$(mktemp -d /tmp/blasr_wl.XXXXXX); cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 --from-reads 0 --other-configs 0 $ARGS --save-workload "$WL/wl.pt" > "$OUT/plain.json" 2> "$OUT/plain.err"
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B --output-format csv -d "$OUT/p" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 \
  --cpu-sample 0 --blasrmc-reads 0 --from-reads 0 --other-configs 0 $ARGS --load-workload "$WL/wl.pt" > "$OUT/pmc.json" 2> "$OUT/pmc.err"
echo "exit $?"
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        per[n][r["Counter_Name"]] += float(r["Counter_Value"])
res = {n: dict(c) for n, c in per.items() if "blasr_hip" in n}
json.dump(res, open(out + "/rdreq.json", "w"), indent=1, sort_keys=True)
for n in sorted(res, key=lambda n: -sum(res[n].values()))[:10]:
    print(n[:60], {k: round(v / 1e6, 1) for k, v in res[n].items()})
PY
rm -rf "$OUT/p" "$WL"
