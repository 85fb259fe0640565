#!/bin/bash
# kernel trace of one end-to-end pass (bench.py --from-reads 1): GPU idle gaps and kernel time inside the anchoring phase
ROOT=$(pwd)
OUT=/tmp/e2e_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o k -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 --from-reads 1 > $OUT/bench.json 2> $OUT/err.txt
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-48:]) for r in rows)
# the last end-to-end pass: from the first anchor_search_kernel of the last run of them to the last kernel
idx = [i for i, e in enumerate(ev) if "anchor_search" in e[2]]
# passes: the untimed one and the timed one -> take the second half of the search launches
half = idx[len(idx) // 2]
t0 = ev[half][0]
last_anchor = max(e[1] for e in ev if e[0] >= t0 and ("anchor_" in e[2]))
seg = [e for e in ev if e[0] >= t0 and e[1] <= last_anchor]
busy, cur, gaps = 0, t0, []
per = {}
for s, e, n in seg:
    per[n] = per.get(n, 0) + (e - s)
    if s > cur:
        gaps.append((s - cur, n)); busy += e - s; cur = e
    elif e > cur:
        busy += e - cur; cur = e
print("anchoring phase of the last pass: %.1f ms, GPU busy %.1f ms, idle %.1f ms in %d gaps (> 1 ms: %d, sum %.1f ms)" % (
    (last_anchor - t0) / 1e6, busy / 1e6, (last_anchor - t0 - busy) / 1e6, len(gaps), sum(1 for g in gaps if g[0] > 1e6), sum(g[0] for g in gaps if g[0] > 1e6) / 1e6))
for n, t in sorted(per.items(), key=lambda x: -x[1])[:10]:
    print("%9.1f ms  %s" % (t / 1e6, n))
big = sorted(gaps, reverse=True)[:8]
print("largest gaps (ms, before kernel):", [(round(g[0] / 1e6, 2), g[1][-30:]) for g in big])
PY
