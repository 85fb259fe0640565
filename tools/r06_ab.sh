#!/bin/bash
# A/B of one environment switch (old = with the switch): refine + full-path tests first, then configs given (default 2 4)
# usage: tools/r06_ab.sh <outdir> <ENV=1> [configs...]
OUT=gpurun_out/${1:-r06_ab}
SW=${2:-BLASR_HIP_NONE=1}
shift; shift
CFGS=${@:-2 4}
mkdir -p $OUT
python -m pytest tests/test_refine_gpu.py tests/test_full_path_gpu.py tests/test_mapread_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for c in $CFGS; do
for v in new old new old; do
  if [ $v = old ]; then export $SW; else unset ${SW%%=*}; fi
  python bench.py --config $c --steps 3 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 1200 --detail-out $OUT/c${c}_$v.json > $OUT/c${c}_$v.log 2> $OUT/c${c}_$v.err
  python - <<P
import json
d=json.load(open("$OUT/c${c}_$v.json"))
st=d["stage_ms_per_step"]
print("config $c $v: %.1f ms/step, sdp %.1f refine %.1f fill %.1f tb %.1f, parity %s/%s" % (d["ms_per_step"], st["sdp"], st["refine"], st["refine_fill"], st["refine_traceback_stats"], (d.get("cpu_baseline") or {}).get("parity_mismatches"), (d.get("cpu_baseline") or {}).get("parity_checked")))
for r in d["kernel_rooflines"]:
    if r.get("stage") == "refine": print("   %-52s %7.1f ms/step  frac %.4f" % (r["kernel"][:52], r["ms_per_step"], r["frac"]))
P
done
done
