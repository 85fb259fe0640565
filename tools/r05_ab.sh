#!/bin/bash
# A/B of one environment switch on configs[2] (and the SDP tests first): usage tools/r05_ab.sh <outdir> <ENV=1> [config]
OUT=gpurun_out/${1:-r05_ab}
SW=${2:-BLASR_HIP_NONE=1}
CFG=${3:-2}
mkdir -p $OUT
python -m pytest tests/test_sdp_gpu.py tests/test_full_path_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for v in new old; do
  if [ $v = old ]; then export $SW; fi
  python bench.py --config $CFG --steps 2 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 1200 --detail-out $OUT/$v.json > $OUT/$v.log 2> $OUT/$v.err
  grep "idx clocks" $OUT/$v.err | head -1
  python - <<P
import json
d=json.load(open("$OUT/$v.json"))
print("$v: %.1f ms/step, sdp %s, parity %s" % (d["ms_per_step"], {k: round(v,1) for k,v in d["stage_ms_per_step"].items() if k.startswith("sdp")}, (d.get("cpu_baseline") or {}).get("parity_mismatches")))
for r in d["kernel_rooflines"][:7]:
    print("   %-52s %7.1f ms/step  %5.1f launches  conc %s" % (r["kernel"][:52], r["ms_per_step"], r["launches_per_step"], r.get("concurrent_launches")))
P
done
