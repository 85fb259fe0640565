#!/bin/bash
# quick check of a build: the SDP tests, then configs[4] and configs[2] without the extras (detail records -> gpurun_out/$1)
OUT=gpurun_out/${1:-r05_q}
mkdir -p $OUT
python -m pytest tests/test_sdp_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for c in 4 2; do
  python bench.py --config $c --steps 2 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample ${2:-600} --detail-out $OUT/c$c.json > $OUT/c$c.log 2> $OUT/c$c.err
  python - <<P
import json
d=json.load(open("$OUT/c$c.json"))
print("config $c: %.1f ms/step, sdp %s, parity %s" % (d["ms_per_step"], {k: round(v,1) for k,v in d["stage_ms_per_step"].items() if k.startswith("sdp")}, (d.get("cpu_baseline") or {}).get("parity_mismatches")))
for r in d["kernel_rooflines"][:9]:
    print("   %-52s %7.1f ms/step  %5.1f launches  conc %s  frags %s" % (r["kernel"][:52], r["ms_per_step"], r["launches_per_step"], r.get("concurrent_launches"), r.get("fragments_per_step")))
P
done
