#!/bin/bash
# round 6, first look at a build: the whole -m gpu suite, then configs[2] and configs[4] without the extras
OUT=gpurun_out/${1:-r06_a}
mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $OUT/gpu_tests.log; tail -3 $OUT/gpu_tests.log
for c in 2 4; do
  python bench.py --config $c --steps 3 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 600 --detail-out $OUT/c$c.json > $OUT/c$c.log 2> $OUT/c$c.err
  python - <<P
import json
d=json.load(open("$OUT/c$c.json"))
print("config $c: %.1f ms/step, stages %s, parity %s" % (d["ms_per_step"], {k: round(v,1) for k,v in d["stage_ms_per_step"].items()}, (d.get("cpu_baseline") or {}).get("parity_mismatches")))
for r in d["kernel_rooflines"][:12]:
    print("   %-52s %7.1f ms/step  %5.1f launches  frac %.4f" % (r["kernel"][:52], r["ms_per_step"], r["launches_per_step"], r["frac"]))
P
done
