#!/bin/bash
# the anchor stage after a change: its tests, the anchoring fuzz, then configs[2]'s end_to_end leg (serial form), optionally
# with an environment switch for comparison:  tools/r05_anchor.sh TAG "VAR=1"
OUT=gpurun_out/${1:-r05_anchor}
mkdir -p $OUT
python -m pytest tests/test_anchor_gpu.py tests/test_intervals_gpu.py tests/test_rbao_gpu.py tests/test_m4_noisy_end_to_end_gpu.py tests/test_example_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
python tools/fuzz_anchoring_gpu.py 77 ${3:-60} > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log
for sw in "" $2; do
  tag=${sw:-default}; tag=${tag//=/_}
  env $sw python bench.py --config 2 --steps 2 --warmup 1 --from-reads 2 --other-configs 0 --blasrmc-reads 0 --cpu-sample 200 --detail-out $OUT/e_$tag.json > $OUT/e_$tag.log 2> $OUT/e_$tag.err
  python - <<P
import json
d=json.load(open("$OUT/e_$tag.json"))
e=d["end_to_end"]
print("$tag: end_to_end %.1f ms  stages %s  anchor %s" % (e["ms_per_step"], {k: round(v,1) for k,v in e["stage_ms_per_step"].items()}, e["anchor_stage_ms"]))
P
done
