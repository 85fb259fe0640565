#!/bin/bash
OUT=gpurun_out/r05_filt; mkdir -p $OUT
for rep in 1 2; do for x in 4 2 1; do
  BLASR_HIP_IDX_FILTER_X2=$x python bench.py --config 4 --steps 3 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 200 --detail-out $OUT/c4_$x.json > /dev/null 2> $OUT/c4_$x.err
  python - <<P
import json
d=json.load(open("$OUT/c4_$x.json"))
mi=[r for r in d["kernel_rooflines"] if "match_index" in r["kernel"]]
print("rep $rep filterX2=$x: %.1f ms/step, match_order %.1f, parity %s, match_index %.1f" % (d["ms_per_step"], d["stage_ms_per_step"]["sdp_match_order"], d["cpu_baseline"]["parity_mismatches"], mi[0]["ms_per_step"]))
P
done; done
