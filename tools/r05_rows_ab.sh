#!/bin/bash
OUT=gpurun_out/${1:-r05_rows_ab}
mkdir -p $OUT
for rep in 1 2 3; do
for th in 0 1024,8 4096,8; do
  BLASR_HIP_CHAIN_ROWS=$th python bench.py --config 4 --steps 3 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 0 --detail-out $OUT/c4_$th.json > /dev/null 2> $OUT/c4_$th.err
  python - <<P
import json
d=json.load(open("$OUT/c4_$th.json"))
print("rep $rep rows=$th: %.1f ms/step, chain %.1f" % (d["ms_per_step"], d["stage_ms_per_step"]["sdp_chain"]))
P
done
done
