#!/bin/bash
# the row sweep's routing thresholds at configs[4] (and configs[2]): BLASR_HIP_CHAIN_ROWS="minFrags,densityX2"
OUT=gpurun_out/${1:-r05_rows}
shift
mkdir -p $OUT
python -m pytest tests/test_sdp_gpu.py -x -q -m gpu 2>&1 | tail -2
for cfg in 4 2; do
for th in ${@:-0 4096,6 1024,8 8192,8}; do
  BLASR_HIP_CHAIN_ROWS=$th python bench.py --config $cfg --steps 2 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 300 --detail-out $OUT/c${cfg}_$th.json > /dev/null 2> $OUT/c${cfg}_$th.err
  python - <<P
import json
d=json.load(open("$OUT/c${cfg}_$th.json"))
rows=[r for r in d["kernel_rooflines"] if "rows" in r["kernel"]]
print("config $cfg rows=$th: %.1f ms/step, chain %.1f, parity %s; rows kernel %s" % (d["ms_per_step"], d["stage_ms_per_step"]["sdp_chain"], (d.get("cpu_baseline") or {}).get("parity_mismatches"), [(round(r["ms_per_step"],1), int(r["fragments_per_step"]), int(r["invocations_per_step"])) for r in rows]))
P
done
done
