#!/bin/bash
# configs[2] (no extras) under a list of environment settings:  tools/r05_env_ab.sh TAG "A=1" "B=2" ...
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for sw in "NONE=1" "$@"; do
  tag=${sw//=/_}; tag=${tag//,/_}
  env $sw python bench.py --config ${CONFIG:-2} --steps 2 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 300 --detail-out $OUT/$tag.json > $OUT/$tag.log 2> $OUT/$tag.err
  python - <<P
import json
d=json.load(open("$OUT/$tag.json"))
k={r["kernel"].split("::")[-1][:28]: round(r["ms_per_step"],1) for r in d["kernel_rooflines"][:8]}
print("%-34s %.1f ms/step, sdp %s, parity %s  %s" % ("$sw", d["ms_per_step"], {a[4:]: round(v,1) for a,v in d["stage_ms_per_step"].items() if a.startswith("sdp_")}, (d.get("cpu_baseline") or {}).get("parity_mismatches"), k))
P
done
