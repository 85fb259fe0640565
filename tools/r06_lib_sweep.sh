#!/bin/bash
# library variants (blasr_amd/libblasr_hip_<name>.so, "base" = the library in place) at a config: usage tools/r06_lib_sweep.sh <outdir> <config> name1 name2 ... (each twice)
OUT=gpurun_out/${1:-r06_lib}; CFG=$2; shift; shift
mkdir -p $OUT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "base" ]; then unset BLASR_HIP_LIBRARY; else export BLASR_HIP_LIBRARY=$PWD/blasr_amd/libblasr_hip_$v.so; fi
  python bench.py --config $CFG --steps 3 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 300 --detail-out $OUT/c${CFG}_${v}_$rep.json > $OUT/c${CFG}_${v}_$rep.log 2> $OUT/c${CFG}_${v}_$rep.err
  python - <<P
import json
d=json.load(open("$OUT/c${CFG}_${v}_$rep.json"))
st=d["stage_ms_per_step"]
ks=[(r["kernel"].split("::")[-1][-22:], round(r["ms_per_step"],1)) for r in d["kernel_rooflines"] if "chain" in r["kernel"]]
print("config $CFG library $v: %.1f ms/step, sdp %.1f match+order %.1f sweep %.1f blocks %.1f, parity %s/%s  %s" % (d["ms_per_step"], st["sdp"], st["sdp_match_order"], st["sdp_chain"], st["sdp_blocks_fills_assembly"], (d.get("cpu_baseline") or {}).get("parity_mismatches"), (d.get("cpu_baseline") or {}).get("parity_checked"), ks))
P
done
done
