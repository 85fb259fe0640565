#!/bin/bash
# the overlapped whole mapper (blasr_hip_pipeline_params.overlap): its tests, then configs[2]'s end_to_end leg for a few
# sub-block sizes beside the serial form (detail records -> gpurun_out/$1)
OUT=gpurun_out/${1:-r05_pipe}
mkdir -p $OUT
python -m pytest tests/test_anchor_gpu.py tests/test_rbao_gpu.py tests/test_m4_noisy_end_to_end_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for pipe in ${2:-100000,0 25000,1 12500,1 50000,1 25000,0}; do
  tag=${pipe/,/_}
  BLASR_HIP_PIPE=$pipe python bench.py --config 2 --steps 2 --warmup 1 --from-reads 2 --other-configs 0 --blasrmc-reads 0 --cpu-sample 200 --detail-out $OUT/p$tag.json > $OUT/p$tag.log 2> $OUT/p$tag.err
  python - <<P
import json
d=json.load(open("$OUT/p$tag.json"))
e=d["end_to_end"]
print("pipe $pipe: end_to_end %.1f ms  stages %s  anchor %s  same_as_serial %s  held %.1f GB  (step %.1f)" % (e["ms_per_step"], {k: round(v,1) for k,v in e["stage_ms_per_step"].items()}, e["anchor_stage_ms"], e.get("hits_equal_the_serial_form"), d["device_memory_held_gb"], d["ms_per_step"]))
P
done
