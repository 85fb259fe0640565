#!/bin/bash
mkdir -p gpurun_out/r02b
for e in "X=1" "BLASR_HIP_NO_SIDE_STREAM=1"; do
  env $e python bench.py --steps 2 --warmup 1 --cpu-sample 0 --blasrmc-reads 0 > gpurun_out/r02b/full.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r02b/full.json')); print('$e', d['value'], d['ms_per_step']); print(d['stage_ms_per_step'])"
done
