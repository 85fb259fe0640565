#!/bin/bash
# chain kernels on chance-cluster candidates: default split / all wide / all bitmap
mkdir -p gpurun_out/r02b
for mode in 0; do
  echo "== BLASR_HIP_CHAIN_SCATTER=$mode"
  BLASR_HIP_CHAIN_SCATTER=$mode python bench.py --reads 30000 --steps 1 --warmup 1 --cpu-sample 0 --blasrmc-reads 0 --families 0 --spurious 2 \
    > gpurun_out/r02b/t.json 2> gpurun_out/r02b/t.err
  python -c "
import json; d=json.load(open('gpurun_out/r02b/t.json')); print(d['stage_ms_per_step']['sdp_chain'], d['stage_ms_per_step']['sdp'])"
done
