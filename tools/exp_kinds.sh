#!/bin/bash
# which kind of candidate interval costs what: bench.py on each mix, SDP trace on stderr
mkdir -p gpurun_out/r02b
for opt in "--families 0 --spurious 0" "--families 0 --spurious 2" "--families 4000 --spurious 0"; do
  echo "== $opt"
  BLASR_HIP_SDP_TRACE=1 python bench.py --reads 5000 --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 $opt \
    > gpurun_out/r02b/t.json 2> gpurun_out/r02b/t.err
  python -c "
import json; d=json.load(open('gpurun_out/r02b/t.json')); print(d['config']['candidate_intervals_per_gpu'], d['stage_ms_per_step'], d['sdp_counters_per_step'])"
  grep "\[sdp\]" gpurun_out/r02b/t.err | head -8
done
