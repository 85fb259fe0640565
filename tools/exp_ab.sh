#!/bin/bash
# A/B of an environment switch: step and SDP stages at configs[2] and [4].  usage: exp_ab.sh VAR val1 val2 ...
V=$1; shift
for C in 2 4; do
  for L in "$@"; do
    env $V=$L python bench.py --config $C --steps 3 --warmup 1 --cpu-sample 0 --blasrmc-reads 0 > gpurun_out/ab_${C}_$L.json 2>/dev/null
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab_${C}_$L.json').read().strip().splitlines()[-1])
s=d['stage_ms_per_step']
print('config $C $V=$L step', round(d['ms_per_step'],1), 'sdp', round(s['sdp'],1), 'chain', round(s['sdp_chain'],1))
"
  done
done
