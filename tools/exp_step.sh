#!/bin/bash
# quick check of a change: the SDP and mapper tests, two short configs[2] runs with parity on 600 reads, and the
# timeline of one step (kernels matching $1)
python -m pytest tests/test_sdp_gpu.py tests/test_mapread_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
  python bench.py --steps 3 --warmup 1 --cpu-sample 600 --blasrmc-reads 0 --from-reads 0 > gpurun_out/st_$i.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/st_$i.json').read().strip().splitlines()[-1])
s=d['stage_ms_per_step']
print('step', round(d['ms_per_step'],1), 'sdp', round(s['sdp'],1), 'match_order', round(s['sdp_match_order'],1), 'chain', round(s['sdp_chain'],1), 'blocks', round(s['sdp_blocks_fills_assembly'],1), 'refine', round(s['refine'],1), 'parity', d['cpu_baseline']['parity_checked'], d['cpu_baseline']['parity_mismatches'])
"
done
bash tools/ktimeline.sh gpurun_out/t6 --steps 2 --warmup 1 --cpu-sample 0 --blasrmc-reads 0 > /dev/null 2>&1
grep -E "${1:-sdp_match}" gpurun_out/t6/timeline.txt | head -12 | cut -c1-100
