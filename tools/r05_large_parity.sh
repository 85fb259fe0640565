#!/bin/bash
# bench.py's own check on larger samples than the default line's (every hit of every read against the reference's own functions)
OUT=gpurun_out/${1:-r05_parity}
mkdir -p $OUT
run() { name=$1; shift; python bench.py --from-reads 0 --other-configs 0 --blasrmc-reads 0 --steps 1 --warmup 0 --detail-out $OUT/$name.json "$@" > $OUT/$name.log 2> $OUT/$name.err; python -c "
import json; d=json.load(open('$OUT/$name.json')); c=d['cpu_baseline']; print(json.dumps({'run': '$name', 'build_id': d['build_id'], 'ms_per_step': round(d['ms_per_step'],1), 'reads_checked': c['parity_checked'], 'reads_that_differ': c['parity_mismatches'], 'cpu_cores': c['cores']}))"; }
run configs2_12000_reads --config 2 --cpu-sample 12000
run configs1_12000_reads --config 1 --cpu-sample 12000
run configs4_all_6250_reads --config 4 --cpu-sample 6250
run configs4_noguided_160_reads --config 4 --no-guided --cpu-sample 160
