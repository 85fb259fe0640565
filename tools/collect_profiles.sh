#!/bin/bash
# The round's evidence in one GPU call: kernel statistics of the timed step (rocprofv3 --kernel-trace --stats), the workspace
# curve, and the PMC passes of the configurations' dominant kernels.   usage (repo root): bash tools/collect_profiles.sh <tag>
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. kernel statistics, configs[2] (the driver's workload), five timed steps
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks2" -o k -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --from-reads 0 \
    --other-configs 0 --blasrmc-reads 0 --cpu-sample 0 --detail-out "$OUT/bench_configs2_under_rocprof_detail.json" > "$OUT/bench_configs2_under_rocprof.json" 2> "$OUT/ks2.err"
cp "$(find "$OUT/ks2" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_configs2_100k_reads.csv"; rm -rf "$OUT/ks2"
# 2. the same for configs[4] -noUseGuidedAlign (the k-band refiner), two timed steps
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks4" -o k -- python3 "$ROOT/bench.py" --config 4 --no-guided --steps 2 --warmup 1 \
    --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 0 --detail-out "$OUT/bench_configs4_noguided_under_rocprof_detail.json" > "$OUT/bench_configs4_noguided_under_rocprof.json" 2> "$OUT/ks4.err"
cp "$(find "$OUT/ks4" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_configs4_noguided_6250_reads.csv"; rm -rf "$OUT/ks4"
# 3. the workspace curve of configs[2]
cd "$ROOT"
for ws in 32 64 96 128 200; do
  python3 bench.py --steps 3 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 0 --workspace-gb $ws 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'workspace_gb': $ws, 'ms_per_step': d['ms_per_step'], 'held_gb': d['device_memory_held_gb'], 'sdp_ms': d['stage_ms_per_step']['sdp'], 'refine_ms': d['stage_ms_per_step']['refine'], 'build_id': d['build_id']}))"
done > "$OUT/workspace_curve_configs2.jsonl"
cat "$OUT/workspace_curve_configs2.jsonl"
