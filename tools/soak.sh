#!/bin/bash
# parity soak of the final kernels against the reference's own functions (bench.py's cpu_baseline check) on other
# seeds / mixes than the default line, both refiners; then the whole mapper against the reference binary on random inputs
# usage (repo root): bash tools/soak.sh <tag>
TAG=${1:-r05}
OUT=gpurun_out/${TAG}_soak
mkdir -p $OUT
run() { name=$1; shift; python bench.py --from-reads 0 --other-configs 0 --blasrmc-reads 0 --steps 1 --warmup 0 --detail-out $OUT/$name.json "$@" > $OUT/$name.log 2> $OUT/$name.err; python -c "
import json; d=json.load(open('$OUT/$name.json')); c=d['cpu_baseline']; print('$name', round(d['ms_per_step'],1), c['parity_checked'], c['parity_mismatches'], c['sample'][:90])"; }
{
run c4_seed11 --config 4 --reads 2000 --seed 11 --cpu-sample 400
run c2_spur2_seed12 --config 2 --reads 20000 --spurious 2 --seed 12 --cpu-sample 2500
run c2_seed13 --config 2 --reads 20000 --seed 13 --cpu-sample 4000
run c1_band30 --config 1 --reads 20000 --band 30 --seed 14 --cpu-sample 3000
run c1_2kb --config 1 --reads 50000 --read-len 2000 --seed 15 --cpu-sample 8000
run kband_c1_2kb --config 1 --no-guided --reads 20000 --read-len 2000 --seed 21 --cpu-sample 3000
run kband_c2_5kb --config 2 --no-guided --reads 8000 --read-len 5000 --seed 22 --cpu-sample 600
run kband_c2_10kb --config 2 --no-guided --reads 4000 --seed 23 --cpu-sample 200
run kband_c1_band40 --config 1 --no-guided --kband 40 --reads 20000 --read-len 3000 --seed 24 --cpu-sample 3000
run kband_c1_band200 --config 1 --no-guided --kband 200 --reads 8000 --read-len 4000 --seed 25 --cpu-sample 800
run kband_c4_seed26 --config 4 --no-guided --reads 300 --seed 26 --cpu-sample 16
} | tee $OUT/summary.txt
for seed in 601 602; do python tools/fuzz_mapper_gpu.py --iters 40 --reads 6000 --seed $seed 2>/dev/null | tail -1; done | tee $OUT/fuzz_mapper.jsonl
python tools/fuzz_mapper_gpu.py --iters 20 --reads 2000 --seed 603 --rbao 2>/dev/null | tail -1 | tee -a $OUT/fuzz_mapper.jsonl
python tools/fuzz_mapper_gpu.py --iters 20 --reads 2000 --seed 604 --sam --variants 2>/dev/null | tail -1 | tee -a $OUT/fuzz_mapper.jsonl
