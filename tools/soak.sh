#!/bin/bash
# parity soak of the final kernels against the reference's own functions (bench.py's cpu_baseline check) on other
# seeds / mixes than the default line
mkdir -p gpurun_out/r03_soak
run() { name=$1; shift; python bench.py "$@" > gpurun_out/r03_soak/$name.json 2> gpurun_out/r03_soak/$name.err; python -c "
import json; d=json.load(open('gpurun_out/r03_soak/$name.json')); c=d['cpu_baseline']; print('$name', d['ms_per_step'], c['parity_checked'], c['parity_mismatches'], c['sample'][:70])"; }
run c4_seed11 --config 4 --reads 2000 --seed 11 --steps 1 --warmup 0 --cpu-sample 400 --blasrmc-reads 0
run c2_spur2_seed12 --config 2 --reads 20000 --spurious 2 --seed 12 --steps 1 --warmup 0 --cpu-sample 2500 --blasrmc-reads 0
run c2_seed13 --config 2 --reads 20000 --seed 13 --steps 1 --warmup 0 --cpu-sample 4000 --blasrmc-reads 0
run c1_band30 --config 1 --reads 20000 --band 30 --seed 14 --steps 1 --warmup 0 --cpu-sample 3000 --blasrmc-reads 0
run c1_2kb --config 1 --reads 50000 --read-len 2000 --seed 15 --steps 1 --warmup 0 --cpu-sample 8000 --blasrmc-reads 0
