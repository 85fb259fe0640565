#!/bin/bash
# iteration 3 of the mapper fuzz of seed 604 alone, R times, under switches: usage tools/r06_fzonly.sh <label> <R> [VAR=value ...]
L=$1; R=$2; shift; shift
for v in "$@"; do export "$v"; done
python tools/fuzz_mapper_gpu.py --iters 4 --reads 2000 --seed 604 --sam --variants --only-it 3 --repeat $R 2>&1 | grep "^\[rep" | awk -v l="$L" '{bad += ($4 != 0)} END {print l, "mode-runs with differences:", bad, "of", NR}'
