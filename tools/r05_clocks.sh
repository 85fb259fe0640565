#!/bin/bash
OUT=gpurun_out/${1:-r05_clk}
mkdir -p $OUT
python bench.py --config 2 --steps 1 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 0 --detail-out $OUT/c2.json > $OUT/c2.log 2> $OUT/c2.err
grep "idx clocks" $OUT/c2.err | tail -3
bash tools/ktimeline.sh ${1:-r05_clk}/tl --steps 2 --warmup 1 --other-configs 0 --blasrmc-reads 0 --cpu-sample 0 | tail -90
