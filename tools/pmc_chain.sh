#!/bin/bash
# PMC counters of the chain kernels on chance-cluster candidates, bitmap sweep 4 per wavefront (mode 0) and 1 (mode 2)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 0 2; do
  export BLASR_HIP_CHAIN_SCATTER=$mode
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc$mode -o p -- python3 $ROOT/bench.py --reads 30000 --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 --families 0 --spurious 2 > /dev/null 2> $OUT/pmc$mode.err
  python3 $ROOT/tools/pmc_summary.py $OUT/pmc$mode.json $(find $OUT/pmc$mode -name '*counter_collection.csv') | grep -i "chain" | head -8
  rm -rf $OUT/pmc$mode
done
