#!/bin/bash
# the round's evidence for one build: full -m gpu suite, the driver's command, kernel statistics + workspace curve, PMC passes
TAG=${1:-r06_v1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc $?" | tee -a $OUT/gpu_tests.log; tail -2 $OUT/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $OUT/bench_default_detail.json > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench rc $?"
tail -1 $OUT/bench_default.log | wc -c
tail -1 $OUT/bench_default.log > $OUT/bench_default_line.json
bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1
bash tools/pmc_config.sh gpurun_out/${TAG}_pmc_c2 --config 2 --from-reads 0 --other-configs 0 > $OUT/pmc_c2.log 2>&1
bash tools/pmc_config.sh gpurun_out/${TAG}_pmc_c4 --config 4 --from-reads 0 --other-configs 0 > $OUT/pmc_c4.log 2>&1
tail -3 $OUT/pmc_c4.log | cut -c1-200
