#!/bin/bash
# round 5, first GPU call: the new bench output, the SDP kernel records, the fragments-per-row trace
OUT=gpurun_out/r05_a
mkdir -p $OUT
python -m pytest tests/test_bench_configs_gpu.py tests/test_sdp_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
python bench.py --detail-out $OUT/default_detail.json > $OUT/default.log 2> $OUT/default.err; echo "bench rc $?"
tail -c 3000 $OUT/default.log
BLASR_HIP_SDP_TRACE=1 python bench.py --steps 1 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 0 --detail-out $OUT/trace2_detail.json > $OUT/trace2.log 2> $OUT/trace2.err
BLASR_HIP_SDP_TRACE=1 python bench.py --config 4 --steps 1 --warmup 1 --from-reads 0 --other-configs 0 --blasrmc-reads 0 --cpu-sample 0 --detail-out $OUT/trace4_detail.json > $OUT/trace4.log 2> $OUT/trace4.err
grep "category\|sweep" $OUT/trace4.err | tail -40
