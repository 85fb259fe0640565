#!/bin/bash
# round 6: large parity samples, the soak over other seeds / mixes and the whole-mapper fuzz of the build in place
TAG=${1:-r06}
bash tools/r05_large_parity.sh ${TAG}_parity | tee gpurun_out/${TAG}_parity_large_samples.jsonl
bash tools/soak.sh $TAG > /dev/null 2>&1
cat gpurun_out/${TAG}_soak/summary.txt; cat gpurun_out/${TAG}_soak/fuzz_mapper.jsonl | cut -c1-300
