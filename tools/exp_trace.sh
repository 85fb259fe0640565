#!/bin/bash
mkdir -p gpurun_out/r02b
BLASR_HIP_SDP_TRACE=1 python bench.py --reads 30000 --steps 1 --warmup 0 --cpu-sample 0 --blasrmc-reads 0 --families 0 --spurious 2 > gpurun_out/r02b/t.json 2> gpurun_out/r02b/t.err
grep "\[sdp\]" gpurun_out/r02b/t.err | grep -v LDS
