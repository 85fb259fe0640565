"""The whole mapper against the REFERENCE BINARY on fresh random inputs, as a test: tools/fuzz_mapper_gpu.py in its
modes for a few iterations each (random multi-contig genomes with planted repeats and runs of N, reads from both strands
at several error mixes, mapped from the reads alone -- index built on the device -- with and without -extend):
-m 4 lines under default switches, under one of 27 non-default switches per iteration, SAM records, and the patched
reference binary (-useGPU).  Every line must equal the reference binary's.  profiles/r02_fuzz_mapper*.jsonl hold the long
runs of the same tool."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLASRMC = os.path.join(ROOT, "oracle", "_ref", "blasrmc")
DROPIN = os.path.join(ROOT, "oracle", "_ref", "blasrmc_gpu")


def fuzz(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_mapper_gpu.py"), "--nproc", "8"] + list(args),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return json.loads(p.stdout.decode().strip().splitlines()[-1])


@pytest.mark.skipif(not os.path.exists(BLASRMC), reason="oracle/_ref/blasrmc not built (needs /root/reference at build time)")
@pytest.mark.parametrize("mode", ["default", "variants", "sam", "dropin", "rbao", "fastq", "sam_reads"])
def test_lines_equal_the_reference_binary(mode):
    if mode == "dropin" and not os.path.exists(DROPIN):
        pytest.skip("oracle/_ref/blasrmc_gpu not built")
    args = {"default": ["--iters", "4", "--reads", "100", "--seed", "901"],
            "variants": ["--iters", "27", "--reads", "40", "--seed", "902", "--variants"],
            "sam": ["--iters", "27", "--reads", "40", "--seed", "903", "--variants", "--sam"],
            "dropin": ["--iters", "6", "--reads", "40", "--seed", "904", "--variants", "--dropin"],
            # round 3: -rbao on both sides (every variant, the patched binary in -gpuBatch form), FASTQ and SAM read files
            "rbao": ["--iters", "27", "--reads", "40", "--seed", "905", "--variants", "--rbao", "--dropin"],
            "fastq": ["--iters", "4", "--reads", "80", "--seed", "906", "--fastq", "--sam"],
            "sam_reads": ["--iters", "4", "--reads", "80", "--seed", "907", "--sam-reads"]}[mode]
    d = fuzz(*args)
    assert d["bad_plain"] == 0 and d["bad_extend"] == 0, d["examples"][:2]
    assert d.get("dropin_bad", 0) == 0, d["examples"][:2]
    # (fewer reads come through -rbao's chain of anchor runs; the FASTQ reader of the fork yields every second record)
    assert d["lines_plain"] > (0.4 if mode in ("rbao", "fastq") else 0.7) * d["reads"] and d["changed_by_extend"] > 0.2 * d["lines_plain"]
    if mode in ("dropin", "rbao"):
        assert d["dropin_lines"] > 0
