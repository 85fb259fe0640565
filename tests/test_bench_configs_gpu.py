"""BASELINE.json's configs as the bench generates them (blasr_amd/synth.py: make_repeat_genome / make_map_workload),
under the driver's GPU tests: a reduced number of reads of configs[2] (hg19-size genome with planted repeat families,
competing repeat copies, chance clusters) and of configs[4] (30 kb reads at 15 % indels with eight long chance clusters
per read -- the scattered-sweep regime of the SDP stage), every hit of every read compared with the reference's own
SDPAlign / AffineGuidedAlign / ComputeAlignmentStats (oracle/_ref/libblasr_ref.so; the oracle's restatement where that
build is absent) composed into MapRead by the checker.  The same run pins the shape of the bench output: the driver's line
(the last one, < 4 KB) and the full record beside it, whose `roofline` names the kernel with the largest measured time of
either stage; all of them follow in `kernel_rooflines`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv):
    """One run of bench.py: the driver's line (the LAST stdout line; the only one that starts like a JSON object) is checked
    for its size, and the full record the run left in --detail-out is what the tests look into."""
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "detail.json")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--blasrmc-reads", "0",
                            "--detail-out", detail] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        out = p.stdout.decode().splitlines()
        lines = [x for x in out if x.startswith("{")]
        assert len(lines) == 1 and out[-1] == lines[0]
        assert len(lines[0]) < 4096, len(lines[0])  # the driver reads it out of an 8 KB tail (round 4's was 20 KB: unparsed)
        line = json.loads(lines[0])
        d = json.load(open(detail))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "build_id"):
        assert k in line, k
    assert line["value"] == pytest.approx(d["value"], rel=1e-3) and line["roofline"]["kernel"] == d["roofline"]["kernel"]
    assert line["cpu_baseline"]["parity_mismatches"] == d["cpu_baseline"]["parity_mismatches"]
    return d


def check_line(d, n_reads):
    cb = d["cpu_baseline"]
    assert cb["parity_checked"] == n_reads and cb["parity_mismatches"] == 0, cb
    assert d["config"]["aligned_reads"] == n_reads
    # the line names the kernel that took the longest -- of either stage -- with the bytes of exactly what it processed
    lines = d["kernel_rooflines"]
    cost = [r["avg_launch_ms"] * r["launches_per_step"] for r in lines]
    assert d["roofline"]["kernel"] == lines[0]["kernel"] and cost[0] == max(cost)
    assert all(r["achieved"] > 0 and r["algorithmic_bytes_per_launch"] > 0 for r in lines), lines
    names = [r["kernel"] for r in lines]
    assert any(n.startswith("blasr_hip::traceback_kernel") for n in names) and any("fillh8_kernel" in n for n in names)
    # the SDP stage's kernels are in the same list, timed on their own streams
    sdp = [r for r in lines if r["stage"] == "sdp"]
    assert any("sdp_chain" in r["kernel"] for r in sdp) and any("sdp_match" in r["kernel"] for r in sdp)
    assert all(r["fragments_per_step"] > 0 for r in sdp if "chain" in r["kernel"])
    chain_ms = sum(r["ms_per_step"] for r in sdp if "sdp_chain" in r["kernel"])
    # (side by side on four streams: their sum is at least the phase's wall time less the list's sort, and not absurdly more)
    assert 0.5 * d["stage_ms_per_step"]["sdp_chain"] <= chain_ms + 5.0 and chain_ms <= 4.5 * d["stage_ms_per_step"]["sdp_chain"] + 5.0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / 8000.0) < 1e-12
    # traffic only from a PMC profile of this very build and workload
    assert d["roofline"]["traffic"] is None or d["roofline"]["pmc_profile"].startswith("profiles/")
    assert len(d["build_id"]) == 12


def test_configs2_generator_against_the_reference_functions():
    d = run_bench("--config", "2", "--reads", "1500", "--cpu-sample", "1500")
    assert "configs[2]" in d["config"]["workload"] and d["config"]["candidate_intervals_per_gpu"] > 1500
    check_line(d, 1500)
    assert any("fillh8_kernel" in r["kernel"] for r in d["kernel_rooflines"])  # (one queue of all candidates, wide rows or not)


def test_configs4_generator_against_the_reference_functions():
    d = run_bench("--config", "4", "--reads", "60", "--cpu-sample", "60")
    assert "configs[4]" in d["config"]["workload"] and d["config"]["candidate_intervals_per_gpu"] >= 60 * 8
    check_line(d, 60)


def test_configs1_generator_against_the_reference_functions():
    """configs[1]: 10 kb reads against an E. coli-sized genome, one candidate per read (the narrow fill only)."""
    d = run_bench("--config", "1", "--reads", "2000", "--cpu-sample", "2000")
    assert "configs[1]" in d["config"]["workload"] and d["config"]["candidate_intervals_per_gpu"] == 2000
    cb = d["cpu_baseline"]
    assert cb["parity_checked"] == 2000 and cb["parity_mismatches"] == 0, cb
    assert d["config"]["aligned_reads"] == 2000 and d["roofline"]["bound"] in ("hbm", "mfma")


def test_configs4_wide_kband_form_against_the_reference_templates():
    """configs[4] as `-noUseGuidedAlign` runs it ("wide k-band"): 30 kb reads, the survivors through RefineAlignment's k-band
    branch (k ~ 2.7 k from the SDP alignment's drift) inside the timed step; every hit against the same branch run with the
    reference's own AffineKBandAlign / ComputeAlignmentStats templates."""
    d = run_bench("--config", "4", "--no-guided", "--reads", "24", "--cpu-sample", "12")
    assert "-noUseGuidedAlign" in d["config"]["workload"] and d["config"]["refiner"] == "AffineKBandAlign"
    cb = d["cpu_baseline"]
    assert cb["parity_checked"] == 12 and cb["parity_mismatches"] == 0, cb
    assert d["config"]["aligned_reads"] == 24
    lines = d["kernel_rooflines"]
    kb = [r for r in lines if r["kernel"] == "blasr_hip::kband_wide_kernel"][0]
    assert kb["bound"] == "hbm" and kb.get("limiter") in (None, "valu issue", "valu issue (modelled)")
    assert kb["kernel_gcups"] > 0 and any(r["kernel"] == "blasr_hip::kband_wide_traceback_kernel" for r in lines)
    # bands of the width the config is named for
    assert kb["algorithmic_bytes_per_launch"] / kb["candidates_per_launch"] > 30000 * 2000 / 2
