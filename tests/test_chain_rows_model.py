"""Row / batch formulations of SDPLongestCommonSubsequence (sdp/SparseDynamicProgramming.h:70-322) against the reference
loop, on the CPU: tests/chain_rows_model.c states (a) the reference's fragment-at-a-time loop, pinned to the oracle's
restatement through the reported chain, (b) the sweep with a whole ROW scored and a whole trail row retired per step,
(c) the sweep as batches of retire / query EVENTS evaluated side by side with conflict detection -- and compares every
fragment's cost, chainPrev and chainLength.  Both formulations are exact; the counters the program prints are why the
device keeps the fragment-at-a-time kernels for the bulk of the scattered invocations (one to four
fragments per row, a first conflict after about six events)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "chain_rows_model")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libblasr_oracle.so"])
    subprocess.check_call(["gcc", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "chain_rows_model.c"),
                           "-L", os.path.join(ROOT, "oracle"), "-lblasr_oracle", "-lm", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return exe


def test_row_and_batch_formulations_equal_the_reference_sweep(tmp_path):
    exe = build(tmp_path)
    out = subprocess.run([exe, "60"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    d = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert d["failures"] == 0 and d["cases"] == 120 and d["fragments"] > 50000
    assert d["widest_row"] >= 8 and d["in_row_above"] > 0  # dense rows and the in-row `above` dependency were exercised
    assert d["rounds_per_batch"] >= 1.0


def test_scattered_invocation_of_the_recursion_levels(tmp_path):
    """A chance-match rectangle as the recursion's Global calls see it (k = 7, ~2 fragments per row)."""
    exe = build(tmp_path)
    out = subprocess.run([exe, "1", "0", "1500", "20000", "7"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    d = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert d["failures"] == 0 and 0.5 < d["fragments_per_row"] < 4
