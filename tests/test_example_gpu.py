"""examples/map_fasta: a C++ program over the C ABI alone -- the library's FASTA readers (next-4), MapRead for the whole
block (blasr_hip_map_reads_host), contig location, mapQV, the -m 4 printer -- run as a binary on the noisy
multi-candidate fixture: FASTA files and the reference's candidate intervals in, the reference binary's -m 4 lines
out, byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from test_dropin_gpu import write_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "map_fasta")
GOLDEN = os.path.join(ROOT, "tests", "golden", "m4_noisy_golden.npz")


@pytest.mark.gpu
def test_cpp_example_prints_the_reference_lines(tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("examples/map_fasta not built (python -c 'import __graft_entry__ as g; g.build()')")
    want = write_inputs(str(tmp_path))
    g = np.load(GOLDEN)
    with open(os.path.join(str(tmp_path), "intervals.txt"), "w") as f:
        for (s, e, qs, qe, strand), r in zip(g["intervals"], g["owner"]):
            f.write("%d %d %d %d %d %d\n" % (r, s, e, qs, qe, strand))
    p = subprocess.run([EXE, os.path.join(str(tmp_path), "reads.fasta"), os.path.join(str(tmp_path), "genome.fasta"),
                        os.path.join(str(tmp_path), "intervals.txt"), "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = sorted(p.stdout.decode().splitlines())
    assert len(got) == len(want) == 80
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert not bad, bad[:3]


@pytest.mark.gpu
@pytest.mark.parametrize("extend", [False, True])
def test_cpp_example_maps_from_fasta_and_suffix_array_alone(tmp_path, extend):
    """The same binary as a whole mapper: reads.fasta + genome.fasta + the .sa file of the reference's sawriter in, the
    reference binary's -m 4 lines out (anchoring on the device, interval finding on the host, then MapRead); with
    -extend, the lines the reference binary prints under -extend (tests/golden/m4_noisy_extend_golden.npz)."""
    sawriter = os.path.join(ROOT, "oracle", "_ref", "sawritermc")
    if not os.path.exists(EXE) or not os.path.exists(sawriter):
        pytest.skip("examples/map_fasta or oracle/_ref/sawritermc not built")
    want = write_inputs(str(tmp_path))
    if extend:
        want = sorted(bytes(np.load(os.path.join(ROOT, "tests", "golden", "m4_noisy_extend_golden.npz"))["lines"]).decode().splitlines())
    fa = os.path.join(str(tmp_path), "genome.fasta")
    subprocess.run([sawriter, fa + ".sa", fa, "-blt", "8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = subprocess.run([EXE, os.path.join(str(tmp_path), "reads.fasta"), fa, fa + ".sa", "1"] + (["-extend"] if extend else []),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = sorted(p.stdout.decode().splitlines())
    assert len(got) == len(want) == 80
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert not bad, bad[:3]


@pytest.mark.gpu
def test_two_strided_processes_print_the_one_process_lines(tmp_path):
    """The read file shared the reference's way (-start / -stride, Blasr.cpp:4057-4058): rank r of two runs `-start r
    -stride 2` on its own context, and the two outputs concatenated in rank order hold exactly the lines of the
    one-process run -- the only "collective" of the path besides the run counters (file I/O, not RCCL)."""
    sawriter = os.path.join(ROOT, "oracle", "_ref", "sawritermc")
    if not os.path.exists(EXE) or not os.path.exists(sawriter):
        pytest.skip("examples/map_fasta or oracle/_ref/sawritermc not built")
    want = write_inputs(str(tmp_path))
    fa = os.path.join(str(tmp_path), "genome.fasta")
    subprocess.run([sawriter, fa + ".sa", fa, "-blt", "8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    base = [EXE, os.path.join(str(tmp_path), "reads.fasta"), fa, fa + ".sa", "1"]
    procs = [subprocess.Popen(base + ["-start", str(r), "-stride", "2", "-device", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(2)]  # side by side, as two ranks would run (here on the one GPU of the test box)
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [e.decode()[-500:] for _, e in outs]
    parts = [o.decode().splitlines() for o, _ in outs]
    assert all(len(x) > 0 for x in parts) and not (set(parts[0]) & set(parts[1]))
    reads_of = [set(line.split()[0].split("/")[0] for line in x) for x in parts]  # read names: noisyK/...
    assert not (reads_of[0] & reads_of[1])
    got = sorted(parts[0] + parts[1])
    assert got == want
