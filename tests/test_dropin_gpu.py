"""The drop-in, compiled for real: oracle/_ref/blasrmc_gpu is the REFERENCE's own alignment/Blasr.cpp with
INTEGRATION.md's patch applied (oracle/make_dropin.py: a scratch copy at build time; oracle/dropin/blasr_hip_dropin.h
holds the new code) and linked against libblasr_hip.so -- a C++ caller of the C ABI inside the reference's MapRead,
one context per mapping thread.  With -useGPU -nproc 2 -- MapReadToGenome (both strands, on the device), SDPAlign +
ComputeAlignmentStats, the two ExtendAlignment calls of -extend and the refine loop through the library, everything else
the reference's own code -- it must
print the -m 4 lines the unpatched binary printed for the noisy reads with competing candidates
(tests/golden/m4_noisy_golden.npz)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "blasrmc_gpu")
GOLDEN = os.path.join(ROOT, "tests", "golden", "m4_noisy_golden.npz")


def write_inputs(tmp):
    g = np.load(GOLDEN)
    with open(os.path.join(tmp, "genome.fasta"), "w") as f:
        for i in (0, 1):
            s = bytes(g["contig%d" % i]).decode()
            f.write(">contig%d\n" % i)
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + "\n")
    with open(os.path.join(tmp, "reads.fasta"), "w") as f:
        off = 0
        for r, n in enumerate(g["read_lens"]):
            f.write(">read%d/%d/0_%d\n%s\n" % (r, r, int(n), bytes(g["reads"][off: off + int(n)]).decode()))
            off += int(n)
    return sorted(bytes(g["lines"]).decode().splitlines())


def run(tmp, extra):
    out = os.path.join(tmp, "out.m4")
    p = subprocess.run([EXE, os.path.join(tmp, "reads.fasta"), os.path.join(tmp, "genome.fasta"), "-m", "4", "-bestn", "1",
                        "-out", out] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    return p, (sorted(open(out).read().splitlines()) if os.path.exists(out) else [])


@pytest.mark.gpu
def test_patched_reference_binary_on_the_gpu(tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/blasrmc_gpu not built (needs /root/reference at build time)")
    want = write_inputs(str(tmp_path))
    p, got = run(str(tmp_path), ["-useGPU", "-nproc", "2"])
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    assert len(got) == len(want) == 80
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert not bad, bad[:3]


@pytest.mark.gpu
def test_patched_reference_binary_under_extend(tmp_path):
    """-extend -useGPU: the two ExtendAlignment calls of AlignIntervals (Blasr.cpp:1826, :1885) go through
    blasr_hip_extend_host as well, the glue around them stays the reference's own; the lines are the unpatched binary's
    under -extend (tests/golden/m4_noisy_extend_golden.npz)."""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/blasrmc_gpu not built (needs /root/reference at build time)")
    write_inputs(str(tmp_path))
    want = sorted(bytes(np.load(os.path.join(ROOT, "tests", "golden", "m4_noisy_extend_golden.npz"))["lines"]).decode().splitlines())
    p, got = run(str(tmp_path), ["-useGPU", "-nproc", "2", "-extend"])
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    assert len(got) == len(want) == 80
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert not bad, bad[:3]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["-gpuBatch", "16", "-nproc", "2"], ["-gpuBatch", "4096", "-nproc", "1"],
                                   ["-gpuBatch", "7", "-nproc", "3", "-extend"]])
def test_patched_reference_binary_mapping_blocks_of_reads(tmp_path, extra):
    """-useGPU -gpuBatch B: MapReads pulls B reads at a time from the reader and ONE library call maps the block
    (anchoring, intervals, AlignIntervals, filters, RefineAlignments); StoreMapQVs and the printers stay the
    reference's own, fed with alignment candidates rebuilt from the block's hits.  The lines are the unpatched binary's
    -- with blocks smaller than the file (several blocks per thread, threads sharing the one context), with one block
    for everything, and under -extend."""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/blasrmc_gpu not built (needs /root/reference at build time)")
    want = write_inputs(str(tmp_path))
    if "-extend" in extra:
        want = sorted(bytes(np.load(os.path.join(ROOT, "tests", "golden", "m4_noisy_extend_golden.npz"))["lines"]).decode().splitlines())
    p, got = run(str(tmp_path), ["-useGPU"] + extra)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    assert len(got) == len(want) == 80
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert not bad, bad[:3]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["-nproc", "2"], ["-gpuBatch", "16", "-nproc", "2"]])
def test_patched_reference_binary_with_the_kband_refiner(tmp_path, extra):
    """-useGPU -noUseGuidedAlign: RefineAlignment's k-band branch (Blasr.cpp:917-1046) through blasr_hip_kband_host
    (per-read mode) or inside the mapping call (-gpuBatch, use_guided_align = 0).  The unpatched binary computes that
    branch's score with an uninitialised matrix (Blasr.cpp:689-690): its score column is stack garbage and a third of
    the reads fail its score cut-off on it (tests/golden/make_m4_noisy_nonguided_golden.py).  So: all 80 reads print
    here, with a score that is the alignment's; the two modes and two runs print the same lines; and where the unpatched
    binary printed the same candidate its identity and its four coordinates are ours."""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/blasrmc_gpu not built (needs /root/reference at build time)")
    write_inputs(str(tmp_path))
    ref = bytes(np.load(os.path.join(ROOT, "tests", "golden", "m4_noisy_nonguided_golden.npz"))["lines"]).decode().splitlines()
    p, got = run(str(tmp_path), ["-useGPU", "-noUseGuidedAlign"] + extra)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    assert len(got) == 80
    p2, again = run(str(tmp_path), ["-useGPU", "-noUseGuidedAlign", "-nproc", "1"])
    assert p2.returncode == 0 and again == got
    ours = {line.split()[0]: line.split() for line in got}
    same = 0
    for line in ref:
        f = line.split()
        g = ours[f[0]]
        assert -40000 < int(g[2]) < 0  # (a score of this alignment, not of the stack)
        if (f[1], f[4], f[8]) == (g[1], g[4], g[8]) and abs(int(f[9]) - int(g[9])) < 2000:  # the same candidate
            assert f[3:12] == g[3:12], (line, " ".join(g))
            same += 1
    assert same >= len(ref) // 2  # (36 of 54; the others are candidates the garbage score preferred)


def test_patched_reference_binary_without_the_flag_is_the_reference(tmp_path):
    """Without -useGPU the patched binary takes the reference's own code (the patch only adds branches), and with it
    but without a device it fails loudly instead of falling back."""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/blasrmc_gpu not built (needs /root/reference at build time)")
    want = write_inputs(str(tmp_path))
    p, got = run(str(tmp_path), ["-nproc", "2"])
    assert p.returncode == 0 and got == want
    import ctypes as C
    from blasr_amd import hip
    lib, h = hip.load_library(), C.c_void_p()
    have_device = lib.blasr_hip_create(0, C.byref(h)) == hip.OK  # (the library's own answer, not torch's)
    if have_device:
        lib.blasr_hip_destroy(h)
    else:
        p, got = run(str(tmp_path), ["-useGPU"])
        assert p.returncode != 0 and b"no gfx950 device" in p.stdout
