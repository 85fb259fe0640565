"""next-4: the suffix array and its lookup table built on the device (blasr_hip_build_suffix_array:
prefix doubling) and written as a .sa file (blasr_hip_sa_write), against the REFERENCE's sawriter
(oracle/_ref/sawritermc: Larsson-Sadakane + BuildLookupTable + Write) on the same genomes: the index arrays, the
lookup tables with their quirks, and the files byte for byte; and the anchors that come out of the device-built index."""
import os
import subprocess

import numpy as np
import pytest

import blasr_amd
from blasr_amd import hip, synth
import blasr_testlib as T

pytestmark = pytest.mark.gpu
have_sawriter = os.path.exists(T.SAWRITER)


def genomes(rng):
    # random, with planted repeats (long common prefixes -> several doubling rounds), a run of N, a tandem repeat,
    # several contigs, and short ones around the first key's length
    g1 = bytearray(bytes(synth.random_genome(60000, 5)))
    g1[1000:4000] = g1[30000:33000]          # an exact 3 kb repeat
    g1[7000:7400] = b"N" * 400
    g1[9000:10000] = (b"ACG" * 400)[:1000]   # a tandem repeat: suffixes that differ only near its end
    yield b">a\n" + bytes(g1[:25000]) + b"\n>b\n" + bytes(g1[25000:]) + b"\n"
    yield b">poly\n" + b"T" * 300 + b"A" * 200 + b"TTTTTTTTTTTTTT" + b"\n"
    yield b">tiny\nACGTTGCA\n"
    yield b">x\n" + bytes(synth.random_genome(23, 9)) + b"\n>y\nAC\n>z\n" + bytes(synth.random_genome(45, 10)) + b"\n"


@pytest.mark.skipif(not have_sawriter, reason="oracle/_ref/sawritermc not built")
def test_suffix_array_and_lookup_table_equal_sawriters(tmp_path):
    rng = np.random.default_rng(1)
    ctx = blasr_amd.Context(0)
    for gi, fasta in enumerate(genomes(rng)):
        for blt in (4, 8, 13) if gi == 0 else (3, 6):
            fa = os.path.join(str(tmp_path), "g%d.fasta" % gi)
            with open(fa, "wb") as f:
                f.write(fasta)
            ref_path = fa + ".%d.sa" % blt
            subprocess.run([T.SAWRITER, ref_path, fa, "-blt", str(blt)], check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=600)
            want = hip.read_suffix_array(ref_path)
            genome, starts, _ = hip.read_genome(fa)
            ctx.upload_genome(genome, starts)
            got = ctx.build_suffix_array(blt)
            assert np.array_equal(got["index"], want["index"]), (gi, blt)
            assert np.array_equal(got["start_pos_table"], want["start_pos_table"]), (gi, blt)
            assert np.array_equal(got["end_pos_table"], want["end_pos_table"]), (gi, blt)
            mine = fa + ".%d.mine.sa" % blt
            hip.write_suffix_array(mine, got)
            assert open(mine, "rb").read() == open(ref_path, "rb").read(), (gi, blt)
            ctx.write_resident_suffix_array(mine + "2")  # the resident copy, streamed from the device to the file
            assert open(mine + "2", "rb").read() == open(ref_path, "rb").read(), (gi, blt)
    ctx.close()


def test_anchors_from_the_device_built_index_equal_the_golden_lists():
    """The golden anchor lists were made by the reference on sawriter's index of this genome (-blt 6): the index built
    here must be that index, and the resident copy must serve the anchor stage without an upload."""
    import test_anchor_gpu as A
    z, sa = A.golden_case()
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(z["genome"])
    got = ctx.build_suffix_array(6)
    assert np.array_equal(got["index"], sa["index"])
    assert np.array_equal(got["start_pos_table"], sa["start_pos_table"]) and np.array_equal(got["end_pos_table"], sa["end_pos_table"])
    case = z["cases"][0]
    ps, min_prefix, sort = tuple(int(v) for v in case[:7]), int(case[7]), int(case[10])
    anchors, off = ctx.anchor_reads(z["reads"], z["pool"], A.params_of(ctx, ps, min_prefix, sort))
    r, strand, n = int(case[8]), int(case[9]), int(z["counts"][0])
    assert np.array_equal(A.as_triples(anchors[int(off[2 * r + strand]): int(off[2 * r + strand + 1])]), z["anchors"][:n])
    with pytest.raises(hip.BlasrHipError):
        ctx.build_suffix_array(15)
    for k in (3, 8):  # the count table of the same genome, on the device and on the host
        assert np.array_equal(ctx.count_tuples(k), hip.count_tuples(z["genome"], k))
    ctx.close()
