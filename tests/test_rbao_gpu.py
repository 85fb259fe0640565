"""-rbao (params.refineBetweenAnchorsOnly; SURVEY next-3): AlignIntervals aligns BETWEEN the chained anchors of each
candidate interval instead of running SDPAlign over the whole window (alignment/Blasr.cpp:1354-1712: AlignSubstring
:1050-1122 per gap -- AffineKBandAlign, or SDPAlign + pinned ends + RefineAlignment -- AppendSubseqAlignment, merge,
statistics).  The library: the chained anchors out of blasr_hip_find_intervals_chains, the branch as
blasr_hip_align_between_anchors_* inside blasr_hip_map_reads (refine_between_anchors_only = 1).  What must come out is
what the REFERENCE BINARY prints under -rbao for the noisy fixture (tests/golden/make_m4_noisy_rbao_golden.py: 75 of
its 80 lines differ from the default run's), byte for byte -- from the step-by-step route, from the one-call mapper,
and from the patched reference binary mapping blocks of reads (-useGPU -gpuBatch -rbao)."""
import os
import subprocess

import numpy as np
import pytest

import blasr_amd
from blasr_amd import hip

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "m4_noisy_golden.npz")
RBAO = os.path.join(ROOT, "tests", "golden", "m4_noisy_rbao_golden.npz")


def fixture():
    g = np.load(GOLDEN)
    contigs = [g["contig0"], g["contig1"]]
    genome = np.concatenate([contigs[0], [ord("N")], contigs[1], [ord("N")]]).astype(np.uint8)
    starts = np.array([0, len(contigs[0]) + 1, len(genome)], dtype=np.uint32)
    read_lens, pool = g["read_lens"], np.ascontiguousarray(g["reads"])
    reads = np.zeros(len(read_lens), dtype=blasr_amd.READ_DTYPE)
    off = 0
    for i in range(len(read_lens)):
        reads[i] = (off, int(read_lens[i]), 0, int(read_lens[i]), 0)
        off += int(read_lens[i])
    want_iv = np.zeros(len(g["intervals"]), dtype=blasr_amd.INTERVAL_DTYPE)
    for k, (s, e, qs, qe, strand) in enumerate(g["intervals"]):
        want_iv[k] = (int(s), int(e), int(qs), int(qe), int(g["owner"][k]), int(strand))
    return genome, starts, reads, pool, read_lens, want_iv


@pytest.mark.parametrize("extend", [0, 1])
def test_m4_lines_of_noisy_reads_under_rbao(extend):
    genome, starts, reads, pool, read_lens, want_iv = fixture()
    want = bytes(np.load(RBAO)["lines_extend" if extend else "lines"]).decode().splitlines()
    plain = bytes(np.load(GOLDEN)["lines"]).decode().splitlines()
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome, starts)
    ctx.build_suffix_array(8, download=False)
    ivs, iv_off, _, chains, chain_off = blasr_amd.candidate_intervals(ctx, genome, starts, ctx.count_tuples(8), pool, reads,
                                                                     chains=True)
    assert np.array_equal(ivs, want_iv)  # the intervals are the reference's; their chains are what -rbao walks
    assert len(chain_off) == len(ivs) + 1 and chain_off[-1] == len(chains) and np.all(np.diff(chain_off) >= 1)
    for k in (0, len(ivs) // 2, len(ivs) - 1):  # an interval spans its chain (FindMaxInterval.h:826-831)
        c = chains[int(chain_off[k]): int(chain_off[k + 1])]
        assert int(c[0][0]) == int(ivs[k]["start"]) and int(c[-1][0] + c[-1][2]) == int(ivs[k]["end"])
        assert int(c[0][1]) == int(ivs[k]["q_start"]) and int(c[-1][1] + c[-1][2]) == int(ivs[k]["q_end"])
    params = ctx.map_params(refine_between_anchors_only=1, extend_alignments=extend)
    got = []
    for r, hits in enumerate(blasr_amd.map_reads(ctx, len(genome), starts, pool, reads, ivs, bestn=1, params=params,
                                                 chains=(chains, chain_off))):
        name = "read%d/%d/0_%d/0_%d" % (r, r, int(read_lens[r]), int(read_lens[r]))
        got += [blasr_amd.m4_record(h, name, "contig%d" % h["contig"]).rstrip("\n") for h in hits]
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert len(got) == len(want) and not bad, (len(got), len(want), len(bad), bad[:3])
    assert sum(a != b for a, b in zip(got, plain)) > 40  # and the switch did change the alignments
    # the switch without the chains is refused, not ignored
    with pytest.raises(hip.BlasrHipError):
        ctx.map_reads(reads, pool, ivs, params=ctx.map_params(refine_between_anchors_only=1), fetch=False)
    ctx.close()


@pytest.mark.parametrize("overlap", [0, 1])
def test_the_one_call_mapper_under_rbao(overlap):
    """blasr_hip_map_from_reads_device with map.refine_between_anchors_only: the chains travel from the interval finder to
    the path inside the library; the hits are those of the step-by-step route."""
    from blasr_testlib import DeviceMemory
    genome, starts, reads, pool, read_lens, want_iv = fixture()
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome, starts)
    ctx.build_suffix_array(8, download=False)
    table = ctx.count_tuples(8)
    ivs, _, _, chains, chain_off = blasr_amd.candidate_intervals(ctx, genome, starts, table, pool, reads, chains=True)
    want_hits, want_off, _, _ = ctx.map_reads(reads, pool, ivs, params=ctx.map_params(refine_between_anchors_only=1),
                                              fetch=False, chains=(chains, chain_off))
    mem = DeviceMemory()
    d_reads, d_pool = mem.upload(reads.view(np.uint8)), mem.upload(pool)
    hits = np.zeros(len(ivs) + 8, dtype=hip.HIT_DTYPE)
    hit_off = np.zeros(len(reads) + 1, dtype=np.int64)
    p = ctx.pipeline_params(reads_per_block=9, overlap=overlap)
    p.map.refine_between_anchors_only = 1
    n, got_iv = ctx.map_from_reads_device(genome, starts, table, reads, d_reads, d_pool, p, hits, hit_off,
                                          intervals_capacity=len(ivs) + 8)
    assert np.array_equal(got_iv, ivs)
    assert n == len(want_hits) and np.array_equal(hit_off, want_off)
    for f in ("read", "interval", "t_strand", "t_aligned_seq_pos", "q_pos", "t_pos", "score", "n_match", "n_ops"):
        assert np.array_equal(hits[:n][f], want_hits[f]), f
    mem.free()
    ctx.close()


def test_patched_reference_binary_mapping_blocks_of_reads_under_rbao(tmp_path):
    from test_dropin_gpu import EXE, run, write_inputs
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/blasrmc_gpu not built (needs /root/reference at build time)")
    write_inputs(str(tmp_path))
    want = sorted(bytes(np.load(RBAO)["lines"]).decode().splitlines())
    p, got = run(str(tmp_path), ["-useGPU", "-gpuBatch", "11", "-nproc", "2", "-rbao"])
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    assert len(got) == len(want) == 80
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert not bad, bad[:3]


def test_cpp_example_maps_under_rbao(tmp_path):
    """examples/map_fasta -rbao: the C++ whole mapper over the C ABI alone (FASTA + .sa in, -m 4 lines out) with the
    switch set in its pipeline parameters."""
    from test_dropin_gpu import write_inputs
    exe = os.path.join(ROOT, "examples", "map_fasta")
    sawriter = os.path.join(ROOT, "oracle", "_ref", "sawritermc")
    if not os.path.exists(exe) or not os.path.exists(sawriter):
        pytest.skip("examples/map_fasta or oracle/_ref/sawritermc not built")
    write_inputs(str(tmp_path))
    want = sorted(bytes(np.load(RBAO)["lines"]).decode().splitlines())
    fa = os.path.join(str(tmp_path), "genome.fasta")
    subprocess.run([sawriter, fa + ".sa", fa, "-blt", "8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = subprocess.run([exe, os.path.join(str(tmp_path), "reads.fasta"), fa, fa + ".sa", "1", "-rbao"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = sorted(p.stdout.decode().splitlines())
    assert len(got) == len(want) == 80
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert not bad, bad[:3]


def test_the_branch_at_function_level_against_the_references_own_alignintervals():
    """blasr_hip_align_between_anchors_host for the candidates of the noisy fixture, one by one against the REFERENCE's
    own AlignIntervals with refineBetweenAnchorsOnly set and the interval's chain handed to it
    (oracle/ref_mapper_harness.cpp: ref_set_interval_matches + ref_align_interval_ex): qPos / tPos, every block, and the
    ComputeAlignmentStats fields of :1709."""
    import blasr_testlib as T
    ref = T.load_refmapper()
    if ref is None:
        pytest.skip("oracle/_ref/libblasr_refmapper.so not built (needs /root/reference at build time)")
    genome, starts, reads, pool, read_lens, want_iv = fixture()
    ctx = blasr_amd.Context(0)
    ctx.upload_genome(genome, starts)
    ctx.build_suffix_array(8, download=False)
    ivs, iv_off, _, chains, chain_off = blasr_amd.candidate_intervals(ctx, genome, starts, ctx.count_tuples(8), pool, reads,
                                                                     chains=True)
    seq, q_off, q_len, t_off, t_len, win = ctx.windows(ivs, reads, pool)
    res, blocks = ctx.align_between_anchors(seq, q_off, q_len, t_off, t_len, win, chains, chain_off)
    checked = with_gaps = 0
    for k in range(len(ivs)):
        iv = ivs[k]
        r = int(iv["read"])
        read = pool[int(reads[r]["off"]): int(reads[r]["off"]) + int(reads[r]["length"])]
        chain = chains[int(chain_off[k]): int(chain_off[k + 1])]
        w, tseq, sdp = ref.align_interval(genome, read, (0, len(read)),
                                          (int(iv["start"]), int(iv["end"]), int(iv["q_start"]), int(iv["q_end"]), int(iv["strand"])),
                                          contigs=starts, rbao_chain=chain)
        mine = res[k]
        b = blocks[int(mine["blk_off"]): int(mine["blk_off"]) + int(mine["n_blocks"])]
        assert (int(mine["q_pos"]), int(mine["t_pos"])) == (sdp["qPos"], sdp["tPos"]), k
        assert np.array_equal(b, sdp["blocks"]), k
        assert (int(mine["stats_score"]), int(mine["n_match"]), int(mine["n_mismatch"]), int(mine["n_ins"]), int(mine["n_del"])) == \
               (sdp["score"], sdp["nMatch"], sdp["nMismatch"], sdp["nIns"], sdp["nDel"]), k
        checked += 1
        with_gaps += int(len(b) > len(chain))
    assert checked == len(ivs) == 125 and with_gaps > 60  # (most candidates have gaps that were aligned)
    ctx.close()
