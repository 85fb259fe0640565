"""Randomised parity fuzz of the DEVICE anchoring against the oracle / the reference (oracle/_ref):
small genomes with exact repeats, tandem repeats, N runs and several contigs; reads with and without errors, N, partial
subreads; random anchor and interval parameters.  Per read: the oracle's MapReadToGenome (+ sort) against the
reference's, and the library's blasr_hip_find_intervals (intervals and p-values, random stale elements) against the
reference's RemoveOverlappingAnchors + FindMaxIncreasingInterval x 2.
    python tools/fuzz_anchoring_cpu.py SEED ITERATIONS"""
import sys, tempfile, numpy as np, ctypes as C, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import blasr_testlib as T
from blasr_amd import hip, synth
import blasr_amd
oracle = T.load_oracle()
ctx = blasr_amd.Context(0)
ref = C.CDLL(T.REFMAPPER_SO); ref.ref_ct_open.restype = C.c_void_p; ref.ref_sa_open.restype = C.c_void_p
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
tmp = tempfile.mkdtemp()
bad = 0; total = 0; t0 = time.time()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    glen = int(rng.integers(3000, 60000))
    g = bytearray(bytes(synth.random_genome(glen, int(rng.integers(1, 1 << 30)))))
    # repeats: exact copies, tandem, N runs
    for _ in range(int(rng.integers(0, 5))):
        L = int(rng.integers(50, 2000)); a = int(rng.integers(0, glen - L)); b = int(rng.integers(0, glen - L))
        g[b:b + L] = g[a:a + L]
    if rng.integers(0, 2):
        u = bytes(synth.random_genome(int(rng.integers(2, 9)), 7)); a = int(rng.integers(0, glen - 800)); g[a:a + 600] = (u * 400)[:600]
    if rng.integers(0, 2):
        a = int(rng.integers(0, glen - 100)); g[a:a + int(rng.integers(1, 60))] = b"N" * 59
        g = g[:glen]
    ncont = int(rng.integers(1, 4)); cuts = sorted(set([0, glen] + [int(x) for x in rng.integers(100, glen - 100, ncont - 1)]))
    fasta = b"".join(b">c%d\n" % i + bytes(g[cuts[i]:cuts[i + 1]]) + b"\n" for i in range(len(cuts) - 1))
    blt = int(rng.choice([5, 6, 8]))
    genome, starts, sa, sa_path = T.write_suffix_array(tmp, fasta, blt)
    ctx.upload_genome(genome, starts)
    mine = ctx.build_suffix_array(blt)
    if not (np.array_equal(mine["index"], sa["index"]) and np.array_equal(mine["start_pos_table"], sa["start_pos_table"])
            and np.array_equal(mine["end_pos_table"], sa["end_pos_table"])):
        bad += 1; print("INDEX DIFF", seed, it, blt, glen)
    h = ref.ref_sa_open(sa_path.encode())
    hct = ref.ref_ct_open(genome.ctypes.data_as(C.c_void_p), len(genome), 8)
    ct = hip.count_tuples(genome, 8)
    nreads = int(rng.integers(2, 7))
    ps = (int(rng.integers(blt, 16)), int(rng.integers(0, 3)), 1, int(rng.choice([1000, 5, 50])), int(rng.choice([0, 0, 3])), int(rng.choice([0, 0, 18])), int(rng.integers(0, 2)))
    kw = dict(n_candidates=int(rng.integers(1, 11)), min_interval=int(rng.choice([0, 20, 100, 300])), remove_contained=int(rng.integers(0, 2)), indel_rate=float(rng.choice([0.05, 0.3, 0.6])))
    for r in range(nreads):
        L = int(rng.integers(40, 3000)); pos = int(rng.integers(0, max(1, len(genome) - L - 1)))
        tmpl = genome[pos:pos + L]
        if rng.integers(0, 2): tmpl = synth.revcomp(tmpl)
        e = float(rng.choice([0.0, 0.05, 0.16]))
        rd = np.ascontiguousarray(synth.mutate(tmpl, rng, e * 0.7, e * 0.25, e * 0.05)) if e else np.ascontiguousarray(tmpl).copy()
        rd[rd == ord("N")] = ord("A") if rng.integers(0, 2) else ord("N")
        if len(rd) < 30: continue
        s0 = int(rng.integers(0, len(rd) // 4)) if rng.integers(0, 3) == 0 else 0
        s1 = len(rd) - (int(rng.integers(0, len(rd) // 4)) if rng.integers(0, 3) == 0 else 0)
        lists = []
        okc = True
        for strand in (0, 1):
            a = T.oracle_anchors(oracle, genome, sa, rd, s0, s1, strand, blt, ps, True)
            if a is None: okc = False; break
            w = T.ref_anchors(ref, h, genome, rd, s0, s1, strand, blt, ps, True)
            if not np.array_equal(a, w): bad += 1; print("ANCHOR DIFF", seed, it, r, strand, ps)
            lists.append(a)
        if not okc: continue
        reads = np.zeros(1, hip.READ_DTYPE); reads[0] = (0, len(rd), s0, s1, 0)
        dp = ctx.anchor_params(min_match_length=ps[0], expand=ps[1], use_lookup_table=ps[2], max_anchors_per_position=ps[3],
                               advance_exact_matches=ps[4], max_lcp_length=ps[5], stop_mapping_once_unique=ps[6],
                               min_prefix_match_length=blt)
        dev, doff = ctx.anchor_reads(reads, rd, dp)
        for strand in (0, 1):
            m = dev[int(doff[strand]): int(doff[strand + 1])]
            got3 = np.stack([m["t"], m["q"], m["l"]], axis=1) if len(m) else np.zeros((0, 3), np.uint32)
            if not np.array_equal(got3, lists[strand]): bad += 1; print("DEVICE ANCHOR DIFF", seed, it, r, strand, ps)
        flat = np.concatenate(lists) if (len(lists[0]) + len(lists[1])) else np.zeros((0, 3), np.uint32)
        an = np.zeros(len(flat), hip.ANCHOR_DTYPE)
        if len(flat): an["t"], an["q"], an["l"] = flat[:, 0], flat[:, 1], flat[:, 2]
        off = np.array([0, len(lists[0]), len(flat)], np.int64)
        stale = np.zeros(2, hip.ANCHOR_DTYPE); stale["t"] = rng.integers(0, len(genome), 2); stale["q"] = rng.integers(0, 3000, 2); stale["l"] = rng.integers(5, 40, 2)
        p = hip.interval_params(n_candidates=kw["n_candidates"], min_interval=kw["min_interval"], remove_contained_intervals=kw["remove_contained"], indel_rate=kw["indel_rate"], threads=1)
        ivs, ioff, pv = hip.find_intervals(genome, starts, ct, reads, an, off, p, stale)
        st3 = np.stack([stale["t"], stale["q"], stale["l"]], axis=1)
        want, wpv = T.ref_find_intervals(ref, hct, genome, starts, len(rd), s0, s1, lists[0], lists[1], st3, **kw)
        got = np.stack([ivs["start"], ivs["end"], ivs["q_start"], ivs["q_end"], ivs["strand"].astype(np.uint32)], axis=1) if len(ivs) else np.zeros((0, 5), np.uint32)
        total += 1
        if not (np.array_equal(got, want) and np.array_equal(pv, wpv)):
            bad += 1; print("INTERVAL DIFF", seed, it, r, kw, ps, got.tolist(), want.tolist())
    ref.ref_sa_close(C.c_void_p(h)); ref.ref_ct_close(C.c_void_p(hct))
print("seed", seed, "reads", total, "bad", bad, "seconds", round(time.time() - t0, 1))
