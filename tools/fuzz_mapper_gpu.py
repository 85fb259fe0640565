#!/usr/bin/env python3
"""Whole-mapper fuzz against the REFERENCE BINARY, on the GPU box: random multi-contig genomes with planted repeats,
random reads (both strands, several error mixes, some hanging over a contig end), mapped (a) by oracle/_ref/blasrmc
-m 4 -bestn 1, with and without -extend, and (b) by the library from the reads alone (index built on the device,
anchoring, interval finding, windows -> SDP -> [extension] -> filters -> refinement -> filters, epilogue).  The -m 4
lines are compared as sorted sets, byte for byte.  One JSON line.

    python tools/fuzz_mapper_gpu.py [--iters 6] [--reads 120] [--seed 1] [--nproc 16]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--reads", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--nproc", type=int, default=16)
    ap.add_argument("--dump", type=int, default=-1, metavar="IT",
                    help="write the FASTA files of iteration IT to ./fuzz_mapper_dump/ and stop (no GPU needed)")
    ap.add_argument("--contig-len", type=int, nargs=2, default=[60000, 160000], metavar=("LO", "HI"))
    ap.add_argument("--read-lens", type=int, nargs="+", default=[300, 700, 1500, 3000, 6000])
    ap.add_argument("--dropin", action="store_true",
                    help="also run the reference binary patched to call the library (oracle/_ref/blasrmc_gpu -useGPU -nproc 2: "
                         "anchoring, SDPAlign and the refine loop through the C ABI, the rest its own code) and compare its lines")
    ap.add_argument("--sam", action="store_true", help="compare SAM records (-sam: CIGAR, flags, tags) instead of -m 4 lines")
    ap.add_argument("--variants", action="store_true", help="cycle through non-default switches (see the table in main)")
    ap.add_argument("--fastq", action="store_true",
                    help="hand the reads over as a FASTQ file: the reference reads it with its FASTQReader (every second "
                         "record of a four-line file, FASTQReader.h:27-66), the library with blasr_hip_fastq_read; with --sam "
                         "the records carry the quality strings")
    ap.add_argument("--sam-reads", action="store_true",
                    help="hand the reads over as a SAM file (unaligned records with qualities): the reference reads it through "
                         "ReaderAgglomerate's SAM branch, the library with blasr_hip_sam_reads_read (-m 4 lines only)")
    ap.add_argument("--only-it", type=int, default=-1, metavar="IT", help="map iteration IT alone (the others only draw their data)")
    ap.add_argument("--repeat", type=int, default=1, help="with --only-it: the library's leg that many times (a fresh context each), "
                                                           "every run compared; for differences that come and go")
    ap.add_argument("--rbao", action="store_true",
                    help="-rbao on both sides: AlignIntervals aligns between the chained anchors of each interval "
                         "(Blasr.cpp:1354-1712; the library: blasr_hip_find_intervals_chains + refine_between_anchors_only)")
    args = ap.parse_args()
    import blasr_amd
    from blasr_amd import hip, synth
    blasr = os.path.join(ROOT, "oracle", "_ref", "blasrmc")
    rng = np.random.default_rng(args.seed)
    totals = dict(iters=0, reads=0, lines_plain=0, lines_extend=0, bad_plain=0, bad_extend=0, changed_by_extend=0)
    # non-default switches of the reference and the library parameters they become (MappingParameters.h:367-545,
    # Blasr.cpp:3999-4135); --variants cycles through them, one per iteration
    variants = [
        dict(cli=[], name="default"),
        dict(cli=["-bestn", "3"], bestn=3, name="bestn3"),
        dict(cli=["-sdpTupleSize", "8"], sdp=dict(k=8), name="sdpTupleSize8"),
        dict(cli=["-bandSize", "30"], map=dict(band=30), name="bandSize30"),
        dict(cli=["-maxScore", "-1500"], map=dict(max_score=-1500), name="maxScore-1500"),
        dict(cli=["-minPctIdentity", "84"], map=dict(min_pct_identity=84.0), name="minPctIdentity84"),
        dict(cli=["-nCandidates", "2"], map=dict(n_candidates=2), interval=dict(n_candidates=2), name="nCandidates2"),
        dict(cli=["-minMatch", "12"], anchor=dict(min_match_length=12), name="minMatch12"),
        dict(cli=["-extendBandSize", "5", "-maxExtendDropoff", "3"], map=dict(extend_band_size=5, max_extend_dropoff=3),
             name="extendBand5Drop3"),
        dict(cli=["-indelRate", "0.2"], sdp=dict(indel_rate=float(np.float32(0.2) * np.float32(3))), interval=dict(indel_rate=0.2),
             name="indelRate0.2"),
        dict(cli=["-maxAnchorsPerPosition", "3"], anchor=dict(max_anchors_per_position=3), name="maxAnchorsPerPosition3"),
        dict(cli=["-sdpIns", "3", "-sdpDel", "7"], sdp=dict(sdp_ins=3, sdp_del=7), name="sdpIns3Del7"),
        dict(cli=["-maxGap", "40"], map=dict(max_gap=40), name="maxGap40"),
        dict(cli=["-insertion", "4", "-deletion", "7"], score=dict(ins=4, dele=7), name="insertion4deletion7"),
        dict(cli=["-affineOpen", "12", "-affineExtend", "2"], score=dict(affine_open=12, affine_extend=2), name="affine12+2"),
        dict(cli=["-minAlignLength", "1000"], min_align_length=1000, name="minAlignLength1000"),
        dict(cli=["-minMapQV", "30"], min_map_qv=30, name="minMapQV30"),
        dict(cli=["-minInterval", "400"], interval=dict(min_interval=400), name="minInterval400"),
        dict(cli=["-removeContained"], interval=dict(remove_contained_intervals=1), name="removeContained"),
        dict(cli=["-forwardOnly"], anchor=dict(forward_only=1), name="forwardOnly"),
        # not here: -advanceExactMatches.  With it the anchors of a list no longer overlap, RemoveOverlappingAnchors
        # removes nothing, and StoreLargestIntervals' read one past the end of the list (FindMaxInterval.h) lands in
        # what an EARLIER read left in the buffer: the reference binary's own lines then depend on the order of the
        # reads (200 reads forwards / backwards under -nproc 1: 11 lines differ; none do by default).  The anchoring
        # function under that switch is pinned at function level (tests/test_anchor_*.py, tools/fuzz_anchoring_gpu.py).
        # not here either: -global (KBandAlign over the whole window instead of the guided refiner; the reference prints
        # nothing for these reads, the library's driver refuses the switch).
        dict(cli=["-maxLCPLength", "16"], anchor=dict(max_lcp_length=16), name="maxLCPLength16"),
        dict(cli=["-sdpMaxAnchorsPerPosition", "2"], sdp=dict(max_per_pos=2), name="sdpMaxAnchorsPerPosition2"),
        dict(cli=["-sdpPrefix", "20"], sdp=dict(prefix=20), name="sdpPrefix20"),
        dict(cli=["-clipping", "soft"], clipping=hip.CLIP_SOFT, name="clippingSoft"),
        dict(cli=["-clipping", "hard"], clipping=hip.CLIP_HARD, name="clippingHard"),
        dict(cli=["-clipping", "subread"], clipping=hip.CLIP_SUBREAD, name="clippingSubread"),
        dict(cli=["-bestn", "4", "-nCandidates", "4", "-minPctIdentity", "70"], bestn=4,
             map=dict(n_candidates=4, min_pct_identity=70.0), interval=dict(n_candidates=4), name="bestn4nCandidates4pct70"),
    ]
    by_variant = {}
    examples = []
    t0 = time.time()
    for it in range(args.iters):
        n_contigs = int(rng.integers(1, 4))
        contigs = [synth.random_genome(int(rng.integers(args.contig_len[0], args.contig_len[1])), int(rng.integers(1, 1 << 30)))
                   for _ in range(n_contigs)]
        for _ in range(int(rng.integers(0, 4))):  # planted repeats, some reverse-complemented
            a, b = int(rng.integers(0, n_contigs)), int(rng.integers(0, n_contigs))
            ln = int(rng.integers(2000, 7000))
            src = int(rng.integers(0, len(contigs[a]) - ln))
            d = float(rng.uniform(0.0, 0.05))
            piece = np.ascontiguousarray(synth.mutate(contigs[a][src:src + ln], rng, d / 3, d / 3, d / 3))
            if rng.integers(0, 2):
                piece = synth.revcomp(piece)
            at = int(rng.integers(0, len(contigs[b]) - len(piece)))
            contigs[b] = np.concatenate([contigs[b][:at], piece, contigs[b][at + len(piece):]])
        if it % 3 == 1:  # a run of N inside a contig (the FASTA reader keeps it; anchors and windows meet it)
            c = int(rng.integers(0, n_contigs))
            at = int(rng.integers(1000, len(contigs[c]) - 1000))
            contigs[c] = contigs[c].copy()
            contigs[c][at:at + int(rng.integers(1, 80))] = ord("N")
        reads = []
        for r in range(args.reads):
            ci = int(rng.integers(0, n_contigs))
            rl = int(rng.choice(args.read_lens))
            rl = min(rl, len(contigs[ci]) - 10)
            mode = int(rng.integers(0, 5))
            if mode == 0:      # starts at the very beginning of the contig
                pos = 0
            elif mode == 1:    # ends at the very end
                pos = len(contigs[ci]) - rl
            else:
                pos = int(rng.integers(0, len(contigs[ci]) - rl))
            s = contigs[ci][pos:pos + rl]
            if rng.integers(0, 2):
                s = synth.revcomp(s)
            ins, dele, sub = [(0.11, 0.04, 0.01), (0.02, 0.01, 0.01), (0.15, 0.08, 0.03), (0.0, 0.0, 0.0)][int(rng.integers(0, 4))]
            rd = np.ascontiguousarray(synth.mutate(s, rng, ins, dele, sub))
            if rng.integers(0, 4) == 0:  # an unrelated tail the extension must not swallow
                rd = np.concatenate([rd, np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(rng.integers(5, 200)))]])
            if rng.integers(0, 12) == 0:  # a few N in the read
                rd = rd.copy()
                rd[rng.integers(0, len(rd), int(rng.integers(1, 4)))] = ord("N")
            if len(rd) < 50:
                continue
            reads.append(rd)
        if args.dump >= 0 and it < args.dump:
            continue
        with tempfile.TemporaryDirectory() as tmp:
            if args.dump >= 0:
                tmp = os.path.join(os.getcwd(), "fuzz_mapper_dump")
                os.makedirs(tmp, exist_ok=True)
            with open(os.path.join(tmp, "genome.fasta"), "w") as f:
                for i, c in enumerate(contigs):
                    s = bytes(c).decode()
                    f.write(">contig%d\n" % i)
                    for k in range(0, len(s), 60):
                        f.write(s[k:k + 60] + "\n")
            reads_file = "reads.fastq" if args.fastq else ("reads.sam" if args.sam_reads else "reads.fasta")
            quals = None
            names = ["read%d/%d/0_%d" % (r, r, len(rd)) for r, rd in enumerate(reads)]
            with open(os.path.join(tmp, reads_file), "w") as f:
                if args.sam_reads:
                    f.write("@HD\tVN:1.4\n@RG\tID:fuzz\tPU:m\n")
                for r, rd in enumerate(reads):
                    if args.sam_reads:
                        q = bytes(rng.choice(list(b"!+5?IJ"), len(rd)).astype(np.uint8)).decode()
                        f.write("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\tRG:Z:fuzz\n" % (names[r], bytes(rd).decode(), q))
                    elif args.fastq:  # (no '@' among the qualities: this reader takes one for a title start; tests/test_formats_cpu.py)
                        q = bytes(rng.choice(list(b"!+5?IJ"), len(rd)).astype(np.uint8)).decode()
                        f.write("@%s\n%s\n+\n%s\n" % (names[r], bytes(rd).decode(), q))
                    else:
                        f.write(">%s\n%s\n" % (names[r], bytes(rd).decode()))
            if args.sam_reads:
                recs = hip.read_sam_reads(os.path.join(tmp, reads_file))
                assert [t.decode() for t, _, _ in recs] == names and all(np.array_equal(a, b) for (_, a, _), b in zip(recs, reads))
            if args.fastq:  # what the reference's reader makes of the file is what gets mapped
                try:
                    recs = hip.read_fastq(os.path.join(tmp, reads_file))
                except hip.BlasrHipError:
                    totals["fastq_outside_the_readers_domain"] = totals.get("fastq_outside_the_readers_domain", 0) + 1
                    continue  # (an '@' among the qualities left a record the reference reads undefined memory for)
                names = [t.decode() for t, _, _ in recs]
                reads = [sq for _, sq, _ in recs]
                quals = [q for _, _, q in recs]
            var = variants[it % len(variants)] if args.variants else variants[0]
            if args.dump >= 0:
                print("iteration %d: variant %s, switches %s, files in %s" % (it, var["name"], var["cli"], tmp))
                return
            if args.only_it >= 0 and it != args.only_it:
                continue
            read_group = [""]

            def run_reference(extra, nproc, exe=None):
                path = os.path.join(tmp, "out.m4")
                p = subprocess.run([exe or blasr, os.path.join(tmp, reads_file), os.path.join(tmp, "genome.fasta")] +
                                   (["-sam"] if args.sam else ["-m", "4"]) +
                                   ["-bestn", "1", "-nproc", str(nproc), "-out", path] + extra,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                if p.returncode != 0:
                    return None  # the reference asserted / crashed on this input
                lines = open(path).read().splitlines()
                for x in lines:
                    if x.startswith("@RG"):
                        read_group[0] = x.split("ID:")[1].split("\t")[0]
                return sorted(x for x in lines if not x.startswith("@"))

            rbao_cli = ["-rbao"] if args.rbao else []
            switches = {"plain": var["cli"] + rbao_cli, "extend": var["cli"] + rbao_cli + ["-extend"]}
            ref = {name: run_reference(extra, args.nproc) for name, extra in switches.items()}
            # the same once more on ONE thread, for the (rare) input on which the binary's own lines change from run to
            # run under -nproc > 1 (its interval finder reads one element past the end of a match list, i.e. whatever the
            # thread's previous read left there): a difference only counts when the one-thread run shows it too
            # (run lazily: only a difference asks for it, and by then the files are gone -- so keep their text)
            fasta_text = {k: open(os.path.join(tmp, k)).read() for k in (reads_file, "genome.fasta")}
            ref1 = {}

            def one_thread(name):
                if args.nproc <= 1:
                    return ref[name]
                if name not in ref1:
                    with tempfile.TemporaryDirectory() as tmp1:
                        for k, v in fasta_text.items():
                            open(os.path.join(tmp1, k), "w").write(v)
                        path = os.path.join(tmp1, "out.m4")
                        p1 = subprocess.run([blasr, os.path.join(tmp1, reads_file), os.path.join(tmp1, "genome.fasta")] +
                                            (["-sam"] if args.sam else ["-m", "4"]) + ["-bestn", "1", "-nproc", "1", "-out", path] +
                                            switches[name], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                        ref1[name] = None if p1.returncode != 0 else sorted(
                            x for x in open(path).read().splitlines() if not x.startswith("@"))
                return ref1[name]
            if args.dropin:
                for name, extra in switches.items():
                    # (odd iterations: -gpuBatch, MapReads pulling blocks of reads through ONE library call each)
                    # (under --rbao always: without -gpuBatch the branch between the anchors stays the reference's own CPU code)
                    got = run_reference(extra + ["-useGPU"] + (["-gpuBatch", str(int(rng.integers(1, 300)))] if it % 2 or args.rbao else []), 2,
                                        os.path.join(ROOT, "oracle", "_ref", "blasrmc_gpu"))
                    if got is None or ref[name] is None:
                        totals["dropin_failed"] = totals.get("dropin_failed", 0) + (got is None)
                        continue
                    totals["dropin_lines"] = totals.get("dropin_lines", 0) + len(got)
                    d = len(set(got) ^ set(ref[name]))
                    if d and one_thread(name) is not None and not (set(got) ^ set(one_thread(name))):
                        d = 0
                    totals["dropin_bad"] = totals.get("dropin_bad", 0) + d
                    if d and len(examples) < 6:
                        examples.append(dict(it=it, mode=name, variant=var["name"], leg="dropin",
                                             only_reference=[x[:300] for x in sorted(set(ref[name]) - set(got))[:2]],
                                             only_dropin=[x[:300] for x in sorted(set(got) - set(ref[name]))[:2]]))
        for _rep in range(args.repeat if args.only_it >= 0 else 1):
            # the library, from the reads alone
            parts, starts = [], [0]
            for c in contigs:
                parts += [c, np.array([ord("N")], np.uint8)]
                starts.append(starts[-1] + len(c) + 1)
            genome = np.concatenate(parts).astype(np.uint8)
            starts = np.array(starts, dtype=np.uint32)
            ctx = blasr_amd.Context(0)
            ctx.upload_genome(genome, starts)
            ctx.build_suffix_array(8, download=False)  # what blasrmc builds on the fly (Blasr.cpp:4407-4420)
            rt = np.zeros(len(reads), dtype=blasr_amd.READ_DTYPE)
            off = 0
            for i, rd in enumerate(reads):
                rt[i] = (off, len(rd), 0, len(rd), 0)
                off += len(rd)
            pool = np.concatenate(reads)
            found = blasr_amd.candidate_intervals(ctx, genome, starts, ctx.count_tuples(8), pool, rt,
                                                  anchor_params=ctx.anchor_params(**var.get("anchor", {})),
                                                  interval_params=hip.interval_params(**var.get("interval", {})),
                                                  chains=args.rbao)
            ivs = found[0]
            chains = (found[3], found[4]) if args.rbao else None
            mine = {}
            mkw = dict(var.get("map", {}))
            if args.rbao:
                mkw["refine_between_anchors_only"] = 1
            if "sdp" in var:
                mkw["sdp"] = hip.SdpParams(**var["sdp"])
            if "score" in var:
                mkw["score"] = hip.ScoreParams(**var["score"])
            for name, prm in (("plain", ctx.map_params(**mkw)), ("extend", ctx.map_params(extend_alignments=1, **mkw))):
                got, unaligned = [], []
                for r, hits in enumerate(blasr_amd.map_reads(ctx, len(genome), starts, pool, rt, ivs, bestn=var.get("bestn", 1),
                                                             params=prm, min_align_length=var.get("min_align_length", 0),
                                                             min_map_qv=var.get("min_map_qv", 0), for_picard=args.sam,
                                                             unaligned=unaligned, chains=chains)):
                    nm = "%s/0_%d" % (names[r], len(reads[r]))
                    if args.sam:  # -sam switches forPicard on (MappingParameters.h:488-491)
                        got += [blasr_amd.sam_record(h, nm, "contig%d" % h["contig"], read_group[0], reads[r],
                                                     quality=None if quals is None else quals[r],
                                                     clipping=var.get("clipping", hip.CLIP_NONE)).rstrip("\n") for h in hits]
                    else:
                        got += [blasr_amd.m4_record(h, nm, "contig%d" % h["contig"]).rstrip("\n") for h in hits]
                if args.sam:  # PrintUnalignedRead for the reads MapRead left without an alignment
                    got += [hip.sam_unaligned_line("%s/0_%d" % (names[r], len(reads[r])), read_group[0],
                                                   reads[r], clipping=var.get("clipping", hip.CLIP_NONE)).rstrip("\n")
                            for r in unaligned]
                mine[name] = sorted(got)
            ctx.close()
            totals["iters"] += 1
            totals["reads"] += len(reads)
            for name in ("plain", "extend"):
                if ref[name] is None:
                    totals.setdefault("reference_failed_" + name, 0)
                    totals["reference_failed_" + name] += 1
                    continue
                totals["lines_" + name] += len(ref[name])
                a, b = set(ref[name]), set(mine[name])
                diff = len(a ^ b)
                if diff and one_thread(name) is not None and not (set(one_thread(name)) ^ b):
                    totals["reference_unstable_under_nproc"] = totals.get("reference_unstable_under_nproc", 0) + 1
                    diff = 0
                if diff and not args.fastq and not args.sam_reads:
                    # The same leak carries over from one read to the next inside ONE thread as well (what a read's match list
                    # leaves behind is what the next read's interval finder finds past the end of its own): a read whose lines
                    # differ is mapped once more ALONE, by a fresh process; a difference only counts when that run shows it too.
                    # (-sam: the file keeps its name, the read group's ID is made from it; the lone run's record of a read it
                    #  leaves unaligned counts as a line like any other.)
                    stay = 0
                    for rid in sorted({x.split()[0].split("/")[0] for x in a ^ b}):
                        r = int(rid[4:])
                        with tempfile.TemporaryDirectory() as tmp1:
                            open(os.path.join(tmp1, "genome.fasta"), "w").write(fasta_text["genome.fasta"])
                            one = os.path.join(tmp1, "one.fasta")
                            open(one, "w").write(">%s\n%s\n" % (names[r], bytes(reads[r]).decode()))
                            path = os.path.join(tmp1, "out.m4")
                            p1 = subprocess.run([blasr, one, os.path.join(tmp1, "genome.fasta")] + (["-sam"] if args.sam else ["-m", "4"]) +
                                                ["-bestn", "1", "-nproc", "1", "-out", path] + switches[name],
                                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                            alone = ({x for x in open(path).read().splitlines() if not x.startswith("@")}
                                     if p1.returncode == 0 and os.path.exists(path) else None)
                        ours = {x for x in b if x.split()[0].split("/")[0] == rid}
                        if args.sam and alone is not None:  # (the read group's ID is an MD5 of the file's PATH: not part of the comparison)
                            import re
                            alone = {re.sub(r"\tRG:Z:[0-9a-f]+", "", x) for x in alone}
                            ours = {re.sub(r"\tRG:Z:[0-9a-f]+", "", x) for x in ours}
                        if alone is not None and alone == ours:
                            totals["reference_depends_on_the_read_before"] = totals.get("reference_depends_on_the_read_before", 0) + 1
                        else:
                            stay += len({x for x in a ^ b if x.split()[0].split("/")[0] == rid})
                    diff = stay
                totals["bad_" + name] += diff
                if args.only_it >= 0:
                    print("[rep %d] %s: %d differ %s" % (_rep, name, diff, sorted({x.split("\t")[0] for x in a ^ b})[:4]), file=sys.stderr)
                v = by_variant.setdefault(var["name"], dict(lines=0, bad=0))
                v["lines"] += len(ref[name])
                v["bad"] += diff
                if diff and args.sam and len(examples) < 6:  # which fields of a record differ
                    byname = {x.split("\t")[0]: x for x in b - a}
                    for x in sorted(a - b)[:2]:
                        y = byname.get(x.split("\t")[0])
                        if y is not None:
                            fx, fy = x.split("\t"), y.split("\t")
                            examples.append(dict(it=it, mode=name, variant=var["name"], fields=[
                                (i, u[:60], v[:60]) for i, (u, v) in enumerate(zip(fx, fy)) if u != v], n=(len(fx), len(fy))))
                if diff and len(examples) < 6:
                    examples.append(dict(it=it, mode=name, variant=var["name"], only_reference=[x[:500] for x in sorted(a - b)[:2]], only_library=[x[:500] for x in sorted(b - a)[:2]]))
        if ref["plain"] is not None and ref["extend"] is not None:
            totals["changed_by_extend"] += len(set(ref["plain"]) ^ set(ref["extend"])) // 2
    totals["seconds"] = round(time.time() - t0, 1)
    totals["seed"] = args.seed
    totals["format"] = "sam" if args.sam else "m4"
    totals["rbao"] = bool(args.rbao)
    totals["by_variant"] = by_variant
    totals["examples"] = examples
    print(json.dumps(totals))


if __name__ == "__main__":
    main()
