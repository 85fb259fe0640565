#!/usr/bin/env python3
"""End-to-end golden on NOISY reads: PacBio-like 1.5-4 kb reads against a two-contig genome with planted repeats,
mapped by the REFERENCE BINARY (oracle/_ref/blasrmc -m 4 -bestn 1 -nproc 1 -V 1, and once more with -sam).  Besides the -m 4 lines and the SAM records the
fixture keeps the candidate intervals the reference derived from its anchors (the ` intv:` lines of -V 1,
Blasr.cpp:2508-2517) -- the one thing the device path does not compute itself (anchoring, SURVEY next-1).
Run from the repo root:  python tests/golden/make_m4_noisy_golden.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from blasr_amd import synth  # noqa: E402


def main():
    rng = np.random.default_rng(20261005)
    c0 = synth.random_genome(150000, 21)
    c1 = synth.random_genome(100000, 22)
    # planted repeats: two 6 kb pieces of contig0 reappear in contig1 with ~2 % and ~6 % divergence, one of them
    # reverse-complemented
    rep_a = np.ascontiguousarray(synth.mutate(c0[20000:26000], rng, 0.005, 0.005, 0.01))
    rep_b = synth.revcomp(np.ascontiguousarray(synth.mutate(c0[90000:96000], rng, 0.02, 0.02, 0.02)))
    c1 = np.concatenate([c1[:30000], rep_a, c1[30000:70000], rep_b, c1[70000:]])
    contigs = [c0, np.ascontiguousarray(c1)]
    n_reads = 80
    reads = []
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "genome.fasta"), "w") as f:
            for i, c in enumerate(contigs):
                s = bytes(c).decode()
                f.write(">contig%d\n" % i)
                for k in range(0, len(s), 60):
                    f.write(s[k:k + 60] + "\n")
        with open(os.path.join(tmp, "reads.fasta"), "w") as f:
            for r in range(n_reads):
                rl = int(rng.integers(1500, 4000))
                if r % 2 == 0:  # from a repeat
                    ci, lo, hi = [(0, 20000, 26000), (0, 90000, 96000), (1, 30000, 30000 + len(rep_a)),
                                  (1, 70000 + len(rep_a), 70000 + len(rep_a) + len(rep_b))][(r // 2) % 4]
                    pos = int(rng.integers(max(lo - rl // 3, 0), min(hi - rl // 2, len(contigs[ci]) - rl)))
                else:
                    ci = int(rng.integers(0, 2))
                    pos = int(rng.integers(0, len(contigs[ci]) - rl))
                s = contigs[ci][pos:pos + rl]
                if rng.integers(0, 2):
                    s = synth.revcomp(s)
                rd = np.ascontiguousarray(synth.mutate(s, rng, 0.11, 0.04, 0.01))
                f.write(">read%d/%d/0_%d\n%s\n" % (r, r, len(rd), bytes(rd).decode()))
                reads.append(rd)
        out = os.path.join(tmp, "out.m4")
        p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "blasrmc"), os.path.join(tmp, "reads.fasta"),
                            os.path.join(tmp, "genome.fasta"), "-m", "4", "-bestn", "1", "-nproc", "1", "-V", "1",
                            "-out", out], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        lines = open(out).read()
        sam_out = os.path.join(tmp, "out.sam")
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "blasrmc"), os.path.join(tmp, "reads.fasta"),
                        os.path.join(tmp, "genome.fasta"), "-sam", "-bestn", "1", "-nproc", "1", "-out", sam_out],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sam = open(sam_out).read()
    sam_header = "".join(x + "\n" for x in sam.splitlines() if x.startswith("@") and not x.startswith("@PG"))
    sam_lines = "".join(x + "\n" for x in sam.splitlines() if not x.startswith("@"))
    assert len(sam_lines.splitlines()) == n_reads
    # the intervals, read by read (one ` intv: index ...` header per read)
    intervals, owner = [], []
    r = -1
    for ln in p.stdout.decode(errors="replace").splitlines():
        if ln.startswith(" intv: index"):
            r += 1
        elif ln.startswith(" intv: "):
            f = ln.split()
            intervals.append([int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[8])])  # start end qStart qEnd strand
            owner.append(r)
    assert r + 1 == n_reads, (r, n_reads)
    print("reads", n_reads, "intervals", len(intervals), "lines", len(lines.splitlines()),
          "reads with several candidates", sum(1 for k in range(n_reads) if owner.count(k) > 1))
    np.savez_compressed(os.path.join(HERE, "m4_noisy_golden.npz"), contig0=contigs[0], contig1=contigs[1],
                        reads=np.concatenate(reads), read_lens=np.array([len(x) for x in reads], dtype=np.int32),
                        intervals=np.array(intervals, dtype=np.int64), owner=np.array(owner, dtype=np.int32),
                        lines=np.frombuffer(lines.encode(), dtype=np.uint8),
                        sam_header=np.frombuffer(sam_header.encode(), dtype=np.uint8),
                        sam_lines=np.frombuffer(sam_lines.encode(), dtype=np.uint8))
    print(lines.splitlines()[0])


if __name__ == "__main__":
    main()
