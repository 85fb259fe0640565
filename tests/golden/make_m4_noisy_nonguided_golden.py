#!/usr/bin/env python3
"""The noisy end-to-end fixture under -noUseGuidedAlign: the reads and the genome of m4_noisy_golden.npz mapped by the
REFERENCE BINARY (oracle/_ref/blasrmc -m 4 -bestn 1 -nproc 1 -noUseGuidedAlign), RefineAlignment's k-band branch
(Blasr.cpp:917-1046).  The reference computes the score of that branch with a score matrix it never initialises
(Blasr.cpp:689-690), so the score column is stack garbage, and so is everything decided by it (which candidate is
printed, whether a read passes the score cut-off at all: 47 to 54 of the 80 reads print, run by run): the fixture is good for the
columns of the lines that do print -- identity and the four coordinates -- and nothing else.
Run from the repo root:  python tests/golden/make_m4_noisy_nonguided_golden.py"""
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    g = np.load(os.path.join(HERE, "m4_noisy_golden.npz"))
    reads, read_lens = g["reads"], g["read_lens"]
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "genome.fasta"), "w") as f:
            for i, c in enumerate((g["contig0"], g["contig1"])):
                s = bytes(c).decode()
                f.write(">contig%d\n" % i)
                for k in range(0, len(s), 60):
                    f.write(s[k:k + 60] + "\n")
        off = 0
        with open(os.path.join(tmp, "reads.fasta"), "w") as f:
            for r, n in enumerate(read_lens):
                f.write(">read%d/%d/0_%d\n%s\n" % (r, r, int(n), bytes(reads[off:off + int(n)]).decode()))
                off += int(n)
        blasr = os.path.join(ROOT, "oracle", "_ref", "blasrmc")
        out = {}
        for name, extra in (("plain", []), ("nonguided", ["-noUseGuidedAlign"])):
            path = os.path.join(tmp, name + ".m4")
            subprocess.run([blasr, os.path.join(tmp, "reads.fasta"), os.path.join(tmp, "genome.fasta"), "-m", "4",
                            "-bestn", "1", "-nproc", "1", "-out", path] + extra, check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            out[name] = open(path).read()
    assert out["plain"] == bytes(g["lines"]).decode(), "the fixture's inputs no longer reproduce its own lines"
    print("%d lines with the guided refiner, %d with the k-band one" % (len(out["plain"].splitlines()),
                                                                         len(out["nonguided"].splitlines())))
    np.savez_compressed(os.path.join(HERE, "m4_noisy_nonguided_golden.npz"),
                        lines=np.frombuffer(out["nonguided"].encode(), dtype=np.uint8))


if __name__ == "__main__":
    main()
