#!/usr/bin/env python3
"""The noisy end-to-end fixture once more under -rbao: the reads and the genome of m4_noisy_golden.npz mapped by the
REFERENCE BINARY with `-rbao` and with `-rbao -extend` (oracle/_ref/blasrmc -m 4 -bestn 1 -nproc 1): AlignIntervals
aligns between the chained anchors of each interval (Blasr.cpp:1354-1712) instead of running SDPAlign over the window.
Keeps the -m 4 lines only (the inputs are those of m4_noisy_golden.npz).  Run from the repo root:
    python tests/golden/make_m4_noisy_rbao_golden.py"""
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
        for name, extra in (("plain", []), ("rbao", ["-rbao"]), ("rbao_extend", ["-rbao", "-extend"])):
            path = os.path.join(tmp, name + ".m4")
            subprocess.run([blasr, os.path.join(tmp, "reads.fasta"), os.path.join(tmp, "genome.fasta"), "-m", "4",
                            "-bestn", "1", "-nproc", "1", "-out", path] + extra, check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            out[name] = open(path).read()
    assert out["plain"] == bytes(g["lines"]).decode(), "the fixture's inputs no longer reproduce its own lines"
    a, b, c = (out[k].splitlines() for k in ("plain", "rbao", "rbao_extend"))
    print("%d lines by default, %d under -rbao (%d of them among the default's), %d under -rbao -extend" %
          (len(a), len(b), len(set(a) & set(b)), len(c)))
    np.savez_compressed(os.path.join(HERE, "m4_noisy_rbao_golden.npz"),
                        lines=np.frombuffer(out["rbao"].encode(), dtype=np.uint8),
                        lines_extend=np.frombuffer(out["rbao_extend"].encode(), dtype=np.uint8))


if __name__ == "__main__":
    main()
