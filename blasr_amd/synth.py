"""Seeded synthetic inputs for the refine stage.

No E. coli / hg19 FASTA exists offline, so the workloads are length-matched
i.i.d. genomes and PacBio-like reads drawn from them:

* genome  -- uniform ACGT of a given length (E. coli K12 MG1655 = 4,641,652)
* read    -- a window of the genome (either strand) pushed through a per-base
             error channel: deletion, substitution, geometric insertions
* task    -- what AlignIntervals hands to SDPAlign for one candidate interval
             (alignment/Blasr.cpp:1247-1336): the forward read as query and the
             genome window, stretched by 1.3x the unanchored read ends and
             clipped to the contig, as target (reverse-complemented for
             reverse-strand hits)

Everything is numpy (uint8 ASCII) and deterministic in the seed.
"""
import numpy as np

ECOLI_K12_LENGTH = 4641652
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[_a] = _b


def random_genome(length, seed):
    rng = np.random.default_rng(seed)
    return _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]


def revcomp(seq):
    return _COMP[seq[::-1]]


def mutate(template, rng, ins=0.11, dele=0.04, sub=0.01):
    """Error channel: per template base delete w.p. `dele`, else substitute w.p.
    `sub`; after every template base insert K ~ Geometric uniform bases with
    P(K >= 1) = `ins`."""
    n = len(template)
    u = rng.random(n)
    keep = u >= dele
    base = template.copy()
    do_sub = (u >= dele) & (u < dele + sub)
    nsub = int(do_sub.sum())
    if nsub:
        code = np.searchsorted(_ACGT, base[do_sub])  # ACGT is sorted in ASCII
        base[do_sub] = _ACGT[(code + rng.integers(1, 4, size=nsub)) & 3]
    k = rng.geometric(1.0 - ins, size=n) - 1 if ins > 0 else np.zeros(n, dtype=np.int64)
    emit = keep.astype(np.int64) + k
    total = int(emit.sum())
    out = _ACGT[rng.integers(0, 4, size=total, dtype=np.uint8)]
    starts = np.cumsum(emit) - emit
    kept_idx = np.nonzero(keep)[0]
    out[starts[kept_idx]] = base[kept_idx]
    return out


def make_task(genome, rng, read_len, ins=0.11, dele=0.04, sub=0.01, max_unanchored=300, stretch=1.3):
    """One (query, target) pair following the window rule of AlignIntervals.

    The candidate interval is the read's true origin minus the unanchored ends
    (uniform 0..max_unanchored bases of the read at each side have no anchors),
    then stretched back by `stretch` x those ends (Blasr.cpp:1247-1265,
    approximateMaxInsertionRate = 1.3) and clipped to the genome."""
    glen = len(genome)
    pos = int(rng.integers(0, glen - read_len))
    strand = int(rng.integers(0, 2))
    template = genome[pos:pos + read_len]
    if strand:
        template = revcomp(template)
    read = mutate(template, rng, ins, dele, sub)
    # unanchored read ends (in read bases) and the genome span they cover
    ua0 = int(rng.integers(0, max_unanchored + 1))
    ua1 = int(rng.integers(0, max_unanchored + 1))
    scale = read_len / max(len(read), 1)
    g0, g1 = int(ua0 * scale), int(ua1 * scale)
    if strand:
        g0, g1 = g1, g0
        ua0, ua1 = ua1, ua0
    start = pos + g0
    end = pos + read_len - g1
    start = max(0, start - int(ua0 * stretch))
    end = min(glen, end + int(ua1 * stretch))
    target = genome[start:end]
    if strand:
        target = revcomp(target)
    return read, np.ascontiguousarray(target), dict(pos=pos, strand=strand, start=start, end=end)


def make_pair(rng, qlen, ins=0.11, dele=0.04, sub=0.01, flank=(0, 200), big_indel=None, n_run=None):
    """A standalone (query, target) pair for function-level tests: a random
    template, optional planted long insertion/deletion, optional run of N in
    the target, random flanks on the target."""
    template = _ACGT[rng.integers(0, 4, size=qlen, dtype=np.uint8)]
    read_src = template
    if big_indel:
        kind, size = big_indel
        at = int(rng.integers(qlen // 4, 3 * qlen // 4))
        if kind == "ins":  # extra bases in the read
            extra = _ACGT[rng.integers(0, 4, size=size, dtype=np.uint8)]
            read_src = np.concatenate([template[:at], extra, template[at:]])
        else:  # bases of the target missing from the read
            read_src = np.concatenate([template[:at], template[at + size:]])
    read = mutate(read_src, rng, ins, dele, sub)
    f0 = int(rng.integers(flank[0], flank[1] + 1))
    f1 = int(rng.integers(flank[0], flank[1] + 1))
    target = np.concatenate([
        _ACGT[rng.integers(0, 4, size=f0, dtype=np.uint8)], template,
        _ACGT[rng.integers(0, 4, size=f1, dtype=np.uint8)]])
    if n_run:
        at = int(rng.integers(0, max(1, len(target) - n_run)))
        target[at:at + n_run] = ord("N")
    return np.ascontiguousarray(read), np.ascontiguousarray(target)
