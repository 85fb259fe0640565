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


# ---------------------------------------------------------------------------
# Bulk workload for bench.py (torch, runs on the GPU so that 1 Gbase of reads
# is generated in seconds; the same code runs on CPU tensors for the tests).
# ---------------------------------------------------------------------------
def make_refine_workload(n_reads, read_len, seed, device, genome_len=ECOLI_K12_LENGTH, ins=0.11, dele=0.04,
                         sub=0.01, slab=4096):
    """Refine-stage inputs for `n_reads` synthetic PacBio-like reads.

    Per read: a `read_len` window of an i.i.d. genome (either strand) is pushed
    through the error channel of `mutate`.  The candidate handed to the refine
    stage is what RefineAlignment builds at alignment/Blasr.cpp:850-859: query
    and target trimmed to the extent of the guide, guide blocks relative to the
    first block.  The guide is the simulated alignment itself (every ungapped
    run of the true path is one block), i.e. the block structure SDPAlign +
    SWAlign gap filling recovers; bench.py states this in its `data` field.

    Returns a dict of torch tensors laid out as blasr_hip_refine_batch wants:
    seq (uint8 pool), q_off/t_off (int64), q_len/t_len (int32), blocks (int32,
    3 per block), blk_off (int64), n_blocks (int32).
    """
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    comp = torch.tensor([3, 2, 1, 0], dtype=torch.int64, device=device)
    genome = torch.randint(0, 4, (genome_len,), generator=gen, device=device, dtype=torch.int64)
    q_parts, t_parts, blk_parts = [], [], []
    q_len_all, t_len_all, nblk_all = [], [], []
    L = read_len
    ar = torch.arange(L, device=device)
    for s0 in range(0, n_reads, slab):
        n = min(slab, n_reads - s0)
        pos = torch.randint(0, genome_len - L, (n,), generator=gen, device=device)
        strand = torch.randint(0, 2, (n,), generator=gen, device=device)
        idx = pos[:, None] + ar[None, :]
        idx = torch.where(strand[:, None] == 1, pos[:, None] + (L - 1 - ar)[None, :], idx)
        tmpl = genome[idx]
        tmpl = torch.where(strand[:, None] == 1, comp[tmpl], tmpl)  # [n, L] codes 0..3
        u = torch.rand((n, L), generator=gen, device=device)
        keep = u >= dele
        do_sub = keep & (u < dele + sub)
        shift = torch.randint(1, 4, (n, L), generator=gen, device=device)
        base = torch.where(do_sub, (tmpl + shift) & 3, tmpl)
        if ins > 0:
            v = torch.rand((n, L), generator=gen, device=device).clamp_min(1e-12)
            k = torch.floor(torch.log(v) / np.log(ins)).to(torch.int64)  # P(K >= j) = ins**j
        else:
            k = torch.zeros((n, L), dtype=torch.int64, device=device)
        emit = keep.to(torch.int64) + k
        qpos = torch.cumsum(emit, dim=1) - emit  # read position of each kept template base
        rlen = emit.sum(dim=1)
        # ungapped runs of kept bases
        prev_ok = torch.zeros_like(keep)
        prev_ok[:, 1:] = keep[:, :-1] & (k[:, :-1] == 0)
        start = keep & ~prev_ok
        nxt_ok = torch.zeros_like(keep)
        nxt_ok[:, :-1] = keep[:, 1:] & (k[:, :-1] == 0)
        end = keep & ~nxt_ok
        nblk = start.sum(dim=1)
        srow, scol = torch.nonzero(start, as_tuple=True)
        erow, ecol = torch.nonzero(end, as_tuple=True)
        bq = qpos[srow, scol]
        bt = scol
        bl = ecol - scol + 1
        # trim to the guide extent: first block start .. last block end (Blasr.cpp:850-859)
        blk_off = torch.cumsum(nblk, 0) - nblk
        first_q, first_t = bq[blk_off], bt[blk_off]
        last = blk_off + nblk - 1
        last_qe, last_te = bq[last] + bl[last], bt[last] + bl[last]
        q_len = last_qe - first_q
        t_len = last_te - first_t
        # read pool of this slab
        roff = torch.cumsum(rlen, 0) - rlen
        total = int(rlen.sum())
        reads = torch.randint(0, 4, (total,), generator=gen, device=device)
        krow, kcol = torch.nonzero(keep, as_tuple=True)
        reads[roff[krow] + qpos[krow, kcol]] = base[krow, kcol]
        # gather the trimmed query / target slices
        qo = torch.cumsum(q_len, 0) - q_len
        qi = torch.repeat_interleave(roff + first_q - qo, q_len) + torch.arange(int(q_len.sum()), device=device)
        q_parts.append(acgt[reads[qi]])
        to = torch.cumsum(t_len, 0) - t_len
        rows = torch.repeat_interleave(torch.arange(n, device=device), t_len)
        cols = torch.arange(int(t_len.sum()), device=device) - torch.repeat_interleave(to - first_t, t_len)
        t_parts.append(acgt[tmpl[rows, cols]])
        brow = srow
        blk_parts.append(torch.stack([bq - first_q[brow], bt - first_t[brow], bl], dim=1).to(torch.int32))
        q_len_all.append(q_len)
        t_len_all.append(t_len)
        nblk_all.append(nblk)
    q_len = torch.cat(q_len_all)
    t_len = torch.cat(t_len_all)
    nblk = torch.cat(nblk_all)
    qpool, tpool = torch.cat(q_parts), torch.cat(t_parts)
    q_off = torch.cumsum(q_len, 0) - q_len
    t_off = torch.cumsum(t_len, 0) - t_len + qpool.numel()
    return dict(seq=torch.cat([qpool, tpool]).contiguous(), q_off=q_off.to(torch.int64).contiguous(),
                q_len=q_len.to(torch.int32).contiguous(), t_off=t_off.to(torch.int64).contiguous(),
                t_len=t_len.to(torch.int32).contiguous(),
                blocks=torch.cat(blk_parts).reshape(-1).contiguous(),
                blk_off=(torch.cumsum(nblk, 0) - nblk).to(torch.int64).contiguous(),
                n_blocks=nblk.to(torch.int32).contiguous())


def make_sdp_workload(n_reads, read_len, seed, device, genome_len=ECOLI_K12_LENGTH, ins=0.11, dele=0.04, sub=0.01,
                      max_unanchored=300, slab=4096):
    """SDP-stage inputs for `n_reads` synthetic PacBio-like reads: what AlignIntervals hands to SDPAlign
    (alignment/Blasr.cpp:1311-1336, :1716-1722) for the one true candidate interval of every read.

    query  = the whole read (forward as sequenced);
    target = the genome window of the read's origin, extended on each side by the 0.3 x (unanchored read
             end) that the 1.3x stretch of Blasr.cpp:1247-1265 adds beyond the read's own footprint
             (unanchored ends uniform in 0..max_unanchored), clipped to the genome, reverse-complemented
             for reverse-strand reads.

    Returns torch tensors laid out as blasr_hip_sdp_batch wants: seq, q_off, q_len, t_off, t_len."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    comp = torch.tensor([3, 2, 1, 0], dtype=torch.int64, device=device)
    genome = torch.randint(0, 4, (genome_len,), generator=gen, device=device, dtype=torch.int64)
    L = read_len
    ar = torch.arange(L, device=device)
    q_parts, t_parts, q_lens, t_lens = [], [], [], []
    for s0 in range(0, n_reads, slab):
        n = min(slab, n_reads - s0)
        pos = torch.randint(0, genome_len - L, (n,), generator=gen, device=device)
        strand = torch.randint(0, 2, (n,), generator=gen, device=device)
        idx = pos[:, None] + ar[None, :]
        idx = torch.where(strand[:, None] == 1, pos[:, None] + (L - 1 - ar)[None, :], idx)
        tmpl = genome[idx]
        tmpl = torch.where(strand[:, None] == 1, comp[tmpl], tmpl)
        u = torch.rand((n, L), generator=gen, device=device)
        keep = u >= dele
        do_sub = keep & (u < dele + sub)
        shift = torch.randint(1, 4, (n, L), generator=gen, device=device)
        base = torch.where(do_sub, (tmpl + shift) & 3, tmpl)
        if ins > 0:
            v = torch.rand((n, L), generator=gen, device=device).clamp_min(1e-12)
            k = torch.floor(torch.log(v) / np.log(ins)).to(torch.int64)
        else:
            k = torch.zeros((n, L), dtype=torch.int64, device=device)
        emit = keep.to(torch.int64) + k
        qpos = torch.cumsum(emit, dim=1) - emit
        rlen = emit.sum(dim=1)
        roff = torch.cumsum(rlen, 0) - rlen
        reads = torch.randint(0, 4, (int(rlen.sum()),), generator=gen, device=device)
        krow, kcol = torch.nonzero(keep, as_tuple=True)
        reads[roff[krow] + qpos[krow, kcol]] = base[krow, kcol]
        q_parts.append(acgt[reads])
        q_lens.append(rlen)
        # window = footprint extended by 0.3 x unanchored ends, clipped
        ext0 = (torch.randint(0, max_unanchored + 1, (n,), generator=gen, device=device) * 3) // 10
        ext1 = (torch.randint(0, max_unanchored + 1, (n,), generator=gen, device=device) * 3) // 10
        w0 = torch.clamp(pos - ext0, min=0)
        w1 = torch.clamp(pos + L + ext1, max=genome_len)
        wl = w1 - w0
        wo = torch.cumsum(wl, 0) - wl
        rows = torch.repeat_interleave(torch.arange(n, device=device), wl)
        off = torch.arange(int(wl.sum()), device=device) - wo[rows]
        gidx = torch.where(strand[rows] == 1, w1[rows] - 1 - off, w0[rows] + off)
        win = genome[gidx]
        win = torch.where(strand[rows] == 1, comp[win], win)
        t_parts.append(acgt[win])
        t_lens.append(wl)
    q_len, t_len = torch.cat(q_lens), torch.cat(t_lens)
    qpool, tpool = torch.cat(q_parts), torch.cat(t_parts)
    return dict(seq=torch.cat([qpool, tpool]).contiguous(),
                q_off=(torch.cumsum(q_len, 0) - q_len).to(torch.int64).contiguous(),
                q_len=q_len.to(torch.int32).contiguous(),
                t_off=(torch.cumsum(t_len, 0) - t_len + qpool.numel()).to(torch.int64).contiguous(),
                t_len=t_len.to(torch.int32).contiguous())
