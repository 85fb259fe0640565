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


# ---------------------------------------------------------------------------
# configs[2]: reads against an hg19-sized genome with planted repeat families, 1-10 candidate intervals per
# read (bench.py).  Everything is generated on the device; nothing here aligns anything.
# ---------------------------------------------------------------------------
# chr1..22, X, Y of hg19 (GRCh37); the contigs sit in one concatenated genome, each followed by one N, as the
# reference lays a multi-FASTA out (SequenceIndexDatabase::seqStartPos)
HG19_CHROM_LENGTHS = (249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022,
                      141213431, 135534747, 135006516, 133851895, 115169878, 107349540, 102531392, 90354753,
                      81195210, 78077248, 59128983, 63025520, 48129895, 51304566, 155270560, 59373566)


def contig_table(lengths):
    """seqStartPos of contigs of the given lengths, one separator base after each."""
    starts = [0]
    for n in lengths:
        starts.append(starts[-1] + int(n) + 1)
    return np.array(starts, dtype=np.int64)


def make_repeat_genome(contig_lengths, seed, device, families=2000, unit=(6000, 20000), copies=(2, 10),
                       divergence=(0.01, 0.06)):
    """An i.i.d. ACGT genome (uint8 ASCII, one N after every contig) with `families` repeat families planted: every
    family is one random unit of `unit` bases that occurs `copies` times, each copy diverged from the unit by
    substitutions at a rate drawn from `divergence`, on either strand, in its own slot of the genome (copies never
    overlap).  Returns (genome tensor, seqStartPos int64 array, copy table): the copy table is a dict of int64
    tensors sorted by position -- start, family, strand, and per family its unit length."""
    import torch

    starts = contig_table(contig_lengths)
    total = int(starts[-1])
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    genome = torch.empty(total, dtype=torch.uint8, device=device)
    step = 1 << 27
    for a in range(0, total, step):
        b = min(total, a + step)
        genome[a:b] = lut[torch.randint(0, 4, (b - a,), generator=gen, device=device)]
    genome[torch.as_tensor(starts[1:] - 1, device=device)] = ord("N")
    # ---- repeat families
    cpu = np.random.default_rng(seed)
    n_copies = cpu.integers(copies[0], copies[1] + 1, size=families)
    units = cpu.integers(unit[0], unit[1] + 1, size=families)
    fam_of = np.repeat(np.arange(families), n_copies)
    cpu.shuffle(fam_of)
    n_total = len(fam_of)
    slot = total // max(n_total, 1)
    assert slot > 2 * unit[1], "genome too small for the repeat families"
    pos = (np.arange(n_total, dtype=np.int64) * slot + cpu.integers(0, slot - unit[1] - 1, size=n_total)).astype(np.int64)
    strand = cpu.integers(0, 2, size=n_total)
    div = cpu.uniform(divergence[0], divergence[1], size=n_total)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    comp[:] = ord("N")
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    unit_seq = {}
    for c in range(n_total):
        f = int(fam_of[c])
        U = int(units[f])
        if f not in unit_seq:
            unit_seq[f] = lut[torch.randint(0, 4, (U,), generator=gen, device=device)]
        s = unit_seq[f].clone()
        hit = torch.rand(U, generator=gen, device=device) < float(div[c])
        sub = lut[torch.randint(0, 4, (U,), generator=gen, device=device)]
        s = torch.where(hit, sub, s)
        if strand[c]:
            s = comp[s.long()].flip(0)
        genome[int(pos[c]): int(pos[c]) + U] = s
    table = dict(start=torch.as_tensor(pos, device=device), family=torch.as_tensor(fam_of.astype(np.int64), device=device),
                 strand=torch.as_tensor(strand.astype(np.int64), device=device),
                 unit=torch.as_tensor(units.astype(np.int64), device=device))
    return genome, starts, table


def make_map_workload(genome, copies, n_reads, read_len, seed, device, ins=0.11, dele=0.04, sub=0.01,
                      max_unanchored=300, min_overlap=1000, spurious=2.0, max_candidates=10, loose=0.0,
                      spurious_span=None, slab=4096):
    """`n_reads` PacBio-like reads drawn uniformly from `genome` (either strand, the error channel of `mutate`) and
    what anchoring would hand MapRead for each: its true locus, the matching stretch of every other copy of a
    repeat family the read overlaps by >= min_overlap bases, and Poisson(spurious) chance clusters elsewhere in the
    genome (a short anchored stretch of the read against unrelated sequence) -- at most `max_candidates` intervals
    per read (params.nCandidates), in random order.  loose > 0 widens the true locus' interval by up to loose x the
    read length on either side (anchor clusters that reach beyond the read: target windows much longer than the
    query).  spurious_span = (q_lo, q_hi, g_lo, g_hi), fractions of the read length, makes the chance clusters LONG
    (read span uniform in [q_lo, q_hi], genome span in [g_lo, g_hi]): what the reference's anchoring reports for 30 kb
    reads at 15 % indels -- eight or nine such intervals per read next to the true one (blasrmc -V 1; they are what
    costs the reference 455 ms of SDPAlign per read); None: a short stretch of 300-1500 bases.
    Interval coordinates follow WeightedInterval
    (datastructures/anchoring/WeightedInterval.h:12-29): genome span on the forward genome, read span on the strand
    of the interval.

    Returns dict: pool (uint8 tensor), reads (READ_DTYPE numpy), intervals (INTERVAL_DTYPE numpy), origin
    (pos, strand int64 tensors), kind (numpy int8 per interval: 0 true, 1 repeat copy, 2 spurious)."""
    import torch

    from .hip import INTERVAL_DTYPE, READ_DTYPE

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    G = int(genome.numel())
    L = read_len
    ar = torch.arange(L, device=device)
    code = torch.zeros(256, dtype=torch.int64, device=device)
    for i, x in enumerate(b"ACGT"):
        code[x] = i
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    q_parts, q_lens, pos_all, strand_all = [], [], [], []
    for s0 in range(0, n_reads, slab):
        n = min(slab, n_reads - s0)
        pos = torch.randint(0, G - L, (n,), generator=gen, device=device)
        strand = torch.randint(0, 2, (n,), generator=gen, device=device)
        idx = torch.where(strand[:, None] == 1, pos[:, None] + (L - 1 - ar)[None, :], pos[:, None] + ar[None, :])
        raw = genome[idx]
        tmpl = code[raw.long()]
        tmpl = torch.where(strand[:, None] == 1, 3 - tmpl, tmpl)
        u = torch.rand((n, L), generator=gen, device=device)
        keep = u >= dele
        do_sub = keep & (u < dele + sub)
        shift = torch.randint(1, 4, (n, L), generator=gen, device=device)
        base = torch.where(do_sub, (tmpl + shift) & 3, tmpl)
        v = torch.rand((n, L), generator=gen, device=device).clamp_min(1e-12)
        k = torch.floor(torch.log(v) / np.log(ins)).to(torch.int64) if ins > 0 else torch.zeros((n, L), dtype=torch.int64, device=device)
        emit = keep.to(torch.int64) + k
        qpos = torch.cumsum(emit, dim=1) - emit
        rlen = emit.sum(dim=1)
        roff = torch.cumsum(rlen, 0) - rlen
        rd = torch.randint(0, 4, (int(rlen.sum()),), generator=gen, device=device)
        krow, kcol = torch.nonzero(keep, as_tuple=True)
        rd[roff[krow] + qpos[krow, kcol]] = base[krow, kcol]
        rd = acgt[rd]
        # a template base that is N (a contig separator) stays N in the read
        nrow, ncol = torch.nonzero((raw == ord("N")) & keep, as_tuple=True)
        rd[roff[nrow] + qpos[nrow, ncol]] = ord("N")
        q_parts.append(rd)
        q_lens.append(rlen)
        pos_all.append(pos)
        strand_all.append(strand)
    q_len = torch.cat(q_lens)
    pool = torch.cat(q_parts)
    pos, strand = torch.cat(pos_all), torch.cat(strand_all)
    n = n_reads
    rid = torch.arange(n, device=device)

    def rnd(hi, m):
        return torch.randint(0, hi, (m,), generator=gen, device=device)

    # ---- (1) the true locus, less the unanchored read ends
    scale = L / q_len.to(torch.float64)
    ua0, ua1 = rnd(max_unanchored + 1, n), rnd(max_unanchored + 1, n)
    lo_extra = rnd(int(loose * L) + 1, n) if loose > 0 else torch.zeros(n, dtype=torch.int64, device=device)
    hi_extra = rnd(int(loose * L) + 1, n) if loose > 0 else torch.zeros(n, dtype=torch.int64, device=device)
    iv_start = [(pos + (ua0 * scale).to(torch.int64) - lo_extra).clamp_min(0)]
    iv_end = [(pos + L - (ua1 * scale).to(torch.int64) + hi_extra).clamp_max(G)]
    iv_qs, iv_qe = [ua0], [q_len - ua1]
    iv_read, iv_strand, iv_kind = [rid], [strand], [torch.zeros(n, dtype=torch.int64, device=device)]
    # ---- (2) the other copies of the repeat family the read overlaps most
    if copies is not None and copies["start"].numel() > 0:
        cs, cf, cstr = copies["start"], copies["family"], copies["strand"]
        cu = copies["unit"][cf]
        # candidates: the copy starting at or before the read's start, and the next one
        j = torch.searchsorted(cs, pos, right=True) - 1
        best_ov = torch.zeros(n, dtype=torch.int64, device=device)
        best_c = torch.full((n,), -1, dtype=torch.int64, device=device)
        for cand in (j, j + 1):
            ok = (cand >= 0) & (cand < cs.numel())
            cc = cand.clamp(0, cs.numel() - 1)
            ov = (torch.minimum(pos + L, cs[cc] + cu[cc]) - torch.maximum(pos, cs[cc])).clamp_min(0)
            ov = torch.where(ok, ov, torch.zeros_like(ov))
            better = ov > best_ov
            best_ov = torch.where(better, ov, best_ov)
            best_c = torch.where(better, cc, best_c)
        has = best_ov >= min_overlap
        r_sel = rid[has]
        c_sel = best_c[has]
        # all copies of the same family: a (family -> copies) table, padded to the largest family
        order = torch.argsort(cf, stable=True)
        fam_sorted = cf[order]
        n_fam = int(copies["unit"].numel())
        fam_first = torch.searchsorted(fam_sorted, torch.arange(n_fam, device=device))
        fam_count = torch.bincount(cf, minlength=n_fam)
        kmax = int(fam_count.max())
        f_sel = cf[c_sel]
        for slot_i in range(kmax):
            m = (slot_i < fam_count[f_sel])
            d = order[(fam_first[f_sel] + slot_i).clamp_max(order.numel() - 1)]
            m = m & (d != c_sel)
            rr, c, dd = r_sel[m], c_sel[m], d[m]
            if rr.numel() == 0:
                continue
            U = cu[c]
            g0 = torch.maximum(pos[rr], cs[c])
            g1 = torch.minimum(pos[rr] + L, cs[c] + U)           # overlapped stretch [g0, g1) of the true locus
            # unit coordinates of the stretch (a copy on strand 1 holds the unit reverse-complemented)
            a = torch.where(cstr[c] == 0, g0 - cs[c], cs[c] + U - g1)
            b = torch.where(cstr[c] == 0, g1 - cs[c], cs[c] + U - g0)
            # the same stretch in copy dd
            h0 = torch.where(cstr[dd] == 0, cs[dd] + a, cs[dd] + U - b)
            h1 = torch.where(cstr[dd] == 0, cs[dd] + b, cs[dd] + U - a)
            rel = cstr[c] ^ cstr[dd]
            # read span of the stretch, in the orientation of the true locus (= the strand of the true interval)
            inv = q_len[rr].to(torch.float64) / L
            qa = ((g0 - pos[rr]) * inv).to(torch.int64)
            qb = ((g1 - pos[rr]) * inv).to(torch.int64).clamp_max(q_len[rr])
            t0, t1 = rnd(100, rr.numel()), rnd(100, rr.numel())
            qa2, qb2 = qa + t0, (qb - t1)
            keepm = qb2 - qa2 >= 200
            qs = torch.where(rel == 0, qa2, q_len[rr] - qb2)
            qe = torch.where(rel == 0, qb2, q_len[rr] - qa2)
            iv_start.append((h0 + torch.where(rel == 0, t0, t1))[keepm])
            iv_end.append((h1 - torch.where(rel == 0, t1, t0))[keepm])
            iv_qs.append(qs[keepm])
            iv_qe.append(qe[keepm])
            iv_read.append(rr[keepm])
            iv_strand.append((strand[rr] ^ rel)[keepm])
            iv_kind.append(torch.ones(int(keepm.sum()), dtype=torch.int64, device=device))
    # ---- (3) chance clusters: a short stretch of the read against unrelated sequence
    if spurious > 0:
        cnt = torch.poisson(torch.full((n,), float(spurious), device=device), generator=gen).to(torch.int64)
        rr = torch.repeat_interleave(rid, cnt)
        m = rr.numel()
        if spurious_span is None:
            span = 300 + rnd(1200, m)
            gspan = span
        else:
            q_lo, q_hi, g_lo, g_hi = spurious_span
            u = torch.rand(m, generator=gen, device=device)
            span = ((q_lo + (q_hi - q_lo) * u) * q_len[rr]).to(torch.int64).clamp_min(50)
            u = torch.rand(m, generator=gen, device=device)
            gspan = ((g_lo + (g_hi - g_lo) * u) * q_len[rr]).to(torch.int64).clamp_min(50)
        qs = (torch.rand(m, generator=gen, device=device) * (q_len[rr] - span).clamp_min(1)).to(torch.int64)
        gs = torch.randint(0, G - 4 * L, (m,), generator=gen, device=device) + L
        iv_start.append(gs)
        iv_end.append(gs + gspan)
        iv_qs.append(qs)
        iv_qe.append(torch.minimum(qs + span, q_len[rr]))
        iv_read.append(rr)
        iv_strand.append(rnd(2, m))
        iv_kind.append(torch.full((m,), 2, dtype=torch.int64, device=device))
    start, end = torch.cat(iv_start), torch.cat(iv_end)
    qs, qe = torch.cat(iv_qs), torch.cat(iv_qe)
    owner, st, kind = torch.cat(iv_read), torch.cat(iv_strand), torch.cat(iv_kind)
    # ---- at most max_candidates per read, the true locus and the repeat copies first; then a random order per read
    key = owner * 4 + kind
    o1 = torch.argsort(key, stable=True)
    owner_s = owner[o1]
    first = torch.searchsorted(owner_s, rid)
    rank = torch.arange(owner_s.numel(), device=device) - first[owner_s]
    o1 = o1[rank < max_candidates]
    shuffle = torch.rand(o1.numel(), generator=gen, device=device)
    o2 = torch.argsort(owner[o1].to(torch.float64) + shuffle.to(torch.float64) * 0.5, stable=True)
    sel = o1[o2]
    iv = np.zeros(int(sel.numel()), dtype=INTERVAL_DTYPE)
    iv["start"] = start[sel].clamp(0, G).cpu().numpy().astype(np.uint32)
    iv["end"] = end[sel].clamp(0, G).cpu().numpy().astype(np.uint32)
    iv["q_start"] = qs[sel].cpu().numpy().astype(np.uint32)
    iv["q_end"] = qe[sel].cpu().numpy().astype(np.uint32)
    iv["read"] = owner[sel].cpu().numpy().astype(np.int32)
    iv["strand"] = st[sel].cpu().numpy().astype(np.int32)
    reads = np.zeros(n, dtype=READ_DTYPE)
    ql = q_len.cpu().numpy()
    reads["off"] = np.cumsum(ql) - ql
    reads["length"] = ql
    reads["subread_start"] = 0
    reads["subread_end"] = ql
    return dict(pool=pool.contiguous(), reads=reads, intervals=iv, origin=(pos, strand), kind=kind[sel].cpu().numpy().astype(np.int8))
