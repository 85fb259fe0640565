"""next-4: the library's readers of the reference's on-disk inputs -- FASTA (records, the concatenated
upper-cased genome with its contig table, names) and the .sa file of sawriter -- against the reference's OWN readers
(FASTAReader / SequenceIndexDatabase / SuffixArray::Read through oracle/_ref/libblasr_refmapper.so) on files written
here, and against a small committed fixture where that build is absent."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from blasr_amd import hip, synth
from blasr_testlib import REFMAPPER_SO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAWRITER = os.path.join(ROOT, "oracle", "_ref", "sawritermc")

AWKWARD = (b"junk before the first record\n>contig one  with spaces\t and a tab\nACGTacgtNNrykm\nacgt ACGT\tGG\n\n  TT  \n"
           b">second\r\nAC\r\nGT\r\n>empty title follows\n>\nAAAA\n>last|no newline at the end\nCCCCGGGG")


def write(tmp_path, name, data):
    p = os.path.join(str(tmp_path), name)
    with open(p, "wb") as f:
        f.write(data)
    return p


def test_fasta_records_and_genome_of_an_awkward_file(tmp_path):
    p = write(tmp_path, "awkward.fasta", AWKWARD)
    recs = hip.read_fasta(p)
    assert [t for t, _ in recs] == [b"contig one  with spaces\t and a tab", b"second\r", b"empty title follows", b"",
                                    b"last|no newline at the end"]
    assert [bytes(s) for _, s in recs] == [b"ACGTacgtNNrykmacgtACGTGGTT", b"ACGT", b"", b"AAAA", b"CCCCGGGG"]
    assert hip.read_fasta(text=AWKWARD)[0][1].tobytes() == recs[0][1].tobytes()
    genome, starts, names = hip.read_genome(p)
    # everything that is not ACGT in either case becomes N (AllToUpper, NucConversion.h:230-263); one N after each contig
    assert bytes(genome) == b"ACGTACGTNNNNNNACGTACGTGGTTN" + b"ACGTN" + b"N" + b"AAAAN" + b"CCCCGGGGN"
    assert starts.tolist() == [0, 27, 32, 33, 38, 47]
    assert names == ["contig", "second", "empty", "", "last|no"]


@pytest.mark.skipif(not os.path.exists(REFMAPPER_SO), reason="oracle/_ref/libblasr_refmapper.so not built")
def test_fasta_against_the_reference_reader(tmp_path):
    ref = C.CDLL(REFMAPPER_SO)
    ref.ref_read_genome.restype = C.c_long
    rng = np.random.default_rng(5)
    # (without the record whose title is empty: the reference's seqdb builds a std::string from its NULL title)
    files = [AWKWARD.replace(b">\nAAAA\n", b"")]
    for it in range(6):  # random multi-FASTA files: line lengths, blank lines, lower case, IUPAC codes, several words
        out = b""
        for c in range(int(rng.integers(1, 6))):
            s = bytes(synth.random_genome(int(rng.integers(0, 900)), 100 * it + c))
            if it % 2:
                s = s.lower()
            s = bytearray(s)
            for _ in range(len(s) // 50):
                s[int(rng.integers(0, len(s)))] = int(rng.choice(list(b"NnRYKMswb")))
            out += b">c%d some description %d\n" % (c, it)
            w = int(rng.integers(10, 80))
            for k in range(0, len(s), w):
                out += bytes(s[k:k + w]) + (b" " if rng.integers(0, 5) == 0 else b"") + b"\n"
                if rng.integers(0, 9) == 0:
                    out += b"\n"
        files.append(out)
    for i, data in enumerate(files):
        p = write(tmp_path, "f%d.fasta" % i, data)
        genome, starts, names = hip.read_genome(p)
        seq = np.zeros(len(data) + 64, np.uint8)
        st = np.zeros(64, np.uint32)
        titles = C.create_string_buffer(len(data) + 64)
        n = C.c_int(0)
        L = ref.ref_read_genome(p.encode(), seq.ctypes.data_as(C.c_void_p), C.c_long(len(seq)), st.ctypes.data_as(C.c_void_p), 64,
                                titles, C.c_long(len(titles)), C.byref(n))
        assert L == len(genome), i
        assert np.array_equal(seq[:L], genome), i
        assert st[: n.value + 1].tolist() == starts.tolist(), i
        recs = hip.read_fasta(p)
        assert titles.value.split(b"\n")[:-1] == [t for t, _ in recs], i
        lengths = (C.c_long * 64)()
        k = ref.ref_read_fasta_records(p.encode(), seq.ctypes.data_as(C.c_void_p), C.c_long(len(seq)), lengths, 64)
        assert k == len(recs) and [lengths[j] for j in range(k)] == [len(s) for _, s in recs], i
        assert bytes(seq[: sum(lengths[j] for j in range(k))]) == b"".join(bytes(s) for _, s in recs), i


def _ref_fastq_in_child(path):
    """The reference's FASTQReader over `path`, in a forked child (it exit(0)s on a malformed record): the list of
    (title, bases, qualities), or None when the child left without an answer."""
    import pickle
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:  # child
        os.close(r)
        ref = C.CDLL(REFMAPPER_SO)
        seq, qual = np.zeros(1 << 20, np.uint8), np.zeros(1 << 20, np.uint8)
        lengths = (C.c_long * 256)()
        titles = C.create_string_buffer(1 << 16)
        k = ref.ref_read_fastq_records(path.encode(), seq.ctypes.data_as(C.c_void_p), qual.ctypes.data_as(C.c_void_p),
                                       C.c_long(len(seq)), lengths, 256, titles, C.c_long(len(titles)))
        recs, at = [], 0
        tl = titles.value.split(b"\n")
        for j in range(max(k, 0)):
            recs.append((tl[j], bytes(seq[at: at + lengths[j]]), bytes(qual[at: at + lengths[j]])))
            at += lengths[j]
        os.write(w, pickle.dumps((k, recs)))
        os._exit(7)
    os.close(w)
    blob = b""
    while True:
        piece = os.read(r, 1 << 16)
        if not piece:
            break
        blob += piece
    os.close(r)
    os.waitpid(pid, 0)
    return pickle.loads(blob) if blob else None


@pytest.mark.skipif(not os.path.exists(REFMAPPER_SO), reason="oracle/_ref/libblasr_refmapper.so not built")
def test_fastq_against_the_reference_reader(tmp_path):
    """FASTQReader::GetNext(FASTQSequence&) of this fork is not a FASTQ parser (after a title it looks for the NEXT '@':
    a four-line file yields every second record, an '@' among the qualities starts a title there): the library
    restates what it does, pinned here against the reference's own reader on plain, multi-line, awkward and random
    files; where the reference exits (no '+' line) the library returns ERR_INVALID."""
    rng = np.random.default_rng(21)

    def rec(name, n, qual_chars=b"!#5<EIJ", lines=1):
        s = bytes(rng.choice(list(b"ACGTN"), n).astype(np.uint8))
        q = bytes(rng.choice(list(qual_chars), n).astype(np.uint8))
        body = b"\n".join(s[i::lines] if False else s[i * (n // lines + 1):(i + 1) * (n // lines + 1)] for i in range(lines))
        return b"@" + name + b"\n" + body.strip(b"\n") + b"\n+\n" + q + b"\n", s, q

    files = {}
    plain = [rec(b"read%d/%d/0_%d" % (i, i, 40 + i), 40 + i) for i in range(6)]
    files["plain"] = b"".join(r[0] for r in plain)
    files["one_record"] = plain[0][0]
    files["two_records"] = plain[0][0] + plain[1][0]
    files["multi_line"] = b"".join(rec(b"m%d desc with spaces" % i, 90, lines=3)[0] for i in range(4))
    files["at_in_quality"] = b"".join(rec(b"q%d" % i, 50, qual_chars=b"@AB5")[0] for i in range(5))
    files["junk_first"] = b"# a comment\n\n" + files["plain"]
    files["no_final_newline"] = files["plain"].rstrip(b"\n")
    files["crlf_titles"] = files["plain"].replace(b"\n+\n", b"\r\n+\n")
    files["plus_with_name"] = b"".join(r[0].replace(b"\n+\n", b"\n+again\n") for r in plain)
    files["long_quality"] = b"@a\nACGT\n+\nIIII\n@b\nACGTAC\n+\nIIIIIIJJJ\n@c\nAC\n+\nII\n@d\nGG\n+\nJJ\n"
    files["empty"] = b""
    files["only_title"] = b"@x\n"
    for k in range(8):  # random mixtures of well-formed records, some with '@' among the qualities
        files["random%d" % k] = b"".join(rec(b"r%d_%d" % (k, i), int(rng.integers(1, 120)),
                                             qual_chars=b"@!5I" if rng.integers(0, 2) else b"!5I",
                                             lines=int(rng.integers(1, 4)))[0] for i in range(int(rng.integers(1, 9))))
    checked = 0
    for name, blob in files.items():
        path = write(tmp_path, name + ".fastq", blob)
        want = _ref_fastq_in_child(path)
        try:
            got = hip.read_fastq(path)
        except hip.BlasrHipError as e:
            # the reference exited (no answer from the child), or went on with uninitialised quality bytes
            assert want is None or e.code == hip.ERR_UNSUPPORTED, (name, e)
            continue
        assert want is not None, name
        k, recs = want
        assert k == len(got), (name, k, len(got))
        for (t, s_, q), (wt, ws, wq) in zip(got, recs):
            assert (t, bytes(s_), bytes(q)) == (wt, ws, wq), name
        checked += len(got)
    assert checked >= 12
    # what the behaviour means on a plain file: records 2, 4, 6 of six
    got = hip.read_fastq(text=files["plain"])
    assert [t for t, _, _ in got] == [b"read1/1/0_41", b"read3/3/0_43", b"read5/5/0_45"]
    assert [bytes(s_) for _, s_, _ in got] == [plain[1][1], plain[3][1], plain[5][1]]
    assert [bytes(q) for _, _, q in got] == [plain[1][2], plain[3][2], plain[5][2]]
    with pytest.raises(hip.BlasrHipError) as e:  # a record whose '+' line is missing: the reference exits
        hip.read_fastq(text=b"@a\nAC\n+\nII\n@b\nACGT\n")
    assert e.value.code == hip.ERR_INVALID


@pytest.mark.skipif(not (os.path.exists(SAWRITER) and os.path.exists(REFMAPPER_SO)), reason="reference tools not built")
def test_suffix_array_file_of_sawriter(tmp_path):
    """A .sa written by the reference's sawriter for a two-contig genome: the library reads the same index array and
    lookup table as SuffixArray::Read; the index is a permutation that sorts the suffixes it lists."""
    g = b">a\n" + bytes(synth.random_genome(30000, 3)) + b"\n>b\n" + bytes(synth.random_genome(12000, 4)) + b"\n"
    fa = write(tmp_path, "g.fasta", g)
    sa_path = os.path.join(str(tmp_path), "g.fasta.sa")
    subprocess.run([SAWRITER, sa_path, fa], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    sa = hip.read_suffix_array(sa_path)
    ref = C.CDLL(REFMAPPER_SO)
    n = len(sa["index"])
    m = len(sa["start_pos_table"]) + 8
    idx, sp, ep = np.zeros(n + 8, np.uint32), np.zeros(m, np.uint32), np.zeros(m, np.uint32)
    out = (C.c_long * 3)()
    rc = ref.ref_read_sa(sa_path.encode(), idx.ctypes.data_as(C.c_void_p), C.c_long(len(idx)), sp.ctypes.data_as(C.c_void_p),
                         ep.ctypes.data_as(C.c_void_p), C.c_long(len(sp)), out)
    assert rc == 0 and out[0] == n and out[2] == sa["lookup_prefix_length"]
    assert np.array_equal(idx[:n], sa["index"])
    assert np.array_equal(sp[: out[1]], sa["start_pos_table"]) and np.array_equal(ep[: out[1]], sa["end_pos_table"])
    genome, starts, _ = hip.read_genome(fa)
    assert n <= len(genome) and len(set(sa["index"][:2000].tolist())) == 2000
    a, b = int(sa["index"][100]), int(sa["index"][101])
    assert bytes(genome[a: a + 40]) <= bytes(genome[b: b + 40])  # neighbours in the array are in suffix order


def test_truncated_or_corrupt_suffix_array_files_are_refused(tmp_path):
    """Header fields are checked against what the file holds before anything is sized from them: a truncated file, an
    index length beyond the file and a lookup table that is not 4^prefix entries come back ERR_INVALID (no allocation
    of what the header claims, no exception across the C boundary)."""
    import struct
    idx = np.arange(100, dtype=np.uint32)
    table = np.zeros(16, dtype=np.uint32)
    good = struct.pack("<Iii", 0xacac0001, 1, 1) + struct.pack("<I", 100) + idx.tobytes() + struct.pack("<II", 16, 2) + \
        table.tobytes() + table.tobytes()
    cases = {"good": good, "cut_in_index": good[:200], "cut_in_table": good[:-20],
             "huge_index": good[:12] + struct.pack("<I", 0xfffffff0) + good[16:],
             "huge_table": good[:416] + struct.pack("<II", 0xfffffff0, 2) + good[424:],
             "table_not_4_to_prefix": good[:416] + struct.pack("<II", 16, 3) + good[424:]}
    for name, blob in cases.items():
        path = os.path.join(str(tmp_path), name + ".sa")
        open(path, "wb").write(blob)
        if name == "good":
            sa = hip.read_suffix_array(path)
            assert np.array_equal(sa["index"], idx) and sa["lookup_prefix_length"] == 2
        else:
            with pytest.raises(hip.BlasrHipError) as e:
                hip.read_suffix_array(path)
            assert e.value.code == hip.ERR_INVALID, name


@pytest.mark.skipif(not os.path.exists(REFMAPPER_SO), reason="oracle/_ref/libblasr_refmapper.so not built")
def test_count_table_file_against_the_reference(tmp_path):
    """.ctab: the library's count table of a genome written by the library is what the reference's TupleCountTable
    reads, and what the reference writes the library reads; the two files are byte for byte the same."""
    ref = C.CDLL(REFMAPPER_SO)
    ref.ref_ct_open.restype = C.c_void_p
    ref.ref_ct_read.restype = C.c_void_p
    ref.ref_ct_table.restype = C.c_long
    g = np.concatenate([synth.random_genome(40000, 8), np.frombuffer(b"NNNN", np.uint8), synth.random_genome(9000, 9),
                        np.frombuffer(b"N", np.uint8)])
    for k in (5, 8):
        mine = hip.count_tuples(g, k)
        p_mine, p_ref = os.path.join(str(tmp_path), "mine%d.ctab" % k), os.path.join(str(tmp_path), "ref%d.ctab" % k)
        hip.write_count_table(p_mine, mine)
        h = ref.ref_ct_open(g.ctypes.data_as(C.c_void_p), len(g), k)
        assert ref.ref_ct_write(C.c_void_p(h), p_ref.encode()) == 0
        assert open(p_mine, "rb").read() == open(p_ref, "rb").read()
        back, kk = hip.read_count_table(p_ref)
        assert kk == k and np.array_equal(back, mine)
        h2 = ref.ref_ct_read(p_mine.encode())
        tab = np.zeros(len(mine), np.int32)
        assert ref.ref_ct_table(C.c_void_p(h2), tab.ctypes.data_as(C.c_void_p), C.c_long(len(tab))) == len(mine)
        assert np.array_equal(tab, mine)
        ref.ref_ct_close(C.c_void_p(h))
        ref.ref_ct_close(C.c_void_p(h2))
    with pytest.raises(hip.BlasrHipError):
        hip.read_count_table(os.path.join(str(tmp_path), "missing.ctab"))


def _ref_sam_in_child(path):
    """The reference's ReaderAgglomerate over a `.sam` read file, in a forked child (it exit(1)s on a line it cannot
    parse): (count, [(title, bases, quality characters or b"")]) or None when the child left without an answer."""
    import pickle
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:  # child
        os.close(r)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        ref = C.CDLL(REFMAPPER_SO)
        seq, qual = np.zeros(1 << 20, np.uint8), np.zeros(1 << 20, np.uint8)
        lengths = (C.c_long * 256)()
        has_qual = (C.c_int * 256)()
        titles = C.create_string_buffer(1 << 16)
        k = ref.ref_read_sam_reads(path.encode(), seq.ctypes.data_as(C.c_void_p), qual.ctypes.data_as(C.c_void_p),
                                   C.c_long(len(seq)), lengths, has_qual, 256, titles, C.c_long(len(titles)))
        recs, at = [], 0
        tl = titles.value.split(b"\n")
        for j in range(max(k, 0)):
            q = bytes((qual[at: at + lengths[j]] + 33).astype(np.uint8)) if has_qual[j] else b""  # stored = character - 33
            recs.append((tl[j], bytes(seq[at: at + lengths[j]]), q))
            at += lengths[j]
        os.write(w, pickle.dumps((k, recs)))
        os._exit(7)
    os.close(w)
    blob = b""
    while True:
        piece = os.read(r, 1 << 16)
        if not piece:
            break
        blob += piece
    os.close(r)
    os.waitpid(pid, 0)
    return pickle.loads(blob) if blob else None


@pytest.mark.skipif(not os.path.exists(REFMAPPER_SO), reason="reference tools not built")
def test_sam_read_files_against_the_reference_reader(tmp_path):
    """A `.sam` file as READ input (files/ReaderAgglomerate.h:308-318): header lines skipped, every later line's first
    eleven words parsed, title / bases / qualities as ConvertSAMToSMRTSequence leaves them -- against the reference's
    own reader on plain, tagged, blank-separated and random files; where the reference exits (a short or empty line, a
    word that is no number) the library returns ERR_INVALID."""
    rng = np.random.default_rng(5)

    def rec(name, n, qual=True, tags=b"", sep=b"\t", flag=b"4"):
        s = bytes(rng.choice(list(b"ACGTNacgt"), n).astype(np.uint8))
        q = bytes(rng.choice(list(b"!#5<EIJ"), n).astype(np.uint8)) if qual else b"*"  # (not '~': printed 93 is stored as another value, SAMQVConversion.h:30)
        return sep.join([name, flag, b"*", b"0", b"0", b"*", b"*", b"0", b"0", s, q]) + (sep + tags if tags else b"") + b"\n"

    head = b"@HD\tVN:1.4\tSO:unsorted\n@SQ\tSN:c1\tLN:1000\n@RG\tID:abc\tPU:movie1\n@PG\tID:x\n@CO\tany text\n"
    files = {
        "plain": head + b"".join(rec(b"m/%d/0_%d" % (i, 30 + i), 30 + i) for i in range(5)),
        "no_header": b"".join(rec(b"r%d" % i, 20 + i) for i in range(3)),
        "no_quality": head + rec(b"a", 25, qual=False) + rec(b"b", 26),
        "tags": head + rec(b"m/7/0_40", 40, tags=b"RG:Z:abc\tzm:i:7\tqs:i:0\tqe:i:40\trq:f:0.85\tnp:i:1\tsn:B:f,4.5,5.5,6.5,7.5"),
        "blanks": rec(b"x", 22, sep=b" ") + rec(b"y", 23, sep=b"  \t "),
        "aligned_record": b"q1\t16\tc1\t101\t254\t10M\t=\t0\t-5\tACGTACGTAC\tIIIIIIIIII\tAS:i:-50\n",
        "no_final_newline": (head + rec(b"a", 30) + rec(b"b", 31)).rstrip(b"\n"),
        "empty": b"",
        "only_header": head,
        "blank_line_inside": rec(b"a", 20) + b"\n" + rec(b"b", 21),       # the reference exits on the empty line
        "short_line": rec(b"a", 20) + b"b\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\n",  # ten words
        "bad_number": rec(b"a", 20) + b"b\t4\t*\tzero\t0\t*\t*\t0\t0\tACGT\tIIII\n",
        "header_after_records": rec(b"a", 20) + b"@CO\tlate\n" + rec(b"b", 21),
    }
    for k in range(6):
        files["random%d" % k] = (head if rng.integers(0, 2) else b"") + b"".join(
            rec(b"r%d_%d" % (k, i), int(rng.integers(1, 200)), qual=bool(rng.integers(0, 3)),
                sep=b"\t" if rng.integers(0, 3) else b" ") for i in range(int(rng.integers(1, 9))))
    checked = 0
    for name, blob in files.items():
        path = write(tmp_path, name + ".sam", blob)
        want = _ref_sam_in_child(path)
        try:
            got = hip.read_sam_reads(path)
        except hip.BlasrHipError as e:
            assert want is None and e.code == hip.ERR_INVALID, (name, e, want)
            continue
        assert want is not None, name
        k, recs = want
        assert k == len(got), (name, k, len(got))
        for (t, s_, q), (wt, ws, wq) in zip(got, recs):
            assert (t, bytes(s_), bytes(q)) == (wt, ws, wq), name
        checked += len(got)
    assert checked >= 25
    assert hip.read_sam_reads(text=files["plain"])[0][0] == b"m/0/0_30"
