// formats_api.hip -- next-4: the reference's on-disk inputs, read by the library's host
// side so that a C++ caller needs nothing of the reference to feed the path.  Host code only (no device involved).
//
//   blasr_hip_fasta_read        FASTAReader::GetNext / ReadAllSequences (FASTAReader.h:166-216, 255-271): a record is a
//                               title line (everything after '>', spaces included) and the whitespace-separated
//                               words of the following lines up to the next line that starts with '>'
//   blasr_hip_fasta_concatenate FASTAReader::ReadAllSequencesIntoOne (:113-146) + the seqdb it fills
//                               (SequenceIndexDatabase: seqStartPos, names) + genome.ToUpper() (Blasr.cpp:4354): the
//                               contigs back to back, one 'N' after each
//   blasr_hip_fastq_read        FASTQReader::GetNext(FASTQSequence&) (FASTQReader.h:27-66) AS THIS FORK HAS IT, see parse_fastq
//   blasr_hip_sa_read           SuffixArray::Read (datastructures/suffixarray/SuffixArray.h:608-630; layout written by
//                               Write :505-532): magic 0xacac0001, the component list, the index array, the lookup table
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <new>
#include <vector>

#include "../../include/blasr_hip.h"
#include "abi_guard.h"

struct blasr_hip_fasta {
  std::vector<std::string> titles;
  std::vector<std::vector<uint8_t>> seqs;
  std::vector<std::vector<uint8_t>> quals;  // FASTQ only: one quality character per base (empty for a FASTA file)
  std::string error;
};

struct blasr_hip_sa {
  std::vector<uint32_t> index, startPos, endPos;
  uint32_t lookupTableLength = 0, lookupPrefixLength = 0;
  int hasArray = 0, hasLookup = 0;
};

namespace {

bool read_file(const char *path, std::vector<char> &buf) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize(n > 0 ? (size_t)n : 0);
  const size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
  fclose(f);
  return got == buf.size();
}

// Records of a FASTA text as the reference's reader sees them (delim '>').  Lines end at '\n' (getline); words are
// separated by the whitespace `operator>>` skips (space, \t, \r, \v, \f); characters before the first line that
// starts with the delimiter are skipped (AdvanceToTitleStart), and the title of a record is the rest of its first
// line after the delimiter, a trailing '\r' and inner spaces included.
void parse_fasta(const char *p, size_t n, blasr_hip_fasta &out) {
  size_t at = 0;
  while (true) {
    while (at < n && p[at] != '>') at++;  // AdvanceToTitleStart looks at every character, not only line starts
    if (at >= n) break;
    size_t e = at;
    while (e < n && p[e] != '\n') e++;
    out.titles.push_back(std::string(p + at + 1, e - at - 1));
    out.seqs.push_back(std::vector<uint8_t>());
    std::vector<uint8_t> &s = out.seqs.back();
    at = e < n ? e + 1 : n;
    while (at < n && p[at] != '>') {  // a line that starts with the delimiter ends the record (file.peek())
      size_t le = at;
      while (le < n && p[le] != '\n') le++;
      for (size_t i = at; i < le; i++) {
        const char c = p[i];
        if (c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f') continue;
        s.push_back((uint8_t)c);
      }
      at = le < n ? le + 1 : n;
    }
  }
}

// FASTQReader::GetNext(FASTQSequence&) of this fork (FASTQReader.h:27-66), character for character.  It is NOT a
// FASTQ parser: after reading a title line ('@' ...) it calls FASTAReader::GetNext(seq, '@') (:41), which looks for the
// NEXT '@' from where it stands -- through the record's own bases, its '+' line and its qualities -- reads THAT line as
// the title and the lines after it, up to one that starts with '+', as the bases (:166-216 with delim '@' and
// endOfReadDelim '+').  Then one line that must start with '+' (:49-53, else the reference prints an error and
// exit(0)s), then as many quality characters as there are bases (:54-60: ifstream::get(buf, length + 1) stops at a
// newline; one more get() swallows it).  On a plain four-line file whose quality strings hold no '@' every SECOND record
// comes out (titles, bases and qualities of records 2, 4, 6, ...); an '@' among the qualities starts a "title" in the
// middle of that line.  The binary's behaviour on FASTQ input is this, so this is what is restated.  Where the
// reference leaves defined ground the restatement stops with an error instead: a missing '+' line (the reference
// exits), fewer quality characters than bases (it copies uninitialised stack), an empty quality line (its stream
// fails and the next call never returns).
// The stream is an in-memory std::ifstream: eof is set by a peek() or get() at the end, or by a getline() that runs
// into the end before a newline.
struct TextStream {
  const char *p;
  size_t n, at = 0;
  bool eof = false;
  int peek() {
    if (at >= n) {
      eof = true;
      return -1;
    }
    return (unsigned char)p[at];
  }
  int get() {
    if (at >= n) {
      eof = true;
      return -1;
    }
    return (unsigned char)p[at++];
  }
  std::string getline() {  // (at the end already: eof + fail, an empty line)
    std::string line;
    if (at >= n) {
      eof = true;
      return line;
    }
    size_t e = at;
    while (e < n && p[e] != '\n') e++;
    line.assign(p + at, e - at);
    if (e < n) at = e + 1;
    else {
      at = n;
      eof = true;
    }
    return line;
  }
};

int parse_fastq(const char *p, size_t n, blasr_hip_fasta &out) {
  TextStream f{p, n};
  auto advance_to_title = [&]() {  // FASTAReader::AdvanceToTitleStart('@') (:90-101)
    while (!f.eof && f.peek() != '@') f.get();
    return !f.eof;
  };
  for (;;) {
    if (f.eof) break;
    if (!advance_to_title()) break;
    (void)f.getline();  // ReadTitle (:41 of FASTQReader.h): overwritten by the one FASTAReader::GetNext reads below
    // ---- FASTAReader::GetNext(seq, '@')
    if (f.eof) break;
    if (!advance_to_title()) break;
    const std::string title = f.getline();
    std::vector<uint8_t> seq;
    while (!f.eof && f.peek() != '+') {  // (file.good() && !eof: nothing here sets failbit before eof)
      const std::string line = f.getline();
      for (char c : line) {
        if (c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n') continue;
        seq.push_back((uint8_t)c);
      }
    }
    // ---- back in FASTQReader::GetNext: the '+' line
    const std::string plus = f.getline();
    if (plus.empty() || plus[0] != '+') {
      out.error = "ERROR in fastq format. Expecting '+' (the reference exits here)";
      return BLASR_HIP_ERR_INVALID;
    }
    std::vector<uint8_t> qual;
    if (!seq.empty()) {
      // file.get(buf, length + 1): up to `length` characters, stopping BEFORE a newline
      while (qual.size() < seq.size() && f.at < f.n && f.p[f.at] != '\n') qual.push_back((uint8_t)f.p[f.at++]);
      if (qual.size() < seq.size()) {
        if (f.at >= f.n) f.eof = true;
        out.error = qual.empty() ? "empty quality line (the reference's stream fails and its next call never returns)"
                                 : "fewer quality values than bases (the reference copies uninitialised memory)";
        return BLASR_HIP_ERR_UNSUPPORTED;
      }
      (void)f.get();  // "get the \\n" -- whatever the next character is
    }
    out.titles.push_back(title.size() > 1 ? title.substr(1) : std::string());  // ReadTitle: the line without its '@'
    out.seqs.push_back(std::move(seq));
    out.quals.push_back(std::move(qual));
  }
  return BLASR_HIP_OK;
}

// Reads from a SAM file, as ReaderAgglomerate::GetNext(SMRTSequence&) takes them (files/ReaderAgglomerate.h:308-318):
// SAMReader::ReadHeader (algorithms/alignment/readers/sam/SAMReader.h:119-155) consumes the leading lines that start
// with '@'; after that EVERY line is a record (SAMReader::GetNextAlignment :180-195: getline + StoreValues): its first
// eleven whitespace-separated words (`operator>>`: tabs or blanks alike) are QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT
// TLEN SEQ QUAL (datastructures/alignmentset/SAMAlignment.h:142-192), and ConvertSAMToSMRTSequence
// (SAMToSMRTSequence.h:53-98) makes the read of QNAME (title), SEQ (bases as they stand) and QUAL (none when "*"); the
// subread is the whole read.  A line with fewer than eleven words, an empty one included, or a number that does not
// parse makes the reference print "Error parsing alignment line" and exit(1): BLASR_HIP_ERR_INVALID here.  The optional
// tags (zm, qs, qe, rq, np, sn and the rich quality values) do not reach the mapper's alignments and are not kept.
int parse_sam_reads(const char *p, size_t n, blasr_hip_fasta &out) {
  size_t at = 0;
  while (at < n && p[at] == '@') {  // PeekLineIsHeader
    while (at < n && p[at] != '\n') at++;
    if (at < n) at++;
  }
  auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; };
  auto is_uint = [](const std::string &w) {
    if (w.empty() || w.size() > 9) return false;
    for (char c : w)
      if (c < '0' || c > '9') return false;
    return true;
  };
  auto is_int = [&](const std::string &w) {
    if (w.empty()) return false;
    const size_t from = (w[0] == '-' || w[0] == '+') ? 1 : 0;
    return is_uint(w.substr(from));
  };
  int64_t line = 0;
  while (at < n) {
    size_t e = at;
    while (e < n && p[e] != '\n') e++;
    std::vector<std::string> words;
    size_t i = at;
    while (i < e && words.size() < 11) {
      while (i < e && is_space(p[i])) i++;
      size_t j = i;
      while (j < e && !is_space(p[j])) j++;
      if (j > i) words.push_back(std::string(p + i, j - i));
      i = j;
    }
    if (words.size() < 11 || !is_uint(words[1]) || !is_uint(words[3]) || !is_int(words[4]) || !is_int(words[7]) || !is_int(words[8])) {
      out.error = "Error parsing alignment line " + std::to_string(line) + " (the reference exits here)";
      return BLASR_HIP_ERR_INVALID;
    }
    out.titles.push_back(words[0]);
    out.seqs.push_back(std::vector<uint8_t>(words[9].begin(), words[9].end()));
    out.quals.push_back(words[10] == "*" ? std::vector<uint8_t>() : std::vector<uint8_t>(words[10].begin(), words[10].end()));
    at = e < n ? e + 1 : n;
    line++;
  }
  return BLASR_HIP_OK;
}

}  // namespace

extern "C" {

int blasr_hip_sam_reads_parse(const char *text, int64_t length, blasr_hip_fasta **out) try {
  if ((!text && length > 0) || length < 0 || !out) return BLASR_HIP_ERR_INVALID;
  blasr_hip_fasta *f = new blasr_hip_fasta();
  const int rc = parse_sam_reads(text, (size_t)length, *f);
  *out = f;  // (also on an error: the records before it and the message, blasr_hip_fasta_error)
  return rc;
} BLASR_HIP_CATCH_CODE

int blasr_hip_sam_reads_read(const char *path, blasr_hip_fasta **out) try {
  if (!path || !out) return BLASR_HIP_ERR_INVALID;
  *out = nullptr;
  std::vector<char> buf;
  if (!read_file(path, buf)) return BLASR_HIP_ERR_INVALID;
  return blasr_hip_sam_reads_parse(buf.data(), (int64_t)buf.size(), out);
} BLASR_HIP_CATCH_CODE

int blasr_hip_fastq_read(const char *path, blasr_hip_fasta **out) try {
  if (!path || !out) return BLASR_HIP_ERR_INVALID;
  *out = nullptr;
  std::vector<char> buf;
  if (!read_file(path, buf)) return BLASR_HIP_ERR_INVALID;
  return blasr_hip_fastq_parse(buf.data(), (int64_t)buf.size(), out);
} BLASR_HIP_CATCH_CODE

int blasr_hip_fastq_parse(const char *text, int64_t length, blasr_hip_fasta **out) try {
  if ((!text && length > 0) || length < 0 || !out) return BLASR_HIP_ERR_INVALID;
  blasr_hip_fasta *f = new blasr_hip_fasta();
  const int rc = parse_fastq(text, (size_t)length, *f);
  *out = f;  // (also on an error: the records before it and the message, blasr_hip_fasta_error)
  return rc;
} BLASR_HIP_CATCH_CODE

int blasr_hip_fastq_quality(const blasr_hip_fasta *f, int64_t i, const uint8_t **qual, int64_t *qual_length) try {
  if (!f || i < 0 || i >= (int64_t)f->quals.size()) return BLASR_HIP_ERR_INVALID;
  if (qual) *qual = f->quals[(size_t)i].data();
  if (qual_length) *qual_length = (int64_t)f->quals[(size_t)i].size();
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

const char *blasr_hip_fasta_error(const blasr_hip_fasta *f) try { return f ? f->error.c_str() : ""; } BLASR_HIP_CATCH_NULL

int blasr_hip_fasta_read(const char *path, blasr_hip_fasta **out) try {
  if (!path || !out) return BLASR_HIP_ERR_INVALID;
  *out = nullptr;
  std::vector<char> buf;
  if (!read_file(path, buf)) return BLASR_HIP_ERR_INVALID;
  blasr_hip_fasta *f = new blasr_hip_fasta();
  parse_fasta(buf.data(), buf.size(), *f);
  *out = f;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_fasta_parse(const char *text, int64_t length, blasr_hip_fasta **out) try {
  if ((!text && length > 0) || length < 0 || !out) return BLASR_HIP_ERR_INVALID;
  blasr_hip_fasta *f = new blasr_hip_fasta();
  parse_fasta(text, (size_t)length, *f);
  *out = f;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_fasta_free(blasr_hip_fasta *f) try {
  delete f;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int64_t blasr_hip_fasta_count(const blasr_hip_fasta *f) try { return f ? (int64_t)f->seqs.size() : 0; } BLASR_HIP_CATCH_CODE

int blasr_hip_fasta_record(const blasr_hip_fasta *f, int64_t i, const char **title, int64_t *title_length,
                           const uint8_t **seq, int64_t *seq_length) try {
  if (!f || i < 0 || i >= (int64_t)f->seqs.size()) return BLASR_HIP_ERR_INVALID;
  if (title) *title = f->titles[(size_t)i].c_str();
  if (title_length) *title_length = (int64_t)f->titles[(size_t)i].size();
  if (seq) *seq = f->seqs[(size_t)i].data();
  if (seq_length) *seq_length = (int64_t)f->seqs[(size_t)i].size();
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int64_t blasr_hip_fasta_name(const blasr_hip_fasta *f, int64_t i, char *name, int64_t capacity) try {
  // FASTASequence::GetName (FASTASequence.h:36-52): the title up to its first blank
  if (!f || i < 0 || i >= (int64_t)f->titles.size()) return BLASR_HIP_ERR_INVALID;
  const std::string &t = f->titles[(size_t)i];
  size_t n = 0;
  while (n < t.size() && t[n] != ' ' && t[n] != '\t' && t[n] != '\n' && t[n] != '\r') n++;
  if (name && capacity > 0) {
    const size_t c = std::min(n, (size_t)capacity - 1);
    memcpy(name, t.data(), c);
    name[c] = 0;
  }
  return (int64_t)n;
} BLASR_HIP_CATCH_CODE

int64_t blasr_hip_fasta_concatenate(const blasr_hip_fasta *f, int to_upper, uint8_t *genome, int64_t capacity,
                                    uint32_t *seq_start_pos) try {
  if (!f) return BLASR_HIP_ERR_INVALID;
  int64_t total = 0;
  for (auto &s : f->seqs) total += (int64_t)s.size() + 1;
  if (total >= (1ll << 32)) return BLASR_HIP_ERR_UNSUPPORTED;  // "Reading fasta files greater than 4Gbytes is not supported."
  if (seq_start_pos) {
    uint32_t at = 0;
    seq_start_pos[0] = 0;
    for (size_t i = 0; i < f->seqs.size(); i++) {
      at += (uint32_t)f->seqs[i].size() + 1;
      seq_start_pos[i + 1] = at;
    }
  }
  if (!genome) return total;
  if (capacity < total) return BLASR_HIP_ERR_CAPACITY;
  int64_t at = 0;
  for (auto &s : f->seqs) {
    for (size_t i = 0; i < s.size(); i++) {
      uint8_t c = s[i];
      if (to_upper) {  // DNASequence::ToUpper through the AllToUpper table (NucConversion.h:230-263): codes 0-3 and
        uint8_t u = 'N';  // ACGT in either case become the letter, EVERYTHING else (IUPAC codes too) becomes N
        if (c == 0 || c == 'A' || c == 'a') u = 'A';
        if (c == 1 || c == 'C' || c == 'c') u = 'C';
        if (c == 2 || c == 'G' || c == 'g') u = 'G';
        if (c == 3 || c == 'T' || c == 't') u = 'T';
        c = u;
      }
      genome[at + (int64_t)i] = c;
    }
    at += (int64_t)s.size();
    genome[at++] = 'N';
  }
  return total;
} BLASR_HIP_CATCH_CODE

int blasr_hip_sa_read(const char *path, blasr_hip_sa **out) try {
  if (!path || !out) return BLASR_HIP_ERR_INVALID;
  *out = nullptr;
  std::vector<char> buf;
  try {
    if (!read_file(path, buf)) return BLASR_HIP_ERR_INVALID;
  } catch (const std::bad_alloc &) {
    return BLASR_HIP_ERR_NOMEM;
  }
  size_t at = 0;
  auto take = [&](void *dst, size_t bytes) {
    if (at + bytes > buf.size()) return false;
    memcpy(dst, buf.data() + at, bytes);
    at += bytes;
    return true;
  };
  uint32_t magic = 0;
  int32_t comp[2] = {0, 0};
  if (!take(&magic, 4) || magic != 0xacac0001u) return BLASR_HIP_ERR_UNSUPPORTED;  // (files without the magic number: not read)
  if (!take(comp, 8)) return BLASR_HIP_ERR_INVALID;
  // sizes come from the header: check them against what the file holds BEFORE anything is allocated (a truncated or
  // corrupt file would otherwise ask for up to 16 GB), and keep std::bad_alloc inside the C boundary
  blasr_hip_sa *sa = nullptr;
  bool ok = true;
  try {
    sa = new blasr_hip_sa();
    sa->hasArray = comp[0];
    sa->hasLookup = comp[1];
    if (comp[0]) {
      uint32_t len = 0;
      ok = take(&len, 4) && at + 4ull * len <= buf.size();
      if (ok) {
        sa->index.resize(len);
        ok = take(sa->index.data(), 4ull * len);
      }
    }
    if (ok && comp[1]) {
      ok = take(&sa->lookupTableLength, 4) && take(&sa->lookupPrefixLength, 4);
      // the table has one range per word of lookupPrefixLength bases (SuffixArray.h:215-222)
      ok = ok && sa->lookupPrefixLength <= 15 && (uint64_t)sa->lookupTableLength == (1ull << (2 * sa->lookupPrefixLength)) &&
           at + 8ull * sa->lookupTableLength <= buf.size();
      if (ok) {
        sa->startPos.resize(sa->lookupTableLength);
        sa->endPos.resize(sa->lookupTableLength);
        ok = take(sa->startPos.data(), 4ull * sa->lookupTableLength) && take(sa->endPos.data(), 4ull * sa->lookupTableLength);
      }
    }
  } catch (const std::bad_alloc &) {
    delete sa;
    return BLASR_HIP_ERR_NOMEM;
  }
  if (!ok) {
    delete sa;
    return BLASR_HIP_ERR_INVALID;
  }
  *out = sa;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_sa_free(blasr_hip_sa *sa) try {
  delete sa;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

int blasr_hip_sa_arrays(const blasr_hip_sa *sa, const uint32_t **index, int64_t *length, const uint32_t **start_pos_table,
                        const uint32_t **end_pos_table, int64_t *lookup_table_length, int32_t *lookup_prefix_length) try {
  if (!sa) return BLASR_HIP_ERR_INVALID;
  if (index) *index = sa->index.data();
  if (length) *length = (int64_t)sa->index.size();
  if (start_pos_table) *start_pos_table = sa->startPos.data();
  if (end_pos_table) *end_pos_table = sa->endPos.data();
  if (lookup_table_length) *lookup_table_length = sa->lookupTableLength;
  if (lookup_prefix_length) *lookup_prefix_length = (int32_t)sa->lookupPrefixLength;
  return BLASR_HIP_OK;
} BLASR_HIP_CATCH_CODE

// TupleCountTable::Write / Read (datastructures/tuplelists/TupleCountTable.h:75-90): the .ctab file Blasr looks for
// next to the genome (Blasr.cpp:4463-4477) -- countTableLength, nTuples, tupleSize, the counts (all int32).
int blasr_hip_ctab_write(const char *path, const int32_t *count_table, int32_t tuple_size) try {
  if (!path || !count_table || tuple_size < 1 || tuple_size > 15) return BLASR_HIP_ERR_INVALID;
  FILE *f = fopen(path, "wb");
  if (!f) return BLASR_HIP_ERR_INVALID;
  const int32_t len = (int32_t)(1ll << (2 * tuple_size));
  int32_t nTuples = 0;  // the reference counts in an int as well: wraps beyond 2^31 tuples
  for (int64_t i = 0; i < (int64_t)len; i++) nTuples = (int32_t)((uint32_t)nTuples + (uint32_t)count_table[i]);
  bool ok = fwrite(&len, 4, 1, f) == 1 && fwrite(&nTuples, 4, 1, f) == 1 && fwrite(&tuple_size, 4, 1, f) == 1 &&
            fwrite(count_table, 4, (size_t)len, f) == (size_t)len;
  ok = fclose(f) == 0 && ok;
  return ok ? BLASR_HIP_OK : BLASR_HIP_ERR_INVALID;
} BLASR_HIP_CATCH_CODE

// Returns the number of counts the file holds (with count_table == NULL or too small a capacity nothing is copied).
int64_t blasr_hip_ctab_read(const char *path, int32_t *count_table, int64_t capacity, int32_t *tuple_size) try {
  if (!path) return BLASR_HIP_ERR_INVALID;
  FILE *f = fopen(path, "rb");
  if (!f) return BLASR_HIP_ERR_INVALID;
  int32_t head[3] = {0, 0, 0};
  int64_t out = BLASR_HIP_ERR_INVALID;
  if (fread(head, 4, 3, f) == 3 && head[0] > 0 && head[2] >= 1 && head[2] <= 15 && head[0] == (int32_t)(1ll << (2 * head[2]))) {
    out = head[0];
    if (tuple_size) *tuple_size = head[2];
    if (count_table && capacity >= head[0] && fread(count_table, 4, (size_t)head[0], f) != (size_t)head[0])
      out = BLASR_HIP_ERR_INVALID;
  }
  fclose(f);
  return out;
} BLASR_HIP_CATCH_CODE

}  // extern "C"
