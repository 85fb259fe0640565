// examples/map_fasta.cpp -- a C++ host over the C ABI of include/blasr_hip.h, nothing else: FASTA in, `-m 4` lines out.
//
//   map_fasta reads.fasta genome.fasta genome.fasta.sa [bestn] [-extend]   the whole mapper: anchoring included
//   map_fasta reads.fasta genome.fasta intervals.txt   [bestn] [-extend]   candidate intervals given (e.g. blasrmc -V 1's)
// -extend: the reference's switch of that name (both ends of every SDP alignment extended before the filters)
// -rbao:   the reference's switch of that name (alignments between the chained anchors instead of SDPAlign over the window;
//          whole-mapper form only: the chained anchors come from the library's interval finder)
// -start S -stride N: the reference's switches of those names (Blasr.cpp:4057-4058, ReaderAgglomerate::Advance): reads S,
//   S + N, S + 2N, ... of the file -- how one process per GPU shares a read file: rank r of N runs with -start r -stride N
//   -device r, and the outputs concatenated in rank order hold the lines of the one-process run
// -device D: the HIP device to run on (default 0)
// reads.fastq / reads.fq instead of reads.fasta: read as the reference's FASTQReader reads it (blasr_hip_fastq_read);
// reads.sam: the records of a SAM file as the reference takes them for reads (blasr_hip_sam_reads_read)
//
// What the reference's MapReads does around MapRead (alignment/Blasr.cpp:3193-3917) with the library's pieces:
//   genome      blasr_hip_fasta_read + blasr_hip_fasta_concatenate     (ReadAllSequencesIntoOne, ToUpper: Blasr.cpp:4329, 4354)
//   reads       blasr_hip_fasta_read                                   (ReaderAgglomerate, FASTA)
//   with a .sa  blasr_hip_sa_read + blasr_hip_upload_suffix_array, blasr_hip_count_tuples_device (:4479-4482), then ALL of
//               MapRead in one call, blasr_hip_map_from_reads_host: MapReadToGenome x 2 + SortMatchPosList on the device
//               (:2279-2340), RemoveOverlappingAnchors / FindMaxIncreasingInterval on host threads (:2380-2500), the path
//   with a text file   one interval per line: read start end qStart qEnd strand
//   MapRead     blasr_hip_map_reads_host                               (:2530-2644, C++ inside the library)
//   epilogue    blasr_hip_ref_contig_location, blasr_hip_map_qvs, blasr_hip_m4_line   (:3509-3516, :3086)
// Build: g++ -O2 -std=c++11 -I include examples/map_fasta.cpp -L blasr_amd -lblasr_hip -Wl,-rpath,'$ORIGIN/../blasr_amd'
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <algorithm>
#include <vector>

#include "blasr_hip.h"

static void die(const char *what, blasr_hip_ctx *ctx) {
  fprintf(stderr, "map_fasta: %s%s%s\n", what, ctx ? ": " : "", ctx ? blasr_hip_last_error(ctx) : "");
  exit(1);
}

int main(int argc, char **argv) {
  setenv("GPU_MAX_HW_QUEUES", "8", 0);  // before the first HIP call (INTEGRATION.md 4'): the library leaves the environment alone
  if (argc < 4) die("usage: map_fasta reads.fasta genome.fasta (genome.fasta.sa | intervals.txt) [bestn]", NULL);
  int bestn = 1, extend = 0, rbao = 0, start = 0, stride = 1, device = 0;
  for (int i = 4; i < argc; i++) {
    if (strcmp(argv[i], "-extend") == 0) extend = 1;
    else if (strcmp(argv[i], "-rbao") == 0) rbao = 1;
    else if (strcmp(argv[i], "-start") == 0 && i + 1 < argc) start = atoi(argv[++i]);
    else if (strcmp(argv[i], "-stride") == 0 && i + 1 < argc) stride = atoi(argv[++i]);
    else if (strcmp(argv[i], "-device") == 0 && i + 1 < argc) device = atoi(argv[++i]);
    else bestn = atoi(argv[i]);
  }
  if (start < 0 || stride < 1) die("-start needs a value >= 0, -stride one >= 1", NULL);
  // ---- genome: contigs back to back, an N after each, upper case; the contig table; names up to the first blank
  blasr_hip_fasta *gf = NULL, *rf = NULL;
  if (blasr_hip_fasta_read(argv[2], &gf) != BLASR_HIP_OK) die("cannot read the genome", NULL);
  const int64_t nContigs = blasr_hip_fasta_count(gf);
  std::vector<uint32_t> starts((size_t)nContigs + 1);
  const int64_t genomeLen = blasr_hip_fasta_concatenate(gf, 1, NULL, 0, starts.data());
  if (genomeLen <= 0) die("empty genome", NULL);
  std::vector<uint8_t> genome((size_t)genomeLen);
  blasr_hip_fasta_concatenate(gf, 1, genome.data(), genomeLen, NULL);
  std::vector<std::string> contigNames;
  for (int64_t i = 0; i < nContigs; i++) {
    char buf[1024];
    blasr_hip_fasta_name(gf, i, buf, sizeof(buf));
    contigNames.push_back(buf);
  }
  // ---- reads
  const size_t rl = strlen(argv[1]);
  const bool fastq = (rl > 6 && strcmp(argv[1] + rl - 6, ".fastq") == 0) || (rl > 3 && strcmp(argv[1] + rl - 3, ".fq") == 0);
  const bool samReads = rl > 4 && strcmp(argv[1] + rl - 4, ".sam") == 0;  // (a .sam read file: blasr_hip_sam_reads_read)
  if ((samReads ? blasr_hip_sam_reads_read(argv[1], &rf) : fastq ? blasr_hip_fastq_read(argv[1], &rf) : blasr_hip_fasta_read(argv[1], &rf)) != BLASR_HIP_OK)
    die(rf ? blasr_hip_fasta_error(rf) : "cannot read the reads", NULL);
  const int nInFile = (int)blasr_hip_fasta_count(rf);
  std::vector<blasr_hip_read> reads;
  std::vector<uint8_t> pool;
  std::vector<std::string> readNames;
  for (int r = start; r < nInFile; r += stride) {  // this process's share of the file (-start / -stride)
    const uint8_t *s;
    int64_t n;
    char buf[1024];
    blasr_hip_fasta_record(rf, r, NULL, NULL, &s, &n);
    blasr_hip_fasta_name(rf, r, buf, sizeof(buf));
    blasr_hip_read rd;
    rd.off = (int64_t)pool.size();
    rd.length = (uint32_t)n;
    rd.subread_start = 0;
    rd.subread_end = (uint32_t)n;
    rd.reserved = 0;
    reads.push_back(rd);
    pool.insert(pool.end(), s, s + n);
    // the reference appends the subread's coordinates to the name it prints (SMRTSequence title handling)
    readNames.push_back(std::string(buf) + "/0_" + std::to_string((long long)n));
  }
  const int nReads = (int)reads.size();
  blasr_hip_ctx *ctx = NULL;
  if (blasr_hip_create(device, &ctx) != BLASR_HIP_OK) die("no gfx950 device", NULL);
  // (the contig table also for a single contig: tLength is the contig's length, Blasr.cpp:1291, not the genome's with its N)
  if (blasr_hip_upload_genome(ctx, genome.data(), genomeLen, starts.data(), (int32_t)nContigs))
    die("upload_genome", ctx);
  // ---- candidate intervals: from the reads (a .sa file given) or from a text file
  std::vector<blasr_hip_interval> ivs;
  bool oneCall = false;
  std::vector<blasr_hip_hit> hitsOne;
  std::vector<int64_t> hitOffOne;
  std::vector<uint8_t> seqArenaOne;
  std::vector<uint32_t> opsArenaOne;
  int64_t seqCapOne = 0, nHitsOne = 0;
  const size_t nameLen = strlen(argv[3]);
  if (nameLen > 3 && strcmp(argv[3] + nameLen - 3, ".sa") == 0) {
    blasr_hip_sa *sa = NULL;
    if (blasr_hip_sa_read(argv[3], &sa) != BLASR_HIP_OK) die("cannot read the suffix array", NULL);
    const uint32_t *index, *startPos, *endPos;
    int64_t saLen, tableLen;
    int32_t prefixLen;
    blasr_hip_sa_arrays(sa, &index, &saLen, &startPos, &endPos, &tableLen, &prefixLen);
    if (blasr_hip_upload_suffix_array(ctx, index, saLen, startPos, endPos, tableLen, prefixLen)) die("upload_suffix_array", ctx);
    blasr_hip_sa_free(sa);
    // the whole of MapRead in one call: anchoring in sub-blocks on the device, interval finding on host threads
    // underneath, then the path (blasr_hip_map_from_reads_host)
    blasr_hip_pipeline_params pp;
    blasr_hip_pipeline_default_params(&pp);
    if (pp.anchor.min_match_length < (uint32_t)prefixLen) pp.anchor.min_match_length = (uint32_t)prefixLen;  // Blasr.cpp:4433-4436
    pp.map.extend_alignments = extend;
    pp.map.refine_between_anchors_only = rbao;
    std::vector<int32_t> countTable((size_t)1 << 16);  // tuples of params.lookupTableLength = 8 bases (Blasr.cpp:4479-4482)
    if (blasr_hip_count_tuples_device(ctx, 8, countTable.data())) die("count_tuples", ctx);
    oneCall = true;
    ivs.resize((size_t)nReads * (size_t)pp.interval.n_candidates + 1);
    hitsOne.resize(ivs.size() + 1);
    hitOffOne.resize((size_t)nReads + 1);
    seqCapOne = 64;
    for (int r = 0; r < nReads; r++) seqCapOne += (int64_t)pp.interval.n_candidates * (5ll * reads[r].length + 4096);
    seqArenaOne.resize((size_t)seqCapOne);
    opsArenaOne.resize((size_t)seqCapOne);
    blasr_hip_map_fetch fetchOne = {seqArenaOne.data(), seqCapOne, 0, opsArenaOne.data(), seqCapOne, 0};
    int64_t nIvs = 0;
    if (blasr_hip_map_from_reads_host(ctx, genome.data(), starts.data(), (int32_t)nContigs + 1, countTable.data(), 8, nReads,
                                      reads.data(), pool.data(), (int64_t)pool.size(), &pp, hitsOne.data(), (int64_t)hitsOne.size(),
                                      &nHitsOne, hitOffOne.data(), &fetchOne, ivs.data(), (int64_t)ivs.size(), &nIvs) != BLASR_HIP_OK)
      die("map_from_reads", ctx);
    ivs.resize((size_t)nIvs);
  } else {
    FILE *f = fopen(argv[3], "r");
    if (!f) die("cannot read the intervals", NULL);
    long long rd, st, en, qs, qe, sd;
    while (fscanf(f, "%lld %lld %lld %lld %lld %lld", &rd, &st, &en, &qs, &qe, &sd) == 6) {
      // (the file numbers the reads as they stand in the read file: keep this process's, renumbered)
      if (rd < start || (rd - start) % stride != 0 || (rd - start) / stride >= nReads) continue;
      blasr_hip_interval iv = {(uint32_t)st, (uint32_t)en, (uint32_t)qs, (uint32_t)qe, (int32_t)((rd - start) / stride), (int32_t)sd};
      ivs.push_back(iv);
    }
    fclose(f);
  }
  // ---- MapRead for the whole block
  std::vector<blasr_hip_hit> hits;
  std::vector<int64_t> hitOff;
  std::vector<uint8_t> seqArena;
  std::vector<uint32_t> opsArena;
  int64_t nHits = 0;
  blasr_hip_map_params prm;  // (the one-call route used the same defaults)
  blasr_hip_map_default_params(&prm);
  prm.extend_alignments = extend;
  if (oneCall) {  // (already mapped above)
    hits.swap(hitsOne);
    hitOff.swap(hitOffOne);
    seqArena.swap(seqArenaOne);
    opsArena.swap(opsArenaOne);
    nHits = nHitsOne;
  } else {
    hits.resize(ivs.size() + 1);
    hitOff.resize((size_t)nReads + 1);
    int64_t seqCap = 64;
    for (size_t k = 0; k < ivs.size(); k++) seqCap += 3ll * reads[ivs[k].read].length + (ivs[k].end - ivs[k].start);
    seqArena.resize((size_t)seqCap);
    opsArena.resize((size_t)seqCap);
    blasr_hip_map_fetch fetch = {seqArena.data(), seqCap, 0, opsArena.data(), seqCap, 0};
    if (blasr_hip_map_reads_host(ctx, nReads, reads.data(), pool.data(), (int64_t)pool.size(), (int32_t)ivs.size(), ivs.data(), &prm,
                                 hits.data(), (int64_t)hits.size(), &nHits, hitOff.data(), &fetch) != BLASR_HIP_OK)
      die("map_reads", ctx);
  }
  // ---- per read: StoreMapQVs over its alignments, then the line(s) of the best
  blasr_hip_mapqv_params qp;
  static const int32_t logProb[25] = {0, 15, 15, 15, 15, 15, 0, 15, 15, 15, 15, 15, 0, 15, 15, 15, 15, 15, 0, 15, 15, 15, 15, 15, 0};
  memcpy(qp.matrix, logProb, sizeof(logProb));  // SMRTLogProbMatrix (ScoreMatrices.h:28-34)
  qp.ins = 5, qp.del = 5, qp.affine_open = 50, qp.affine_extend = 0;  // InitializeScoreFunction (MappingParameters.h:587-592)
  qp.affine_align = 1;
  qp.min_fraction_overlapping = 0.75f;
  for (int r = 0; r < nReads; r++) {
    const int64_t a = hitOff[r], b = hitOff[r + 1];
    if (a == b) continue;
    std::vector<blasr_hip_mapqv_alignment> alns;
    for (int64_t i = a; i < b; i++) {
      const blasr_hip_hit &h = hits[i];
      blasr_hip_mapqv_alignment m;
      m.q_aligned_seq = seqArena.data() + h.q_off;
      m.t_aligned_seq = seqArena.data() + h.t_off;
      m.ops = opsArena.data() + h.ops_off;
      m.n_ops = h.n_ops;
      m.q_pos = h.q_pos;
      m.t_pos = h.t_pos;
      m.q_align_start = h.q_aligned_seq_pos + (uint32_t)h.q_pos;
      m.q_align_end = m.q_align_start + (uint32_t)(h.n_match + h.n_mismatch + h.n_ins);
      m.q_strand = 0;
      m.q_length = h.q_length;
      m.n_match = h.n_match;
      m.reserved = 0;
      alns.push_back(m);
    }
    std::vector<int32_t> qv(alns.size());
    // the mapQVs are stored only when the best alignment beats maxScore (:3511-3516); they are 50 until then
    if (hits[a].score < prm.max_score) {
      if (blasr_hip_map_qvs(alns.data(), (int32_t)alns.size(), &qp, NULL, qv.data(), NULL) < 0) die("map_qvs", NULL);
    } else {
      std::fill(qv.begin(), qv.end(), 50);
    }
    blasr_hip_print_params gates = {prm.max_score, prm.min_pct_identity, 0, 0};  // PrintAlignment (:2957-2978)
    for (int64_t i = a; i < b && i - a < bestn; i++) {
      const blasr_hip_hit &h = hits[i];
      if (blasr_hip_alignment_is_printed(h.score, h.pct_similarity, h.t_aligned_seq_length, qv[(size_t)(i - a)], &gates) != 1)
        continue;
      int32_t contig = 0;
      uint32_t contigPos = 0;
      blasr_hip_ref_contig_location(h.t_aligned_seq_pos, h.t_aligned_seq_length, h.t_strand, starts.data(), (int32_t)nContigs + 1,
                                    (uint32_t)genomeLen, &contig, &contigPos);
      blasr_hip_m4_fields m4;
      m4.q_name = readNames[r].c_str();
      m4.t_name = contigNames[(size_t)contig].c_str();
      m4.score = h.score;
      m4.pct_similarity = h.pct_similarity;
      m4.q_strand = 0;
      m4.q_align_start = alns[(size_t)(i - a)].q_align_start;
      m4.q_align_end = alns[(size_t)(i - a)].q_align_end;
      m4.q_length = h.q_length;
      m4.t_strand = h.t_strand;
      m4.t_align_start = contigPos + (uint32_t)h.t_pos;
      m4.t_align_end = m4.t_align_start + (uint32_t)(h.n_match + h.n_mismatch + h.n_del);
      m4.t_length = h.t_length;
      m4.map_qv = qv[(size_t)(i - a)];
      char line[4096];
      int64_t len = 0;
      if (blasr_hip_m4_line(&m4, line, sizeof(line), &len) < 0) die("m4_line", NULL);
      fwrite(line, 1, (size_t)len, stdout);
    }
  }
  blasr_hip_destroy(ctx);
  blasr_hip_fasta_free(gf);
  blasr_hip_fasta_free(rf);
  return 0;
}
