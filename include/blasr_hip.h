/* blasr_hip.h -- C ABI of the MI355X refine stage (libblasr_hip.so).
 *
 * The reference (mchaisso/blasr) has no FFI: its refine stage is a set of
 * header templates instantiated inside alignment/Blasr.cpp and called once per
 * candidate interval from one pthread.  This library replaces exactly those
 * calls with batch entry points, so that N host threads (or one driver) hand
 * over ARRAYS of candidates:
 *
 *   blasr_hip_refine_*      replaces the body of RefineAlignment's guided
 *                           branch, alignment/Blasr.cpp:850-914, i.e.
 *                           AffineGuidedAlign (AffineGuidedAlign.h:561-578 ->
 *                           :31-488) + ComputeAlignmentStats
 *                           (AlignmentUtils.h:535-584), called per candidate
 *                           by RefineAlignments (Blasr.cpp:2164-2181)
 *   blasr_hip_sdp_*         replaces SDPAlign + ComputeAlignmentStats in
 *                           AlignIntervals, alignment/Blasr.cpp:1716-1724
 *
 * Conventions: plain C, fixed-width integers, no STL/torch types; every
 * function returns a status (0 = ok, < 0 = error, see blasr_hip_status);
 * per-candidate problems are reported in the result record's `status`, never
 * by abort()/exit() as the reference does.  One context per host thread: a
 * context owns a HIP stream and a grow-only device workspace (the analogue of
 * the per-thread MappingBuffers, Blasr.cpp:318-369).  There is NO CPU
 * fallback: every entry point fails with BLASR_HIP_ERR_NO_DEVICE when no
 * gfx950 device is usable.
 *
 * Sequences are ASCII, one byte per base, as the reference's DNASequence
 * (DNASequence.h:16-24).  A/C/G/T (either case) are codes 0..3, everything
 * else is N (code 4); score matrices are indexed [query code][target code]
 * like DistanceMatrixScoreFunction::Match (DistanceMatrixScoreFunction.h:99-104).
 */
#ifndef BLASR_HIP_H_
#define BLASR_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: blasr_hip_map_params grew the -extend fields (and blasr_hip_pipeline_params with it); a caller built against
 * version 1 must be recompiled (compare blasr_hip_abi_version() with the header's value at start-up). */
/* 3: blasr_hip_sdp_params grew sw_align_type, blasr_hip_map_params the -rbao fields (and the chains' two pointers). */
/* 4: blasr_hip_map_params' reserved word became use_guided_align (a caller that fills the struct by hand instead of through
 * blasr_hip_map_default_params would get the k-band refiner with its old 0); BLASR_HIP_ERR_INTERNAL; the k-band entry points
 * take bands of any width and report TASK_EMPTY_GUIDE for an empty candidate. */
/* 5: blasr_hip_sdp_last_kernels / blasr_hip_sdp_kernel_name (every heavy SDP kernel timed on the stream it ran on); no
 * struct of version 4 changed. */
/* 6: blasr_hip_borrow_index; blasr_hip_pipeline_params' reserved word became `overlap` (0 keeps version 5's behaviour). */
#define BLASR_HIP_ABI_VERSION 6

typedef struct blasr_hip_ctx blasr_hip_ctx;

enum blasr_hip_status {
  BLASR_HIP_OK = 0,
  BLASR_HIP_ERR_INVALID = -1,     /* bad argument */
  BLASR_HIP_ERR_NO_DEVICE = -2,   /* no usable gfx950 device */
  BLASR_HIP_ERR_HIP = -3,         /* a HIP runtime call failed (see blasr_hip_last_error) */
  BLASR_HIP_ERR_NOMEM = -4,       /* device workspace could not grow */
  BLASR_HIP_ERR_UNSUPPORTED = -5, /* parameters outside the kernel's domain */
  BLASR_HIP_ERR_CAPACITY = -6,    /* caller-provided output arena too small */
  BLASR_HIP_ERR_INTERNAL = -7     /* a C++ exception other than bad_alloc was caught at the boundary (no exception ever
                                     crosses it: bad_alloc comes back as BLASR_HIP_ERR_NOMEM) */
};

/* per-candidate status bits in result records */
enum {
  BLASR_HIP_TASK_OK = 0,
  BLASR_HIP_TASK_EMPTY_GUIDE = 1, /* no guide blocks: the reference returns 0 and leaves the
                                     candidate untouched (AffineGuidedAlign.h:151-155) */
  BLASR_HIP_TASK_BAD_GUIDE = 2,   /* guide blocks not monotone / outside the sequences: the
                                     reference would run off its buffers or assert */
  BLASR_HIP_TASK_BAD_PATH = 4,    /* traceback met NoArrow: reference prints "path has gone
                                     awry" and exit(1)s (AffineGuidedAlign.h:394-408) */
  BLASR_HIP_TASK_OPS_OVERFLOW = 8 /* ops arena exhausted; counts are valid, ops are not */
};

/* DistanceMatrixScoreFunction + BaseScoreFunction fields that reach the
 * kernels (DistanceMatrixScoreFunction.h:10-46, BaseScoreFunction.h:4-26). */
typedef struct blasr_hip_score_params {
  int32_t matrix[25]; /* [query][target], codes A C G T N; default SMRTDistanceMatrix */
  int32_t ins;        /* params.insertion  (5)  */
  int32_t del;        /* params.deletion   (5)  */
  int32_t affine_open;   /* params.affineOpen   (50) */
  int32_t affine_extend; /* params.affineExtend (0)  */
} blasr_hip_score_params;

/* A batch of refine candidates, structure-of-arrays.  For candidate i:
 *   query  = seq[q_off[i] .. q_off[i]+q_len[i])   (qSeq  of Blasr.cpp:856-859)
 *   target = seq[t_off[i] .. t_off[i]+t_len[i])   (tSeq  of Blasr.cpp:850-853)
 *   guide  = blocks[3*blk_off[i] .. ) n_blocks[i] triplets (qPos,tPos,length),
 *            the candidate's Alignment.blocks in the coordinates of query/target
 * All pointers are DEVICE pointers for the *_device entry points and HOST
 * pointers for the *_host ones. */
typedef struct blasr_hip_refine_batch {
  int32_t n;
  const uint8_t *seq;
  const int64_t *q_off;
  const int32_t *q_len;
  const int64_t *t_off;
  const int32_t *t_len;
  const int32_t *blocks;
  const int64_t *blk_off;
  const int32_t *n_blocks;
} blasr_hip_refine_batch;

/* One record per candidate: everything RefineAlignment copies back into the
 * T_AlignmentCandidate (Blasr.cpp:888-914).  The base-by-base alignment is
 * returned as run-length ops (see below) from which blocks[] and gaps[][] are
 * rebuilt 1:1. */
typedef struct blasr_hip_refine_result {
  int32_t status;     /* BLASR_HIP_TASK_* bits */
  int32_t q_pos;      /* refinedAlignment.qPos (after RemoveAlignmentPrefixGaps) */
  int32_t t_pos;      /* refinedAlignment.tPos */
  int32_t dp_score;   /* value AffineGuidedAlign returns: S[qEnd-1][tEnd-1] */
  int32_t n_cells;    /* alignment.nCells (AffineGuidedAlign.h:470) */
  int32_t score;      /* ComputeAlignmentStats score (affine form) -- the printed one */
  int32_t n_match;
  int32_t n_mismatch;
  int32_t n_ins;
  int32_t n_del;
  int32_t n_ops;      /* number of ops of this candidate */
  int32_t reserved;
  int64_t ops_off;    /* first op of this candidate in the ops arena */
} blasr_hip_refine_result;

/* op = (length << 2) | type; forward order, first and last op are always M. */
enum { BLASR_HIP_OP_M = 0, /* block: `length` aligned columns */
       BLASR_HIP_OP_I = 1, /* Gap::Target, arrow Up: `length` query bases unaligned */
       BLASR_HIP_OP_D = 2  /* Gap::Query, arrow Left: `length` target bases unaligned */ };
#define BLASR_HIP_OP_TYPE(op) ((op) & 3u)
#define BLASR_HIP_OP_LEN(op) ((op) >> 2)

/* ---- context ---------------------------------------------------------- */
int blasr_hip_abi_version(void);
/* Creates a context on HIP device `device`; fails unless it is a gfx950. */
int blasr_hip_create(int device, blasr_hip_ctx **ctx);
int blasr_hip_destroy(blasr_hip_ctx *ctx);
/* Human-readable text of the last failure on this context (never NULL). */
const char *blasr_hip_last_error(const blasr_hip_ctx *ctx);
/* Upper bound for the grow-only device workspace, bytes (default 48 GiB);
 * larger batches are processed in several waves of kernels. */
int blasr_hip_set_workspace_limit(blasr_hip_ctx *ctx, int64_t bytes);
/* Device memory the context holds right now, bytes: the workspace (shared by the SDP and the refine stage), the
 * mapper's arenas and a genome it owns.  Everything a call needs beyond it is released before the call returns. */
int64_t blasr_hip_held_bytes(const blasr_hip_ctx *ctx);
/* The context's stream as a hipStream_t (void* to keep HIP out of this header). */
void *blasr_hip_stream(blasr_hip_ctx *ctx);

/* ---- refine: AffineGuidedAlign + ComputeAlignmentStats ---------------- */
/* band = params.bandSize (16 after MakeSane, MappingParameters.h).
 * results: n records; ops: arena of ops_capacity uint32; *ops_used receives
 * the number of ops written (host pointer, may be NULL).  Supported domain:
 * affine_open >= 0, del >= affine_extend >= 0 (else ERR_UNSUPPORTED).
 * Device variant: batch pointers, results and ops are device memory; work is
 * enqueued on the context's stream and completed before return. */
int blasr_hip_refine_device(blasr_hip_ctx *ctx, const blasr_hip_refine_batch *batch, int32_t band,
                            const blasr_hip_score_params *score,
                            blasr_hip_refine_result *results, uint32_t *ops,
                            int64_t ops_capacity, int64_t *ops_used);
/* Host variant: uploads the batch, runs the device variant, downloads. */
int blasr_hip_refine_host(blasr_hip_ctx *ctx, const blasr_hip_refine_batch *batch, int32_t band,
                          const blasr_hip_score_params *score, blasr_hip_refine_result *results,
                          uint32_t *ops, int64_t ops_capacity, int64_t *ops_used);

/* Timing of the kernels of the last refine call on this context, measured
 * with HIP events on the context's stream: ms[0] = guide geometry,
 * ms[1] = band fill (dominant), ms[2] = traceback+stats, ms[3] = op compaction;
 * counters[0] = band cells filled, counters[1] = algorithmic bytes of the fill
 * (DESIGN.md), counters[2] = fill kernel launches. */
int blasr_hip_refine_last_timing(const blasr_hip_ctx *ctx, float ms[4], int64_t counters[3]);
/* The dominant kernel on its own: the fillp_kernel launches (four candidates per wavefront) of the last refine
 * call -- HIP-event time on the library's stream, number of launches, algorithmic bytes (1 B per band cell +
 * the bases read + 8 B per row record) and band cells of the candidates those launches processed. */
int blasr_hip_refine_last_fill4(const blasr_hip_ctx *ctx, float *ms, int64_t *launches, int64_t *algorithmic_bytes,
                                int64_t *cells);
/* Every heavy kernel of the refine stage on its own (last refine call on this context): HIP-event time of its
 * launches on the stream they ran on, and the algorithmic bytes / band cells of exactly the candidates those launches
 * processed.  Fill kernels: 1 B per band cell (arrows out) + the bases read + 8 B per row record;
 * traceback: one 16-byte arrow window + one 8-byte row record per row walked.  `kind` names the kernel:
 * blasr_hip_kernel_name(kind) is its full instantiation as rocprofv3 prints it. */
enum { BLASR_HIP_KERNEL_FILL_NARROW = 0, /* fillp_kernel<16, ext0, false>: four candidates per wavefront (extension penalty
                                            != 0, or a matrix entry outside [-32, 31]; else FILL_EIGHT runs) */
       BLASR_HIP_KERNEL_FILL_HYBRID = 1, /* candidates with wide rows: fillh8_kernel (eight per wavefront, each at its own row;
                                            the record's ext0 is 2 then) or, outside fill8_kernel's domain, fillp_kernel<16, ext0, true> */
       BLASR_HIP_KERNEL_FILL_WAVE = 2,   /* fill_kernel: one candidate per wavefront (compare-based variant) */
       BLASR_HIP_KERNEL_TRACEBACK = 3,   /* traceback_kernel */
       /* the k-band refiner (blasr_hip_kband_*): cells = the band cells inside the matrix, sum over the rows q of
        * min(tLen, q + k) - max(1, q - k) + 1; algorithmic bytes = the arrows written (half a byte per cell in the
        * strip kernel, one byte in the rows kernel) + the bases read */
       BLASR_HIP_KERNEL_KBAND_STRIPS = 4,    /* kband_wide_kernel: bands of any width, strips of the (q, t) matrix */
       BLASR_HIP_KERNEL_KBAND_TRACEBACK = 5, /* kband_wide_traceback_kernel */
       BLASR_HIP_KERNEL_KBAND_ROWS = 6,      /* kband_fill_kernel: band rows in LDS, 2k+1 <= 1088 */
       BLASR_HIP_KERNEL_FILL_EIGHT = 7,      /* fill8_kernel: the narrow candidates, eight per wavefront, each at its own row */
       BLASR_HIP_KERNEL_KINDS = 8 };
typedef struct blasr_hip_kernel_record {
  int32_t kind;
  int32_t launches;
  float ms;
  int32_t ext0;               /* 1: the affine_extend == 0 instantiation ran; 2 (FILL_HYBRID only): fillh8_kernel ran */
  int64_t algorithmic_bytes;
  int64_t cells;
  int64_t candidates;
} blasr_hip_kernel_record;
/* Fills out[0..min(n, capacity)) with the kinds that ran in the last refine or k-band call; returns their number n
 * (or a negative error code). */
int blasr_hip_refine_last_kernels(const blasr_hip_ctx *ctx, blasr_hip_kernel_record *out, int capacity);
const char *blasr_hip_kernel_name(int kind, int ext0);
/* Identifies the BUILD of the library: a hash over its sources taken by the Makefile.  Profiles collected with
 * rocprofv3 carry it, and bench.py reports PMC traffic only from a profile whose id equals the loaded library's. */
const char *blasr_hip_build_id(void);

/* ---- SDP: SDPAlign + ComputeAlignmentStats ----------------------------- */
/* The parameters AlignIntervals passes to SDPAlign (alignment/Blasr.cpp:1716-1722;
 * defaults MappingParameters.h:192-361).  Supported on the device: align_type
 * Local (0) or Global (1), word_size <= 16, |sdp_ins|, |sdp_del| < 64, sequences shorter than 2^24 bases (else
 * ERR_UNSUPPORTED); detailed = 0 (no fills between blocks, SDPAlign.h:497-499) and extend_front = 1 (front fill of
 * a Local call :437-474, tail fill with word size k / 2 :546-598) included. */
typedef struct blasr_hip_sdp_params {
  int32_t word_size;        /* sdpTupleSize (11) */
  int32_t sdp_ins;          /* params.sdpIns (5) */
  int32_t sdp_del;          /* params.sdpDel (10) */
  float indel_rate;         /* params.indelRate * 3 (0.9) */
  int32_t align_type;       /* Local */
  int32_t detailed;         /* params.detailedSDPAlignment (1) */
  int32_t extend_front;     /* params.extendFrontAlignment (0) */
  int32_t prefix_length;    /* params.sdpPrefix (50) */
  int32_t recurse;          /* params.recurse (2) */
  int32_t recurse_over;     /* params.recurseOver (1000) */
  int32_t max_per_position; /* params.sdpMaxAnchorsPerPosition (0) */
  int32_t sw_align_type;    /* smithWatermanAlignType: 0 = SDPAlign's default (EndAnchored, SDPAlign.h:116); the one place it
                               acts is the front fill's recursive call, where it lands in the maxMatchesPerPosition slot
                               (:456).  AlignSubstring of -rbao passes Global = 1 (Blasr.cpp:1089) */
} blasr_hip_sdp_params;

/* Candidate i: query = seq[q_off[i] ..+q_len[i]) (alignment->qAlignedSeq, Blasr.cpp:1332-1336),
 * target = seq[t_off[i] ..+t_len[i]) (alignment->tAlignedSeq, :1311-1325). */
typedef struct blasr_hip_sdp_batch {
  int32_t n;
  const uint8_t *seq;
  const int64_t *q_off;
  const int32_t *q_len;
  const int64_t *t_off;
  const int32_t *t_len;
} blasr_hip_sdp_batch;

/* What AlignIntervals keeps of the SDP alignment: Alignment.qPos/tPos/blocks, the value SDPAlign
 * returns, and the ComputeAlignmentStats fields (Blasr.cpp:1723-1724) that order and filter the
 * candidates before refinement (:2547, :2594-2596). */
typedef struct blasr_hip_sdp_result {
  int32_t status;      /* 0, or BLASR_HIP_TASK_OPS_OVERFLOW when the blocks arena was too small */
  int32_t score;       /* return value of SDPAlign (SDPAlign.h:624) */
  int32_t q_pos;       /* alignment.qPos */
  int32_t t_pos;       /* alignment.tPos */
  int32_t n_blocks;
  int32_t stats_score; /* alignment.score after ComputeAlignmentStats(affine) */
  int32_t n_match;
  int32_t n_mismatch;
  int32_t n_ins;
  int32_t n_del;
  int64_t blk_off;     /* first block (triplet index) of this candidate in the blocks arena */
} blasr_hip_sdp_result;

/* blocks: arena of blocks_capacity (qPos,tPos,length) int32 triplets, relative to (q_pos,t_pos)
 * like Alignment.blocks; *blocks_used receives the number of triplets produced. */
int blasr_hip_sdp_device(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *batch, const blasr_hip_sdp_params *sdp,
                         const blasr_hip_score_params *score, blasr_hip_sdp_result *results, int32_t *blocks,
                         int64_t blocks_capacity, int64_t *blocks_used);
int blasr_hip_sdp_host(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *batch, const blasr_hip_sdp_params *sdp,
                       const blasr_hip_score_params *score, blasr_hip_sdp_result *results, int32_t *blocks,
                       int64_t blocks_capacity, int64_t *blocks_used);
/* Wall time (ms) of the last SDP call's device phases: [0] k-mer keys + sort, [1] match + order,
 * [2] chain, [3] blocks + gap fills + assembly; counters: [0] SDPAlign invocations (all levels),
 * [1] fragments chained, [2] SWAlign gap fills. */
int blasr_hip_sdp_last_timing(const blasr_hip_ctx *ctx, float ms[4], int64_t counters[3]);
/* Algorithmic bytes of the same four phases of the last SDP call, by the model of SURVEY.md 8(d) summed over all
 * SDPAlign invocations (recursion levels included): [0] (tLen-k+1) x 8 B x (1 + 2P) key records through P = ceil(2k/8)
 * radix passes, [1] (qLen-k+1) x 8 B of query k-mers + 16 B per fragment written, [2] 16 B per fragment read by the
 * sweep, [3] 12 B per block. */
int blasr_hip_sdp_last_bytes(const blasr_hip_ctx *ctx, int64_t bytes[4]);
/* The heavy kernels of the last SDP call, each on its own: HIP events around every launch on the stream it ran on,
 * summed over the recursion levels (and the pieces of a batch taken in pieces).  The sweep (SDPLongestCommonSubsequence,
 * sdp/SparseDynamicProgramming.h:70-322) is split by the kernel that took an invocation; algorithmic bytes by SURVEY.md
 * 8(d)'s model for the invocations the kernel processed: a sweep kernel 16 B per fragment read; a matcher 8 B per query
 * position + 16 B per fragment written.  `concurrent` = how many kernels of the stage ran side by side with this one on
 * other streams (their times overlap: the sum over the records exceeds the phase's wall time). */
enum { BLASR_HIP_SDP_KERNEL_MATCH_INDEX = 0, /* sdp_match_index_kernel<false> + <true> + the copy kernel: targets >= 1536 bases */
       BLASR_HIP_SDP_KERNEL_MATCH_SHORT = 1, /* sdp_match_kernel: the other targets */
       BLASR_HIP_SDP_KERNEL_SMALL = 2,       /* sdp_small4_kernel + sdp_small_kernel: order .. sweep of <= 64 fragments, fused */
       BLASR_HIP_SDP_KERNEL_CHAIN4 = 3,      /* sdp_chain4_kernel: near-diagonal invocations, four per wavefront */
       BLASR_HIP_SDP_KERNEL_CHAIN_BITS_SMALL = 4, /* sdp_chain_bits_kernel<512>: scattered, target <= 16384 columns */
       BLASR_HIP_SDP_KERNEL_CHAIN_BITS_LARGE = 5, /* sdp_chain_bits_kernel<4096>: target <= 131072 columns */
       BLASR_HIP_SDP_KERNEL_CHAIN_BITS4 = 6, /* sdp_chain_bits4_kernel<false>: short scattered invocations, four per wavefront (target <= 4096 columns) */
       BLASR_HIP_SDP_KERNEL_CHAIN_ROWS = 7,  /* sdp_chain_rows_kernel: scattered invocations, a batch of sweep rows per step */
       BLASR_HIP_SDP_KERNEL_FILLS = 8,       /* sw_small_kernel + sw_kernel: SWAlign(Global) gap fills */
       BLASR_HIP_SDP_KERNEL_CHAIN_BITS_MID = 9, /* sdp_chain_bits_kernel<2048>: target <= 65536 columns */
       BLASR_HIP_SDP_KERNEL_CHAIN_BITS4_WIDE = 10, /* sdp_chain_bits4_kernel<true>: the same on targets of 4097 .. 16384 columns */
       BLASR_HIP_SDP_KERNEL_KINDS = 11 };
typedef struct blasr_hip_sdp_kernel_record {
  int32_t kind;
  int32_t launches;
  float ms;
  int32_t concurrent;
  int64_t algorithmic_bytes;
  int64_t fragments;    /* fragments (sweep kernels, matchers) or fills */
  int64_t invocations;  /* SDPAlign invocations the kernel took */
} blasr_hip_sdp_kernel_record;
int blasr_hip_sdp_last_kernels(const blasr_hip_ctx *ctx, blasr_hip_sdp_kernel_record *out, int capacity);
const char *blasr_hip_sdp_kernel_name(int kind);


/* ------------------------------------------------------------------------------------------------
 * a-1: candidate windows -- what AlignIntervals does before it calls SDPAlign
 * (alignment/Blasr.cpp:1208-1336): stretch the anchor interval by approximateMaxInsertionRate x the
 * unanchored read ends, clip to the genome / contig, copy the window (reverse-complemented for
 * Reverse-strand intervals, DNASequence::CopyAsRC) next to the forward read's subread.
 * The genome stays resident on the device.
 * ------------------------------------------------------------------------------------------------ */
/* WeightedInterval (datastructures/anchoring/WeightedInterval.h:12-29): the fields a task needs */
typedef struct blasr_hip_interval {
  uint32_t start, end;     /* genome span of the anchor cluster */
  uint32_t q_start, q_end; /* read span of the anchor cluster, on the strand of the interval */
  int32_t read;            /* index into the read table */
  int32_t strand;          /* GetStrandIndex(): 0 Forward, 1 Reverse */
} blasr_hip_interval;

typedef struct blasr_hip_read {
  int64_t off;             /* first base in the read pool (forward strand, ASCII) */
  uint32_t length;
  uint32_t subread_start;  /* SMRTSequence::subreadStart / subreadEnd */
  uint32_t subread_end;
  uint32_t reserved;
} blasr_hip_read;

typedef struct blasr_hip_window_params {
  float approximate_max_insertion_rate; /* params.approximateMaxInsertionRate (1.30) */
  int32_t extend_ends;                  /* params.extendEnds (1) */
  int32_t do_global_alignment;          /* params.doGlobalAlignment (0) */
  int32_t reserved;
} blasr_hip_window_params;

/* the T_AlignmentCandidate fields AlignIntervals sets here (Blasr.cpp:1291-1336) */
typedef struct blasr_hip_window {
  uint32_t t_aligned_seq_pos;    /* on the strand of the interval (:1309, :1323) */
  uint32_t t_aligned_seq_length;
  int32_t t_strand;
  int32_t t_index;               /* contig (seqdb index), 0 without a contig table */
  uint32_t t_length;             /* contig length as the reference computes it (:1291, :1301) */
  uint32_t q_aligned_seq_pos;    /* read.subreadStart (:1332) */
  uint32_t q_aligned_seq_length;
  uint32_t q_length;             /* read.length */
} blasr_hip_window;

/* Keep `length` ASCII bases resident on the device.  contig_starts: n_contigs + 1 offsets
 * (SequenceIndexDatabase::seqStartPos) or NULL when the genome is one sequence (useSeqDB = 0). */
int blasr_hip_upload_genome(blasr_hip_ctx *ctx, const uint8_t *ascii, int64_t length, const uint32_t *contig_starts,
                            int32_t n_contigs);

/* Device pointers throughout.  Writes, per interval, the query then the target window into seq_out and
 * the (q_off, q_len, t_off, t_len) arrays of a blasr_hip_sdp_batch over seq_out.  *seq_used (host)
 * receives the bytes needed; BLASR_HIP_ERR_CAPACITY when seq_capacity is too small. */
int blasr_hip_windows_device(blasr_hip_ctx *ctx, int32_t n, const blasr_hip_interval *intervals,
                             const blasr_hip_read *reads, const uint8_t *read_pool,
                             const blasr_hip_window_params *params, uint8_t *seq_out, int64_t seq_capacity,
                             int64_t *q_off, int32_t *q_len, int64_t *t_off, int32_t *t_len,
                             blasr_hip_window *windows, int64_t *seq_used);
/* Same with host pointers (n_reads / read_pool_bytes size the uploads). */
int blasr_hip_windows_host(blasr_hip_ctx *ctx, int32_t n, const blasr_hip_interval *intervals, int32_t n_reads,
                           const blasr_hip_read *reads, const uint8_t *read_pool, int64_t read_pool_bytes,
                           const blasr_hip_window_params *params, uint8_t *seq_out, int64_t seq_capacity,
                           int64_t *q_off, int32_t *q_len, int64_t *t_off, int32_t *t_len,
                           blasr_hip_window *windows, int64_t *seq_used);

/* ------------------------------------------------------------------------------------------------
 * a-10: ordering and pruning of a read's candidates between the stages (host code in the reference
 * and here; <= nCandidates records per read): SortAlignmentPointersByScore (Blasr.cpp:226-236),
 * RemoveLowQualityAlignments (:2022-2079), RemoveOverlappingAlignments (:2081-2161).
 * `order` lists the live candidates (indices into `cands`), is rewritten in place, and the return
 * value is the new count (< 0: error).  No device is involved.
 * ------------------------------------------------------------------------------------------------ */
typedef struct blasr_hip_candidate {
  int32_t score;
  uint32_t t_pos;
  uint32_t t_aligned_seq_pos;
  uint32_t t_end;          /* Alignment::TEnd(): tPos + length of the last block, 0 without blocks */
  int32_t t_index;
  int32_t t_strand;
  float pct_similarity;
  int32_t n_blocks;
  int64_t blk_off;         /* first (qPos,tPos,length) triplet of this candidate in `blocks` */
} blasr_hip_candidate;

int blasr_hip_sort_candidates(const blasr_hip_candidate *cands, int32_t n, int32_t *order);
/* may overwrite cands[i].score with max_score (the maxGap flagging of :2031-2046) */
int blasr_hip_remove_low_quality(blasr_hip_candidate *cands, const int32_t *blocks, int32_t n, int32_t *order,
                                 int32_t n_candidates, int32_t max_gap, int32_t max_score);
int blasr_hip_remove_overlapping(const blasr_hip_candidate *cands, int32_t n, int32_t *order,
                                 float min_pct_identity);

/* ------------------------------------------------------------------------------------------------
 * RefineAlignment's trimming (alignment/Blasr.cpp:850-859) inside the library: the SDP stage's own batch,
 * result records and block arena go straight into the refine stage.  Candidate i of the call is candidate
 * sel[i] of the SDP batch (sel = NULL: all n_sel = batch->n of them, in order); its query / target are the SDP
 * window cut to [q_pos, q_pos + last block's qPos + length) / [t_pos, ...), its guide the SDP blocks.
 * A candidate without blocks comes back with BLASR_HIP_TASK_EMPTY_GUIDE.  Device pointers throughout.
 * ------------------------------------------------------------------------------------------------ */
int blasr_hip_refine_from_sdp_device(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *batch,
                                     const blasr_hip_sdp_result *sdp_results, const int32_t *blocks, int32_t n_sel,
                                     const int32_t *sel, int32_t band, const blasr_hip_score_params *score,
                                     blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity,
                                     int64_t *ops_used);

/* ------------------------------------------------------------------------------------------------
 * MapRead from the candidate intervals on (alignment/Blasr.cpp:2530-2644), for a whole block of reads at once:
 *   AlignIntervals: windows (:1208-1336) -> SDPAlign + ComputeAlignmentStats per interval (:1716-1724)
 *   sort by score (:2547), RemoveLowQualityAlignments (:2594-2596)
 *   RefineAlignments on the survivors (:2607, :2164-2181, copy-back :888-914), sort
 *   RemoveLowQualityAlignments, RemoveOverlappingAlignments (:2615-2616)
 * The per-interval stages run as device batches over all reads, the per-read ordering and pruning on the host
 * in the reference's order (C++, as there).  What comes back per read is the list of alignments MapRead leaves in
 * alignmentPtrs, best first: the input of AssignRefContigLocation / StoreMapQVs / the printers
 * (blasr_hip_ref_contig_location, blasr_hip_map_qvs, blasr_hip_m4_line, blasr_hip_cigar, blasr_hip_sam_line).
 * Anchoring -- where the intervals come from -- is the caller's (SURVEY next-1).
 * The genome must be resident (blasr_hip_upload_genome / blasr_hip_adopt_genome_device).
 * ------------------------------------------------------------------------------------------------ */
typedef struct blasr_hip_map_params {
  blasr_hip_window_params window;
  blasr_hip_sdp_params sdp;      /* as AlignIntervals passes them (:1716-1722) */
  blasr_hip_score_params score;  /* SMRTDistanceMatrix + InitializeScoreFunction (:807-809) */
  int32_t band;                  /* params.bandSize (16 once MakeSane has run with useGuidedAlign, MappingParameters.h:395-397;
                                    0 under -noUseGuidedAlign unless -bandSize was given: k then follows the drift) */
  int32_t n_candidates;          /* params.nCandidates (10) */
  int32_t max_gap;               /* params.maxGap (0) */
  int32_t max_score;             /* params.maxScore (-200) */
  float min_pct_identity;        /* params.minPctIdentity (0) */
  int32_t sort_refined;          /* params.sortRefinedAlignments (1) */
  int32_t extend_alignments;     /* params.extendAlignments (0), -extend: both ends of every SDP alignment are
                                    extended before the first filter (:1749-1940, blasr_hip_extend_* below) */
  int32_t extend_band_size;      /* params.extendBandSize (10) */
  int32_t max_extend_dropoff;    /* params.maxExtendDropoff (10) */
  int32_t refine_between_anchors_only; /* params.refineBetweenAnchorsOnly (0), -rbao: AlignIntervals aligns between the
                                    chained anchors of each interval instead of running SDPAlign over the whole window
                                    (:1354-1712, blasr_hip_align_between_anchors_* below); needs iv_chains */
  int32_t indel;                 /* params.indel (5): AffineKBandAlign's penalties in AlignSubstring (:1067-1070) */
  float sdp_bypass_threshold;    /* params.sdpBypassThreshold (1000000.0) */
  int32_t use_guided_align;      /* params.useGuidedAlign (1).  0 = -noUseGuidedAlign: RefineAlignment's other branch
                                    (alignment/Blasr.cpp:917-1046): ComputeDrift over the SDP blocks, k = band or, when band
                                    is 0, |drift| * 1.5, at least 10; PairwiseLocalAlign (:675-746) = AffineKBandAlign with
                                    k * 1.2 and the penalties of `indel`, then ComputeAlignmentStats in its non-affine form.
                                    The hits' q_pos / t_pos are 0 and their ops keep the leading gap list.  Their score is
                                    that formula with score.matrix in place (the reference's is uninitialised memory,
                                    see blasr_hip_kband_stats_device) */
  const uint32_t *iv_chains;     /* HOST: WeightedInterval::matches of the intervals, (t, q, l) triplets as
                                    blasr_hip_find_intervals_chains hands them out; interval k's are
                                    iv_chains[3 * iv_chain_off[k] .. 3 * iv_chain_off[k + 1]) */
  const int64_t *iv_chain_off;   /* HOST: n_iv + 1 entries */
} blasr_hip_map_params;

/* One alignment MapRead keeps: the T_AlignmentCandidate fields after RefineAlignment's copy-back. */
typedef struct blasr_hip_hit {
  int32_t read;                  /* index into the read table */
  int32_t interval;              /* index of the candidate interval it came from */
  int32_t t_strand, t_index;
  uint32_t t_aligned_seq_pos;    /* on the strand of the interval, after += tPos of the SDP alignment (:909) */
  uint32_t t_aligned_seq_length; /* tSeq.length (:915) */
  uint32_t t_length;             /* contig length as AlignIntervals computed it */
  uint32_t q_aligned_seq_pos;    /* after += qPos of the SDP alignment (:910) */
  uint32_t q_aligned_seq_length;
  uint32_t q_length;             /* read.length */
  int32_t q_pos, t_pos;          /* refinedAlignment.qPos / tPos */
  int32_t score;                 /* ComputeAlignmentStats score of the refined alignment */
  int32_t n_match, n_mismatch, n_ins, n_del;
  int32_t n_cells;
  float pct_similarity;
  int32_t n_ops;
  int64_t ops_off;               /* first op in the ops arena (device arena, or the host one when fetched) */
  int64_t q_off, t_off;          /* qAlignedSeq / tAlignedSeq in the sequence arena (ditto) */
} blasr_hip_hit;

/* Optional host arenas: the aligned sequences and op lists of the hits only, gathered on the device and
 * downloaded in one copy each -- what the epilogue needs.  On return seq_used / ops_used hold what was needed;
 * BLASR_HIP_ERR_CAPACITY when a capacity was too small (the hits themselves are valid). */
typedef struct blasr_hip_map_fetch {
  uint8_t *seq;
  int64_t seq_capacity, seq_used;
  uint32_t *ops;
  int64_t ops_capacity, ops_used;
} blasr_hip_map_fetch;

/* reads / read_pool / intervals: device pointers; everything else host.  hits: at most hits_capacity records,
 * *n_hits receives the number needed; read_hit_off (n_reads + 1 entries, may be NULL): the hits of read r are
 * hits[read_hit_off[r] .. read_hit_off[r+1]).  Intervals may come in any order; the candidates of a read keep the
 * order of their intervals (the order of the reference's interval set). */
int blasr_hip_map_reads_device(blasr_hip_ctx *ctx, int32_t n_reads, const blasr_hip_read *reads,
                               const uint8_t *read_pool, int32_t n_intervals, const blasr_hip_interval *intervals,
                               const blasr_hip_map_params *params, blasr_hip_hit *hits, int64_t hits_capacity,
                               int64_t *n_hits, int64_t *read_hit_off, blasr_hip_map_fetch *fetch);
/* Same with host pointers (read_pool_bytes sizes the upload). */
int blasr_hip_map_reads_host(blasr_hip_ctx *ctx, int32_t n_reads, const blasr_hip_read *reads,
                             const uint8_t *read_pool, int64_t read_pool_bytes, int32_t n_intervals,
                             const blasr_hip_interval *intervals, const blasr_hip_map_params *params,
                             blasr_hip_hit *hits, int64_t hits_capacity, int64_t *n_hits, int64_t *read_hit_off,
                             blasr_hip_map_fetch *fetch);
/* Defaults of MappingParameters.h (the values every field comment above names). */
/* RefineAlignment's k-band branch (-noUseGuidedAlign, alignment/Blasr.cpp:917-1046) on the SDP stage's device records,
 * the counterpart of blasr_hip_refine_from_sdp_device: candidate i = candidate sel[i] of the SDP batch trimmed to its SDP
 * extent (:968-981), its band from ComputeDrift over its blocks (band = params.bandSize, 0: |drift| * 1.5; at least 10;
 * times 1.2, :990-1003, :699), AffineKBandAlign with the penalties PairwiseLocalAlign derives from `indel` (:695-699),
 * ComputeAlignmentStats in its non-affine form (:740, see blasr_hip_kband_stats_device).  All pointers device memory. */
int blasr_hip_kband_from_sdp_device(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *batch,
                                    const blasr_hip_sdp_result *sdp_results, const int32_t *blocks, int32_t n_sel,
                                    const int32_t *sel, int32_t band, int32_t indel, const blasr_hip_score_params *score,
                                    blasr_hip_refine_result *results, uint32_t *ops, int64_t ops_capacity,
                                    int64_t *ops_used);
int blasr_hip_map_default_params(blasr_hip_map_params *params);
/* The switch -noUseGuidedAlign as MappingParameters::MakeSane leaves the parameters (MappingParameters.h:533-545: bandSize
 * becomes 16 only WITH useGuidedAlign): use_guided_align = 0 and, unless the caller has set a band of its own (band_given
 * != 0, the reference's -bandSize), band = 0 -- k then follows the SDP alignment's drift (alignment/Blasr.cpp:991-1001).
 * Call it on a struct blasr_hip_map_default_params filled; setting use_guided_align = 0 by hand keeps band = 16. */
int blasr_hip_map_set_no_guided_align(blasr_hip_map_params *params, int32_t band_given);
/* Wall time (ms) of the stages of the last map call: [0] windows, [1] SDP, [2] sort + first filter (host),
 * [3] refine, [4] sort + the two filters (host), [5] fetch; counters: [0] candidate intervals, [1] candidates
 * refined, [2] reads with an alignment, [3] bases of those reads, [4] band cells of the refined candidates,
 * [5] hits. */
int blasr_hip_map_last_timing(const blasr_hip_ctx *ctx, float ms[6], int64_t counters[6]);
/* Without fetch arenas a hit's q_off / t_off / ops_off index the context's own device arenas: these
 * (valid until the next map call on the context). */
int blasr_hip_map_device_arenas(blasr_hip_ctx *ctx, const uint8_t **seq, const uint32_t **ops);
/* The genome is already in device memory and stays the caller's (not copied, not freed). */
int blasr_hip_adopt_genome_device(blasr_hip_ctx *ctx, const uint8_t *ascii, int64_t length,
                                  const uint32_t *contig_starts_host, int32_t n_contigs);

/* ------------------------------------------------------------------------------------------------
 * a-11: AffineKBandAlign, Global (algorithms/alignment/AffineKBandAlign.h:12-401) -- the refiner
 * PairwiseLocalAlign runs under -noUseGuidedAlign (alignment/Blasr.cpp:675-746): fixed band of
 * 2k+1 columns, affine insertion + affine homopolymer insertion + linear deletion.
 * ------------------------------------------------------------------------------------------------ */
typedef struct blasr_hip_kband_params {
  int32_t matrix[25];     /* SMRTDistanceMatrix (Blasr.cpp:695) */
  int32_t hp_ins_open;    /* params.indel + 2 */
  int32_t hp_ins_extend;  /* params.indel - 3 */
  int32_t ins_open;       /* params.indel + 2 */
  int32_t ins_extend;     /* params.indel - 1 */
  int32_t del;            /* params.indel */
} blasr_hip_kband_params;

/* Candidate i: query / target as for the guided refiner (trimmed to the SDP extent, Blasr.cpp:968-981),
 * k[i] = the band half-width handed to AffineKBandAlign (k * 1.2 of :699), any k >= 1: bands up to 1088 columns
 * keep their rows in LDS, wider ones (-noUseGuidedAlign gives k ~ 1.3 k for 10 kb reads and ~ 2.7 k for 30 kb ones)
 * are swept as strips of the (q, t) matrix.  (rows + columns + 2) * the largest penalty must stay below 2^25, else the
 * candidate gets BLASR_HIP_TASK_BAD_GUIDE. */
typedef struct blasr_hip_kband_batch {
  int32_t n;
  const uint8_t *seq;
  const int64_t *q_off;
  const int32_t *q_len;
  const int64_t *t_off;
  const int32_t *t_len;
  const int32_t *k;
} blasr_hip_kband_batch;

/* Results reuse blasr_hip_refine_result: dp_score = the function's return value, n_cells = (qLen+1)(2k+1),
 * q_pos = t_pos = 0, ops as for the guided refiner but WITHOUT RemoveAlignmentPrefixGaps: leading gap ops form
 * Alignment.gaps[0]; the op list ends with the last block (ArrowPathToAlignment empties the gap list behind it,
 * Alignment.h:190-254).  The statistics fields are 0 until blasr_hip_kband_stats_device fills them. */
int blasr_hip_kband_device(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *batch,
                           const blasr_hip_kband_params *params, blasr_hip_refine_result *results, uint32_t *ops,
                           int64_t ops_capacity, int64_t *ops_used);
int blasr_hip_kband_host(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *batch,
                         const blasr_hip_kband_params *params, blasr_hip_refine_result *results, uint32_t *ops,
                         int64_t ops_capacity, int64_t *ops_used);

/* ComputeAlignmentStats (algorithms/alignment/AlignmentUtils.h:535-584) on the results of blasr_hip_kband_device, as
 * PairwiseLocalAlign calls it (alignment/Blasr.cpp:740; use_affine = 0 there): score, n_match, n_mismatch, n_ins,
 * n_del of every result with status 0.  `ops` is the arena blasr_hip_kband_device compacted the op lists into
 * (device memory, as `results`).  NB the reference makes this call with a score function whose 5x5 matrix was never
 * initialised (Blasr.cpp:689-690; contrast :807-809): the score it prints under -noUseGuidedAlign is uninitialised
 * memory.  Here `score` carries the matrix; the counts do not depend on it. */
int blasr_hip_kband_stats_device(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *batch,
                                 const blasr_hip_score_params *score, int32_t use_affine,
                                 blasr_hip_refine_result *results, const uint32_t *ops);

/* KBandAlign with a DistanceMatrixScoreFunction (algorithms/alignment/KBandAlign.h:74-403): the banded,
 * linear-gap aligner behind -doGlobalAlignment (alignment/Blasr.cpp:819-823, Global) and behind
 * PairwiseLocalAlign for reads with quality values (Blasr.cpp:714-730, Fit).  ins / del are the function's
 * boundary penalties, fn_ins / fn_del what scoreFn.Insertion / Deletion return inside the recurrence;
 * scoreFn.Match = matrix[ThreeBit[ref]][ThreeBit[query]] (DistanceMatrixScoreFunction.h:43-45). */
typedef struct blasr_hip_kband_linear_params {
  int32_t matrix[25];
  int32_t ins, del;
  int32_t fn_ins, fn_del;
  int32_t align_type;  /* AlignmentType (AlignmentUtils.h:12-30): Global 1, QueryFit 2, TargetFit 3, Fit 7 */
} blasr_hip_kband_linear_params;

/* Results as for blasr_hip_kband_*: q_pos / t_pos = alignment.qPos / tPos (KBandAlign.h:388-398),
 * n_cells = alignment.nCells, dp_score = the return value. */
int blasr_hip_kband_linear_device(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *batch,
                                  const blasr_hip_kband_linear_params *params, blasr_hip_refine_result *results,
                                  uint32_t *ops, int64_t ops_capacity, int64_t *ops_used);
int blasr_hip_kband_linear_host(blasr_hip_ctx *ctx, const blasr_hip_kband_batch *batch,
                                const blasr_hip_kband_linear_params *params, blasr_hip_refine_result *results,
                                uint32_t *ops, int64_t ops_capacity, int64_t *ops_used);

/* ------------------------------------------------------------------------------------------------
 * next-3, first piece: ExtendAlignmentForward / ExtendAlignmentReverse (algorithms/alignment/ExtendAlign.h:115-445),
 * the -extend step AlignIntervals runs on both ends of every SDP alignment (alignment/Blasr.cpp:1749-1940): band of
 * 2k+1 diagonals, linear gaps, rows added until max_n_drops rows have raised the row minimum; the alignment ends at
 * the first cell holding the global minimum.
 * ------------------------------------------------------------------------------------------------ */
typedef struct blasr_hip_extend_params {
  int32_t matrix[25];          /* distScoreFn.scoreMatrix, [target][query] (DistanceMatrixScoreFunction.h:43-45) */
  int32_t ins, del;            /* distScoreFn.ins / del = params.insertion / deletion (Blasr.cpp:3666-3667) */
  int32_t band;                /* k = params.extendBandSize (10); 0..31 here */
  int32_t min_extend_n_bases;  /* 1 (Blasr.cpp:1833) */
  int32_t max_n_drops;         /* params.maxExtendDropoff (10) */
} blasr_hip_extend_params;
int blasr_hip_extend_default_params(blasr_hip_extend_params *params);

/* Task i mirrors one call: querySeq = seq[q_off[i] .. +q_len[i]), refSeq = seq[t_off[i] .. +t_len[i]),
 * queryPos = q_pos[i], refPos = t_pos[i].  reverse[i] == 0: ExtendAlignmentForward aligns query[queryPos..) to
 * target[refPos..); != 0: ExtendAlignmentReverse aligns the bases BEFORE queryPos / refPos, walking left (the
 * reference reads them as query[queryPos - q], so q_pos <= q_len and t_pos <= t_len are required; otherwise the
 * task gets BLASR_HIP_TASK_BAD_GUIDE).  Device pointers for _device, host pointers for _host. */
typedef struct blasr_hip_extend_batch {
  int32_t n;
  const uint8_t *seq;
  const int64_t *q_off;
  const int32_t *q_len;
  const int64_t *t_off;
  const int32_t *t_len;
  const int32_t *q_pos;
  const int32_t *t_pos;
  const uint8_t *reverse;
} blasr_hip_extend_batch;

/* Results reuse blasr_hip_refine_result: dp_score = the function's return value (0 when a sequence is shorter
 * than min_extend_n_bases, INT_MAX when no row was scored), q_pos / t_pos = alignment.qPos / tPos (Forward: queryPos /
 * refPos, ExtendAlign.h:401-402; Reverse: the leftmost aligned base, :97-103), n_cells = cells scored, ops = the
 * ordered arrow path run-length encoded (Alignment::ArrowPathToAlignment rebuilds blocks / gaps from it and drops
 * a trailing gap run, Alignment.h:190-254).  The statistics fields are 0. */
int blasr_hip_extend_device(blasr_hip_ctx *ctx, const blasr_hip_extend_batch *batch,
                            const blasr_hip_extend_params *params, blasr_hip_refine_result *results, uint32_t *ops,
                            int64_t ops_capacity, int64_t *ops_used);
int blasr_hip_extend_host(blasr_hip_ctx *ctx, const blasr_hip_extend_batch *batch,
                          const blasr_hip_extend_params *params, blasr_hip_refine_result *results, uint32_t *ops,
                          int64_t ops_capacity, int64_t *ops_used);

/* ------------------------------------------------------------------------------------------------
 * next-3, second piece: -rbao ("refine between anchors only", params.refineBetweenAnchorsOnly), the branch of
 * AlignIntervals that replaces the ONE SDPAlign over the candidate's window by small alignments between consecutive
 * chained anchors of the interval (alignment/Blasr.cpp:1354-1712).  Per candidate:
 *   the anchors made relative to the window, flipped for a Reverse-strand window (:1385-1408); overlapping anchors
 *   dropped (:1414-1439); runs of anchors cut where the diagonal moves by 25 or more and the best chain of runs kept
 *   (:1443-1564, GAP = -0.1 a base of drift) -- host code, a few hundred anchors;
 *   one AlignSubstring (:1050-1122) per gap between consecutive anchors (both gaps > 0, ratio within 1000, < 100000):
 *   AffineKBandAlign, Global, k = bandSize, when a side is shorter than bandSize and the lengths are within 2x
 *   (blasr_hip_kband_device), else SDPAlign (Global, extendFront, indelRate 0.25, smithWatermanAlignType Global; word
 *   size 7 when window and subread are both under 1000 bases) + the two pinned end blocks + RefineAlignment on the
 *   gap (blasr_hip_sdp_device + blasr_hip_refine_device) -- device batches over the gaps of all candidates;
 *   AppendSubseqAlignment (:1124-1147), the last anchor, qPos / tPos taken out, abutting blocks merged (:1640-1705),
 *   ComputeAlignmentStats of an alignment without gap lists (:1709) -- a kernel.
 * What comes back is what blasr_hip_sdp_* returns for the default branch (results[i].score = 0: alignScore is never
 * set here), so the rest of MapRead (filters, RefineAlignments, ...) runs unchanged behind it.  An interval whose
 * anchors cover sdp_bypass_threshold of the subread would take the -sdpbypass branch (:1728-1739), which is not
 * offered: BLASR_HIP_ERR_UNSUPPORTED.
 * ------------------------------------------------------------------------------------------------ */
typedef struct blasr_hip_rbao_params {
  blasr_hip_sdp_params sdp;     /* word_size = sdpTupleSize, sdp_ins, sdp_del, detailed, prefix_length, recurse,
                                   recurse_over, max_per_position as AlignIntervals has them; indel_rate, align_type,
                                   extend_front and sw_align_type are set by the function (:1080-1089) */
  blasr_hip_score_params score; /* distScoreFn of AlignIntervals (:1176-1178), and of RefineAlignment (:807-809) */
  int32_t band;                 /* params.bandSize (16) */
  int32_t indel;                /* params.indel (5) */
  float sdp_bypass_threshold;   /* params.sdpBypassThreshold (1000000.0) */
  int32_t reserved;
} blasr_hip_rbao_params;

/* batch, windows, results, blocks: DEVICE memory, laid out as for blasr_hip_windows_device / blasr_hip_sdp_device
 * (candidate i: query = the subread, target = the window; windows[i] names the window's place and strand).
 * chains / chain_off: HOST memory, candidate i's chained anchors as (t, q, l) triplets
 * chains[3 * chain_off[i] .. 3 * chain_off[i + 1]) in the coordinates FindMaxIncreasingInterval leaves them in
 * (blasr_hip_find_intervals_chains).  genome_length: DNASequence::length of the whole genome (MakeRCCoordinate). */
int blasr_hip_align_between_anchors_device(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *batch,
                                           const blasr_hip_window *windows, uint32_t genome_length,
                                           const uint32_t *chains, const int64_t *chain_off,
                                           const blasr_hip_rbao_params *params, blasr_hip_sdp_result *results,
                                           int32_t *blocks, int64_t blocks_capacity, int64_t *blocks_used);
/* The same with everything in HOST memory (seq_bytes sizes the upload of batch->seq). */
int blasr_hip_align_between_anchors_host(blasr_hip_ctx *ctx, const blasr_hip_sdp_batch *batch, int64_t seq_bytes,
                                         const blasr_hip_window *windows, uint32_t genome_length,
                                         const uint32_t *chains, const int64_t *chain_off,
                                         const blasr_hip_rbao_params *params, blasr_hip_sdp_result *results,
                                         int32_t *blocks, int64_t blocks_capacity, int64_t *blocks_used);

/* ------------------------------------------------------------------------------------------------
 * Output epilogue, first pieces: host functions, no GPU work.
 * ------------------------------------------------------------------------------------------------ */
/* AssignRefContigLocation (alignment/Blasr.cpp:2183-2220): the contig of an alignment whose target window starts
 * at t_aligned_seq_pos of the concatenated genome (of its reverse complement when t_strand = 1), and the window's
 * position inside that contig.  seq_start_pos: SequenceIndexDatabase::seqStartPos, n_seq_pos entries (contigs + 1;
 * SearchForIndex, SequenceIndexDatabase.h:108-118). */
int blasr_hip_ref_contig_location(uint32_t t_aligned_seq_pos, uint32_t t_aligned_seq_length, int32_t t_strand,
                                  const uint32_t *seq_start_pos, int32_t n_seq_pos, uint32_t genome_length,
                                  int32_t *contig, uint32_t *contig_pos);

/* SAMPrinter::Clipping (printers/SAMPrinter.h:26) */
enum { BLASR_HIP_CLIP_HARD = 0, BLASR_HIP_CLIP_SOFT = 1, BLASR_HIP_CLIP_SUBREAD = 2, BLASR_HIP_CLIP_NONE = 3 };
typedef struct blasr_hip_cigar_params {
  int32_t clipping;   /* BLASR_HIP_CLIP_* */
  int32_t t_strand;   /* 1: the op list is reversed at the end (:408-411) */
  uint32_t read_length, low_quality_prefix, low_quality_suffix, subread_start, subread_end; /* SMRTSequence fields */
  uint32_t q_align_start, q_align_end; /* AlignmentCandidate::QAlignStart() / QAlignEnd() (AlignmentCandidate.h:150-156) */
} blasr_hip_cigar_params;
/* CreateCIGARString (printers/SAMPrinter.h:346-416) with CreateNoClippingCigarOps (:206-298), AddGaps (:122-136),
 * AddUngappedOperations / AddUnmatchedOperations (:138-201), SetSoftClip / SetHardClip (:300-335).
 * The alignment is given EITHER as the refine stage's op list (ops, n_ops: (length << 2) | M 0 / I 1 / D 2 -- an
 * alignment with gap lists) OR as blocks (q, t, length triplets relative to q_pos / t_pos -- an alignment without
 * gap lists, e.g. straight from SDPAlign); q_pos / t_pos = alignment.qPos / tPos inside the aligned sequences.
 * Matches are split into = and X by comparing the raw characters.  clips (may be NULL) receives
 * {prefixSoftClip, suffixSoftClip, prefixHardClip, suffixHardClip}; *cigar_len the length without the NUL. */
int blasr_hip_cigar(const uint8_t *q_aligned_seq, const uint8_t *t_aligned_seq, int32_t q_pos, int32_t t_pos,
                    const uint32_t *ops, int64_t n_ops, const int32_t *blocks, int64_t n_blocks,
                    const blasr_hip_cigar_params *params, char *cigar, int64_t cigar_capacity, int64_t *cigar_len,
                    uint32_t clips[4]);

/* IntervalAlignmentPrinter::Print (printers/IntervalAlignmentPrinter.h:9-38): the 13 columns of -m 4.
 * q_align_start / q_align_end = QAlignStart() / QAlignEnd(); t_align_start = TAlignStart(); t_align_end =
 * tAlignedSeqPos + tPos + last block's tPos + length (:27-30). */
typedef struct blasr_hip_m4_fields {
  const char *q_name, *t_name;
  int32_t score;
  float pct_similarity;
  int32_t q_strand;
  uint32_t q_align_start, q_align_end, q_length;
  int32_t t_strand;
  uint32_t t_align_start, t_align_end, t_length;
  int32_t map_qv;
} blasr_hip_m4_fields;
int blasr_hip_m4_line(const blasr_hip_m4_fields *fields, char *line, int64_t line_capacity, int64_t *line_len);

/* The gates of PrintAlignment (alignment/Blasr.cpp:2957-2978) an alignment passes before any printer sees it: score
 * <= maxScore, pctSimilarity >= minPctIdentity, tAlignedSeq.length >= minAlignLength, mapQV >= minMapQV.  PrintAlignments
 * (:3085-3110) offers it the first nBest alignments of a read; one that is held back is not replaced.  Returns 1
 * (printed) or 0. */
typedef struct blasr_hip_print_params {
  int32_t max_score;        /* params.maxScore (-200) */
  float min_pct_identity;   /* params.minPctIdentity (0) */
  int32_t min_align_length; /* params.minAlignLength (0) */
  int32_t min_map_qv;       /* params.minMapQV (0) */
} blasr_hip_print_params;
int blasr_hip_alignment_is_printed(int32_t score, float pct_similarity, uint32_t t_aligned_seq_length, int32_t map_qv,
                                   const blasr_hip_print_params *params);

/* StoreMapQVs (alignment/Blasr.cpp:2761-2950) with ComputeAlignmentScore on gap lists (AlignmentUtils.h:127-170),
 * PartitionOverlappingAlignments / AlignmentsOverlap (:2679-2759), SumMismatches (:2647-2671), LogSumOfTwo
 * (utils/LogUtils.h:11-41), Phred (utils/PhredUtils.h:10-13): the mapping quality of every alignment of one read,
 * from the alignments that overlap it on the read.  Alignments in the order of alignmentPtrs. */
typedef struct blasr_hip_mapqv_alignment {
  const uint8_t *q_aligned_seq, *t_aligned_seq; /* alignment.qAlignedSeq / tAlignedSeq */
  const uint32_t *ops;                          /* the refine stage's op list = blocks + gap lists */
  int64_t n_ops;
  int32_t q_pos, t_pos;                         /* alignment.qPos / tPos */
  uint32_t q_align_start, q_align_end;          /* QAlignStart() / QAlignEnd() */
  int32_t q_strand;
  uint32_t q_length;
  int32_t n_match;
  int32_t reserved;
} blasr_hip_mapqv_alignment;
typedef struct blasr_hip_mapqv_params {
  int32_t matrix[25];              /* SMRTLogProbMatrix (ScoreMatrices.h:28-34), [target][query] as Match() indexes it */
  int32_t ins, del, affine_open, affine_extend; /* MappingParameters::InitializeScoreFunction (:587-592) */
  int32_t affine_align;            /* params.affineAlign */
  float min_fraction_overlapping;  /* params.minFractionToBeConsideredOverlapping (0.75) */
} blasr_hip_mapqv_params;
/* substitution_qv: read.substitutionQV (one byte per read base) or NULL; prob_score (may be NULL) receives the
 * alignments' probScore after the adjustment of :2851-2873. */
int blasr_hip_map_qvs(const blasr_hip_mapqv_alignment *alignments, int32_t n, const blasr_hip_mapqv_params *params,
                      const uint8_t *substitution_qv, int32_t *map_qv, float *prob_score);

/* SAMOutput::PrintAlignment (printers/SAMPrinter.h:418-548, passthrough = false, no supplemental QVs) with
 * BuildFlag (:29-50) and SetAlignedSequence (:58-105): one SAM record, newline included.  cigar / clips are what
 * blasr_hip_cigar returned for the same alignment and clipping. */
typedef struct blasr_hip_sam_fields {
  const char *q_name, *t_name, *read_group_id, *cigar;
  const uint8_t *read_seq;   /* the read, forward strand, read_length bases */
  const uint8_t *read_qual;  /* read.qual.data or NULL */
  uint32_t clips[4];         /* {prefixSoftClip, suffixSoftClip, prefixHardClip, suffixHardClip} */
  int32_t clipping;          /* BLASR_HIP_CLIP_* */
  int32_t q_strand, t_strand, is_primary;
  uint32_t read_length, low_quality_prefix, low_quality_suffix, subread_start, subread_end;
  uint32_t q_align_start, q_align_end; /* QAlignStart() / QAlignEnd() */
  uint32_t t_align_start;    /* TAlignStart() + blocks[0].tPos (:437) */
  uint32_t t_end;            /* TEnd(): last block's tPos + length */
  uint32_t t_length, q_length;
  uint32_t q_aligned_seq_pos, q_aligned_seq_length; /* FI and XL */
  int32_t map_qv, score, edit_dist; /* edit_dist = AlignmentContext::editDist */
  int32_t hole_number, qs, qe, num_passes, subread_context; /* SMRTSequence fields */
  float read_quality, snr[4];
} blasr_hip_sam_fields;
int blasr_hip_sam_line(const blasr_hip_sam_fields *fields, char *line, int64_t line_capacity, int64_t *line_len);
/* SAMOutput::PrintUnalignedRead (printers/SAMPrinter.h:578-664): the record -sam prints for a (sub)read that has no
 * alignment (alignment/Blasr.cpp:3848-3866): flag 4, the subread's bases, the same tag block.  Reads q_name,
 * read_group_id, read_seq, read_length, subread_start / subread_end, clipping, qs, qe, hole_number, read_quality,
 * num_passes, subread_context, snr; *line_len = 0 where the reference prints nothing (empty subread). */
int blasr_hip_sam_unaligned_line(const blasr_hip_sam_fields *fields, char *line, int64_t line_capacity,
                                 int64_t *line_len);

/* Alignment::OrderGapsByType (datastructures/alignment/Alignment.h:255-317), which MapRead applies under -forPicard
 * (alignment/Blasr.cpp:2621-2626), on an op list: where two blocks are separated by more than one gap, the gaps
 * are condensed to one of the longer kind, shortened by the bases of the other kind, which extend the block
 * before.  out may alias nothing; *n_out receives the new length (<= n_ops). */
int blasr_hip_order_gaps_by_type(const uint32_t *ops, int64_t n_ops, uint32_t *out, int64_t out_capacity,
                                 int64_t *n_out);

/* ------------------------------------------------------------------------------------------------
 * next-4: the reference's on-disk inputs.  Host functions, no GPU work.
 * ------------------------------------------------------------------------------------------------ */
typedef struct blasr_hip_fasta blasr_hip_fasta;
/* FASTAReader::ReadAllSequences (FASTAReader.h:166-216, :255-271): every record of a FASTA file / text -- the title
 * (the whole line after '>') and the sequence (the whitespace-separated words of the lines up to the next '>'). */
int blasr_hip_fasta_read(const char *path, blasr_hip_fasta **out);
int blasr_hip_fasta_parse(const char *text, int64_t length, blasr_hip_fasta **out);
int blasr_hip_fasta_free(blasr_hip_fasta *fasta);
/* FASTQ input as the reference's FASTQReader::GetNext(FASTQSequence&) reads it (common/FASTQReader.h:27-66, dispatched
 * by files/ReaderAgglomerate.h:250,279) -- which in this fork is not a FASTQ parser: after a title line it looks for
 * the NEXT '@' (FASTAReader::GetNext(seq, '@'), :41), so a plain four-line file yields every second record, and an '@'
 * among the quality characters starts a title there.  Restated as it behaves, because the binary's lines on FASTQ
 * input are these.  The records come back through blasr_hip_fasta_count / _record / _name, their quality characters
 * (one per base) through blasr_hip_fastq_quality.  Where the reference leaves defined ground -- a record without its
 * '+' line (it exits), fewer quality characters than bases (it copies uninitialised memory; ERR_UNSUPPORTED) -- the
 * call returns an error with the records read so far in *out and the reason in blasr_hip_fasta_error. */
int blasr_hip_fastq_read(const char *path, blasr_hip_fasta **out);
int blasr_hip_fastq_parse(const char *text, int64_t length, blasr_hip_fasta **out);
int blasr_hip_fastq_quality(const blasr_hip_fasta *fasta, int64_t i, const uint8_t **qual, int64_t *qual_length);
/* Reads from a SAM file (a `.sam` read file: files/ReaderAgglomerate.h:308-318, SAMReader::GetNextAlignment,
 * SAMAlignment::StoreValues, ConvertSAMToSMRTSequence): the leading '@' lines are the header, every later line a record
 * whose first eleven whitespace-separated words are QNAME ... SEQ QUAL; title = QNAME, bases = SEQ, qualities = QUAL's
 * characters (blasr_hip_fastq_quality; none when "*"), the subread the whole read.  A line the reference exits on
 * (fewer than eleven words, a number that does not parse): BLASR_HIP_ERR_INVALID, the records before it stay readable. */
int blasr_hip_sam_reads_read(const char *path, blasr_hip_fasta **out);
int blasr_hip_sam_reads_parse(const char *text, int64_t length, blasr_hip_fasta **out);
const char *blasr_hip_fasta_error(const blasr_hip_fasta *fasta);
int64_t blasr_hip_fasta_count(const blasr_hip_fasta *fasta);
/* Record i; the pointers stay valid until blasr_hip_fasta_free.  Any output may be NULL. */
int blasr_hip_fasta_record(const blasr_hip_fasta *fasta, int64_t i, const char **title, int64_t *title_length,
                           const uint8_t **seq, int64_t *seq_length);
/* FASTASequence::GetName (FASTASequence.h:36-52): the title up to its first blank; returns its length. */
int64_t blasr_hip_fasta_name(const blasr_hip_fasta *fasta, int64_t i, char *name, int64_t capacity);
/* FASTAReader::ReadAllSequencesIntoOne (:113-146) as Blasr.cpp reads a genome (:4329, :4354): the records back to
 * back, one 'N' after each, upper-cased when to_upper is set; seq_start_pos (count + 1 entries, may be NULL) receives
 * SequenceIndexDatabase::seqStartPos.  Returns the length (with genome == NULL: only the length and the table), < 0 on
 * error. */
int64_t blasr_hip_fasta_concatenate(const blasr_hip_fasta *fasta, int to_upper, uint8_t *genome, int64_t capacity,
                                    uint32_t *seq_start_pos);

typedef struct blasr_hip_sa blasr_hip_sa;
/* SuffixArray::Read (datastructures/suffixarray/SuffixArray.h:608-630): a .sa file as sawriter writes it (:505-532):
 * magic 0xacac0001, the component list, the index array, the lookup table (start / end of every prefix's range). */
int blasr_hip_sa_read(const char *path, blasr_hip_sa **out);
int blasr_hip_sa_free(blasr_hip_sa *sa);
int blasr_hip_sa_arrays(const blasr_hip_sa *sa, const uint32_t **index, int64_t *length, const uint32_t **start_pos_table,
                        const uint32_t **end_pos_table, int64_t *lookup_table_length, int32_t *lookup_prefix_length);

/* ------------------------------------------------------------------------------------------------
 * next-1, device half: anchoring a read in the suffix array of the genome.
 * Replaces MapReadToGenome (common/algorithms/anchoring/MapBySuffixArray.h:217-311), with
 * LocateAnchorBoundsInSuffixArray (:24-215) and SuffixArray::StoreLCPBounds / SearchLeftBound /
 * SearchRightBound (common/datastructures/suffixarray/SuffixArray.h:928-1070, :733-816) underneath, as
 * MapRead calls it for both strands of a read (alignment/Blasr.cpp:2279-2297), and SortMatchPosList
 * (:2339-2340; datastructures/anchoring/MatchPos.h:120-131,166-168) on the two lists.  One lane per read
 * position walks the suffix array; the genome, the index and the lookup table stay resident on the device.
 * FindMaxIncreasingInterval (the host half of next-1) consumes the sorted lists.
 * ------------------------------------------------------------------------------------------------ */
/* AnchorParameters (datastructures/anchoring/AnchorParameters.h:9-42) as Blasr.cpp fills it (:398, :2273), and
 * the arguments of the two calls. */
typedef struct blasr_hip_anchor_params {
  uint32_t min_match_length;         /* anchorParameters.minMatchLength = params.minMatchLength (14, MappingParameters.h:244) */
  uint32_t min_prefix_match_length;  /* MapReadToGenome's minPrefixMatchLength = params.lookupTableLength (8); match
                                        lengths are counted from it even when the table's prefix differs */
  int32_t expand;                    /* anchorParameters.expand, 0..7: the MapRead iteration (:2273) */
  int32_t max_anchors_per_position;  /* 1000 */
  int32_t advance_exact_matches;     /* 0 */
  int32_t max_lcp_length;            /* 0: no limit */
  int32_t use_lookup_table;          /* 1 */
  int32_t stop_mapping_once_unique;  /* 1 (MappingParameters.h:309) */
  int32_t forward_only;              /* params.forwardOnly (0): no reverse-complement list */
  int32_t sorted;                    /* 1: SortMatchPosList applied (t, then q); 0: the order MapReadToGenome emits */
} blasr_hip_anchor_params;
/* removeEncompassedMatches is not offered: the reference indexes its per-position arrays with read.length there
 * (MapBySuffixArray.h:252-256), past their end. */

/* MatchPos (datastructures/anchoring/MatchPos.h:11-40): t genome position, q read position on the strand of the
 * list, l match length */
typedef struct blasr_hip_anchor {
  uint32_t t, q, l;
} blasr_hip_anchor;

int blasr_hip_anchor_default_params(blasr_hip_anchor_params *params);
/* The suffix array of the uploaded genome (host arrays as blasr_hip_sa_arrays returns them; length must equal the
 * genome's).  Replaces the arrays SuffixArray::Read leaves in memory. */
int blasr_hip_upload_suffix_array(blasr_hip_ctx *ctx, const uint32_t *index, int64_t length,
                                  const uint32_t *start_pos_table, const uint32_t *end_pos_table,
                                  int64_t lookup_table_length, int32_t lookup_prefix_length);
/* A second context on the same device that anchors in the genome and the index `from` holds, without a copy of either
 * (one context per mapping thread next to its MappingBuffers, alignment/Blasr.cpp:3238, all reading the one SuffixArray
 * of main(), :4360-4440): genome, contig table, index array, lookup table and the search's companion array are shared,
 * streams, workspace and result buffers are the context's own.  `from` must outlive the borrower, and it must be IDLE
 * during this call (no call on it from another thread: the companion arrays of its index are made on its stream here when
 * it has not anchored yet).  After `from` has uploaded / built another genome or index the borrower must borrow again: its
 * anchor calls fail with BLASR_HIP_ERR_INVALID until it has (the owner counts such changes; addresses are not compared). */
int blasr_hip_borrow_index(blasr_hip_ctx *ctx, blasr_hip_ctx *from);
/* reads: HOST table; read_pool: device (forward strands, ASCII); anchors: device, at most capacity records;
 * *n_anchors receives the number needed (BLASR_HIP_ERR_CAPACITY when it exceeds capacity; nothing is valid then).
 * list_off (host, 2 * n_reads + 1 entries): list 2r is read r's matchPosList, list 2r+1 its rcMatchPosList (q on the
 * reverse-complemented read, as the reference holds it).  BLASR_HIP_ERR_INVALID also where the reference's own
 * asserts (MapBySuffixArray.h:283, :301-302) would fire. */
int blasr_hip_anchor_reads_device(blasr_hip_ctx *ctx, int32_t n_reads, const blasr_hip_read *reads,
                                  const uint8_t *read_pool, const blasr_hip_anchor_params *params,
                                  blasr_hip_anchor *anchors, int64_t capacity, int64_t *n_anchors, int64_t *list_off);
/* Same with a host read pool and a host anchor array. */
int blasr_hip_anchor_reads_host(blasr_hip_ctx *ctx, int32_t n_reads, const blasr_hip_read *reads,
                                const uint8_t *read_pool, int64_t read_pool_bytes,
                                const blasr_hip_anchor_params *params, blasr_hip_anchor *anchors, int64_t capacity,
                                int64_t *n_anchors, int64_t *list_off);
/* Device time (ms) of the last anchor call: [0] suffix-array search, [1] count scan + emission, [2] sort, [3] whole
 * call (wall); counters: [0] read positions searched, [1] anchors, [2] pieces, [3] probes of the index array made by
 * the search (when counted: BLASR_HIP_ANCHOR_COUNT_PROBES=1, else 0). */
int blasr_hip_anchor_last_timing(const blasr_hip_ctx *ctx, float ms[4], int64_t counters[4]);

/* ------------------------------------------------------------------------------------------------
 * next-1, host half: from the sorted anchor lists to the candidate intervals of a read.  Host code, no
 * GPU work (1 % of the anchoring time).  Replaces, per read, MapRead between SortMatchPosList and
 * AlignIntervals (alignment/Blasr.cpp:2342-2500) with the reference's default switches (pValueType 0,
 * globalChainType 0, no overlap / noSelf / maxAnchorGap / limsAlign / piecewiseMatch):
 * RemoveOverlappingAnchors on both lists (algorithms/anchoring/FindMaxInterval.h:282-315), then
 * FindMaxIncreasingInterval(Forward, ...) and (Reverse, ...) (:661-843: StoreLargestIntervals :394-640,
 * GlobalChain algorithms/anchoring/GlobalChain.h:101-207, the p-value of LISPValue.h:84-172, the size
 * weight) into ONE WeightedIntervalSet of nCandidates (datastructures/anchoring/WeightedInterval.h:189-268),
 * optionally RemoveContained (:133-181).
 * ------------------------------------------------------------------------------------------------ */
typedef struct blasr_hip_interval_params {
  int32_t n_candidates;               /* params.nCandidates (10): the interval set's maxSize */
  int32_t min_interval;               /* params.minInterval (100) */
  float indel_rate;                   /* params.indelRate (0.30): intervalLength = subread * (1 + indelRate), at most +50000 */
  float max_p_value;                  /* log(0.5), set in MapRead (Blasr.cpp:2371) */
  int32_t remove_contained_intervals; /* params.removeContainedIntervals (0) */
  int32_t threads;                    /* host threads over the reads; 0 = up to 16 (32 for the device route's host end) */
} blasr_hip_interval_params;
int blasr_hip_interval_default_params(blasr_hip_interval_params *params);
/* TupleCountTable::AddSequenceTupleCountsLR (datastructures/tuplelists/TupleCountTable.h:63-73) as Blasr.cpp:4479-4482
 * builds it when no .ctab is given (tuple_size = params.lookupTableLength, 8): count_table has 4^tuple_size entries. */
int blasr_hip_count_tuples(const uint8_t *genome, int64_t genome_length, int32_t tuple_size, int32_t *count_table);
/* The same table counted on the device from the uploaded genome (count_table: host, 4^tuple_size entries). */
int blasr_hip_count_tuples_device(blasr_hip_ctx *ctx, int32_t tuple_size, int32_t *count_table);
/* anchors / list_off: what blasr_hip_anchor_reads_* returned with sorted = 1 (host arrays).  seq_start_pos:
 * SequenceIndexDatabase::seqStartPos (n_seq_pos = contigs + 1 entries).  intervals (at most capacity; *n_intervals
 * receives the number needed, BLASR_HIP_ERR_CAPACITY beyond): read r's are intervals[read_interval_off[r] ..
 * read_interval_off[r+1]) in the order AlignIntervals walks the set (p-value, then start); p_values (may be NULL)
 * receives their p-values.
 * stale (2 * n_reads anchors, may be NULL = zeros): StoreLargestIntervals reads the element one past the end of its
 * list (FindMaxInterval.h:493, :581, :617).  After RemoveOverlappingAnchors removed anything that element is the
 * list's own old entry at that index -- modelled here; when nothing was removed the reference sees whatever its
 * reused buffer held, and `stale` is that element (list 2r, 2r+1 of read r). */
int blasr_hip_find_intervals(const uint8_t *genome, int64_t genome_length, const uint32_t *seq_start_pos,
                             int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size, int32_t n_reads,
                             const blasr_hip_read *reads, const blasr_hip_anchor *anchors, const int64_t *list_off,
                             const blasr_hip_anchor *stale, const blasr_hip_interval_params *params,
                             blasr_hip_interval *intervals, float *p_values, int64_t capacity, int64_t *n_intervals,
                             int64_t *read_interval_off);

/* The same, and the chained anchors of every interval with it (WeightedInterval::matches: the `lis` vector of
 * FindMaxIncreasingInterval, FindMaxInterval.h:776-833, which the -rbao branch of AlignIntervals reads): interval k's
 * chain is chains[3 * chain_off[k] .. 3 * chain_off[k + 1]) as (t, q, l) triplets in the order of the chain; chain_off
 * has capacity + 1 entries; *chains_used receives the number of triplets needed (BLASR_HIP_ERR_CAPACITY beyond
 * chains_capacity). */
int blasr_hip_find_intervals_chains(const uint8_t *genome, int64_t genome_length, const uint32_t *seq_start_pos,
                                    int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size,
                                    int32_t n_reads, const blasr_hip_read *reads, const blasr_hip_anchor *anchors,
                                    const int64_t *list_off, const blasr_hip_anchor *stale,
                                    const blasr_hip_interval_params *params, blasr_hip_interval *intervals,
                                    float *p_values, int64_t capacity, int64_t *n_intervals, int64_t *read_interval_off,
                                    uint32_t *chains, int64_t chains_capacity, int64_t *chains_used, int64_t *chain_off);

/* The same with the anchors in DEVICE memory (what blasr_hip_anchor_reads_device wrote, sorted = 1; list_off, reads and
 * everything else host): RemoveOverlappingAnchors, the sliding window of StoreLargestIntervals and GlobalChain run as
 * kernels over all lists of the block (blasr_amd/csrc/intervals_device.hip); one 32-byte record per cluster of anchors
 * comes back, and the host ends with the chains' p-values (its libm's logarithms) and the interval sets.  The element
 * the reference reads one past a list nothing was removed from (`stale` above) is taken as zeros. */
int blasr_hip_find_intervals_device(blasr_hip_ctx *ctx, const uint8_t *genome, int64_t genome_length,
                                    const uint32_t *seq_start_pos, int32_t n_seq_pos, const int32_t *count_table,
                                    int32_t count_tuple_size, int32_t n_reads, const blasr_hip_read *reads,
                                    const blasr_hip_anchor *anchors_device, const int64_t *list_off,
                                    const blasr_hip_interval_params *params, blasr_hip_interval *intervals,
                                    float *p_values, int64_t capacity, int64_t *n_intervals, int64_t *read_interval_off);

/* blasr_hip_find_intervals_device that also hands out the intervals' chained anchors, like blasr_hip_find_intervals_chains:
 * the chain kernel writes every cluster's chain on the device, the chains of the clusters that became intervals are gathered
 * there and come back in one copy (chains / chain_off: host memory). */
int blasr_hip_find_intervals_device_chains(blasr_hip_ctx *ctx, const uint8_t *genome, int64_t genome_length,
                                           const uint32_t *seq_start_pos, int32_t n_seq_pos, const int32_t *count_table,
                                           int32_t count_tuple_size, int32_t n_reads, const blasr_hip_read *reads,
                                           const blasr_hip_anchor *anchors_device, const int64_t *list_off,
                                           const blasr_hip_interval_params *params, blasr_hip_interval *intervals,
                                           float *p_values, int64_t capacity, int64_t *n_intervals, int64_t *read_interval_off,
                                           uint32_t *chains, int64_t chains_capacity, int64_t *chains_used, int64_t *chain_off);

/* ------------------------------------------------------------------------------------------------
 * next-4: the suffix array and its lookup table built on the device, and sawriter's file written.  Replaces
 * LarssonBuildSuffixArray + BuildLookupTable (common/datastructures/suffixarray/SuffixArray.h:256-269, :193-249) as
 * Blasr.cpp runs them when no .sa is given (alignment/Blasr.cpp:4407-4421) and as sawriter does
 * (alignment/SAWriter.cpp:187-232), and SuffixArray::Write (:505-532).  The array of the uploaded genome stays
 * resident for blasr_hip_anchor_reads_*; the host copies are optional (index_out: genome length entries,
 * start_pos_out / end_pos_out: 4^lookup_prefix_length entries; lookup_prefix_length 0..14, 0 = no table).
 * Working memory on the device: 32 bytes per genome base.
 * ------------------------------------------------------------------------------------------------ */
int blasr_hip_build_suffix_array(blasr_hip_ctx *ctx, int32_t lookup_prefix_length, uint32_t *index_out,
                                 uint32_t *start_pos_out, uint32_t *end_pos_out);
int blasr_hip_sa_write(const char *path, const uint32_t *index, int64_t length, const uint32_t *start_pos_table,
                       const uint32_t *end_pos_table, int64_t lookup_table_length, int32_t lookup_prefix_length);
/* The same file from the index resident on the context's device (after blasr_hip_build_suffix_array or
 * blasr_hip_upload_suffix_array), streamed to the file in pieces: sawriter's output for the resident genome without a
 * host copy of the 4 bytes per base. */
int blasr_hip_sa_write_resident(blasr_hip_ctx *ctx, const char *path);

/* The .ctab file of a tuple count table (TupleCountTable::Write / Read, datastructures/tuplelists/TupleCountTable.h:75-90;
 * Blasr.cpp:4463-4477 picks up <genome>.ctab): blasr_hip_ctab_read returns the number of counts in the file (< 0 on
 * error) and copies them when count_table has room. */
int blasr_hip_ctab_write(const char *path, const int32_t *count_table, int32_t tuple_size);
int64_t blasr_hip_ctab_read(const char *path, int32_t *count_table, int64_t capacity, int32_t *tuple_size);

/* ------------------------------------------------------------------------------------------------
 * MapRead for a block of reads from the reads alone (alignment/Blasr.cpp:2247-2644): anchoring on the device
 * (blasr_hip_anchor_reads_device, in sub-blocks of reads_per_block reads), the anchors of a sub-block to pinned host
 * memory and into blasr_hip_find_intervals on host threads while the device searches the next sub-block, then
 * blasr_hip_map_reads_device on all the intervals found.  Needs the genome and its suffix array on the device
 * (blasr_hip_upload_genome / _adopt_, blasr_hip_upload_suffix_array / blasr_hip_build_suffix_array) and, for the
 * p-values of the interval finder, the genome, its contig table and its tuple count table on the host.
 * reads_host / reads_device: the same read table in both places; read_pool_device: device.
 * intervals_out (may be NULL): the candidate intervals found, read by read.
 * ------------------------------------------------------------------------------------------------ */
typedef struct blasr_hip_pipeline_params {
  blasr_hip_anchor_params anchor;      /* sorted is forced to 1 */
  blasr_hip_interval_params interval;
  blasr_hip_map_params map;
  int32_t reads_per_block;             /* 100000 */
  /* 0: every sub-block is anchored, then the path runs once on all the intervals found.  1: the path runs sub-block by
   * sub-block while a host thread anchors the next one on a helper context of the same device (the search is bound by
   * the latency of random sectors, the path by its instructions: side by side they take less than one after the other).
   * Same hits in the same order; without fetch arenas blasr_hip_map_device_arenas then holds the LAST sub-block only. */
  int32_t overlap;
} blasr_hip_pipeline_params;
int blasr_hip_pipeline_default_params(blasr_hip_pipeline_params *params);
int blasr_hip_map_from_reads_device(blasr_hip_ctx *ctx, const uint8_t *genome_host, const uint32_t *seq_start_pos,
                                    int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size,
                                    int32_t n_reads, const blasr_hip_read *reads_host, const blasr_hip_read *reads_device,
                                    const uint8_t *read_pool_device, const blasr_hip_pipeline_params *params,
                                    blasr_hip_hit *hits, int64_t hits_capacity, int64_t *n_hits, int64_t *read_hit_off,
                                    blasr_hip_map_fetch *fetch, blasr_hip_interval *intervals_out,
                                    int64_t intervals_capacity, int64_t *n_intervals);
/* Same with a host read table and pool (uploaded for the call). */
int blasr_hip_map_from_reads_host(blasr_hip_ctx *ctx, const uint8_t *genome_host, const uint32_t *seq_start_pos,
                                  int32_t n_seq_pos, const int32_t *count_table, int32_t count_tuple_size, int32_t n_reads,
                                  const blasr_hip_read *reads, const uint8_t *read_pool, int64_t read_pool_bytes,
                                  const blasr_hip_pipeline_params *params, blasr_hip_hit *hits, int64_t hits_capacity,
                                  int64_t *n_hits, int64_t *read_hit_off, blasr_hip_map_fetch *fetch,
                                  blasr_hip_interval *intervals_out, int64_t intervals_capacity, int64_t *n_intervals);
/* Wall time (ms) of the last call: [0] anchoring of all sub-blocks (host stage underneath), [1] waiting for the host
 * stage's last sub-block, [2] the path, [3] host time spent finding intervals; counters: anchors, intervals. */
int blasr_hip_pipeline_last_timing(const blasr_hip_ctx *ctx, float ms[4], int64_t counters[2]);

#ifdef __cplusplus
}
#endif
#endif
