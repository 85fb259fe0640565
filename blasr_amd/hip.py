"""ctypes binding of include/blasr_hip.h.

Host-side mirror of the reference's refine call surface:

* ``RefineBatch``  collects what RefineAlignment hands to AffineGuidedAlign per
  candidate (alignment/Blasr.cpp:850-867): the trimmed query, the trimmed
  target window and the candidate's SDP blocks.
* ``Context.refine``  runs the batch on the GPU and returns, per candidate,
  what RefineAlignment copies back (Blasr.cpp:888-914): qPos/tPos, blocks,
  gaps, score, nCells and the statistics.

No algorithm lives here -- only packing/unpacking.
"""
import ctypes as C
import os

import numpy as np

# BLASR_HIP_LIBRARY: another build of the same library (tools/sanitized_cpu_tests.sh loads libblasr_hip_san.so, the host side
# under ASan / UBSan); never anything but a build of blasr_amd/csrc
LIB_PATH = os.environ.get("BLASR_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libblasr_hip.so")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_UNSUPPORTED, ERR_CAPACITY = 0, -1, -2, -3, -4, -5, -6
TASK_EMPTY_GUIDE, TASK_BAD_GUIDE, TASK_BAD_PATH, TASK_OPS_OVERFLOW = 1, 2, 4, 8
OP_M, OP_I, OP_D = 0, 1, 2
GAP_QUERY, GAP_TARGET = 0, 1  # Gap::GapSeq, AlignmentGapList.h:11

EXPORTS = [
    "blasr_hip_abi_version", "blasr_hip_create", "blasr_hip_destroy", "blasr_hip_last_error",
    "blasr_hip_set_workspace_limit", "blasr_hip_stream", "blasr_hip_refine_device", "blasr_hip_refine_host",
    "blasr_hip_refine_last_timing", "blasr_hip_refine_last_fill4", "blasr_hip_sdp_device", "blasr_hip_sdp_host", "blasr_hip_sdp_last_timing",
    "blasr_hip_upload_genome", "blasr_hip_windows_device", "blasr_hip_windows_host",
    "blasr_hip_sort_candidates", "blasr_hip_remove_low_quality", "blasr_hip_remove_overlapping",
    "blasr_hip_kband_device", "blasr_hip_kband_host", "blasr_hip_kband_linear_device", "blasr_hip_kband_linear_host",
    "blasr_hip_extend_default_params", "blasr_hip_extend_device", "blasr_hip_extend_host",
    "blasr_hip_alignment_is_printed", "blasr_hip_sam_unaligned_line",
    "blasr_hip_ref_contig_location", "blasr_hip_cigar", "blasr_hip_m4_line", "blasr_hip_map_qvs",
    "blasr_hip_sam_line", "blasr_hip_order_gaps_by_type",
    "blasr_hip_refine_from_sdp_device", "blasr_hip_map_reads_device", "blasr_hip_map_reads_host",
    "blasr_hip_map_default_params", "blasr_hip_map_last_timing", "blasr_hip_map_device_arenas",
    "blasr_hip_adopt_genome_device", "blasr_hip_held_bytes", "blasr_hip_sdp_last_bytes",
    "blasr_hip_fasta_read", "blasr_hip_fasta_parse", "blasr_hip_fasta_free", "blasr_hip_fasta_count",
    "blasr_hip_fasta_record", "blasr_hip_fasta_name", "blasr_hip_fasta_concatenate",
    "blasr_hip_fastq_read", "blasr_hip_fastq_parse", "blasr_hip_fastq_quality", "blasr_hip_fasta_error",
    "blasr_hip_sa_read", "blasr_hip_sa_free", "blasr_hip_sa_arrays",
    "blasr_hip_anchor_default_params", "blasr_hip_upload_suffix_array", "blasr_hip_borrow_index", "blasr_hip_anchor_reads_device",
    "blasr_hip_anchor_reads_host", "blasr_hip_anchor_last_timing",
    "blasr_hip_interval_default_params", "blasr_hip_count_tuples", "blasr_hip_find_intervals",
    "blasr_hip_build_suffix_array", "blasr_hip_sa_write", "blasr_hip_ctab_write", "blasr_hip_ctab_read",
    "blasr_hip_count_tuples_device", "blasr_hip_pipeline_default_params", "blasr_hip_map_from_reads_device", "blasr_hip_map_from_reads_host",
    "blasr_hip_pipeline_last_timing", "blasr_hip_find_intervals_device", "blasr_hip_find_intervals_chains", "blasr_hip_find_intervals_device_chains",
    "blasr_hip_align_between_anchors_device", "blasr_hip_align_between_anchors_host",
    "blasr_hip_sam_reads_read", "blasr_hip_sam_reads_parse", "blasr_hip_refine_last_kernels", "blasr_hip_kernel_name", "blasr_hip_build_id",
    "blasr_hip_kband_stats_device", "blasr_hip_kband_from_sdp_device", "blasr_hip_sa_write_resident",
    "blasr_hip_sdp_last_kernels", "blasr_hip_sdp_kernel_name", "blasr_hip_map_set_no_guided_align",
]

# include/blasr_hip.h: blasr_hip_sdp_kernel_record
SDP_KERNEL_RECORD_DTYPE = np.dtype([("kind", "<i4"), ("launches", "<i4"), ("ms", "<f4"), ("concurrent", "<i4"),
                                    ("algorithmic_bytes", "<i8"), ("fragments", "<i8"), ("invocations", "<i8")])
# include/blasr_hip.h: blasr_hip_kernel_record
KERNEL_RECORD_DTYPE = np.dtype([("kind", "<i4"), ("launches", "<i4"), ("ms", "<f4"), ("ext0", "<i4"),
                                ("algorithmic_bytes", "<i8"), ("cells", "<i8"), ("candidates", "<i8")])


def build_id():
    """The loaded library's build id (a hash over its sources, blasr_hip_build_id)."""
    lib = load_library()
    lib.blasr_hip_build_id.restype = C.c_char_p
    return lib.blasr_hip_build_id().decode()


# include/blasr_hip.h: blasr_hip_interval, blasr_hip_read, blasr_hip_window, blasr_hip_candidate
INTERVAL_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("q_start", "<u4"), ("q_end", "<u4"), ("read", "<i4"),
                           ("strand", "<i4")])
READ_DTYPE = np.dtype([("off", "<i8"), ("length", "<u4"), ("subread_start", "<u4"), ("subread_end", "<u4"),
                       ("reserved", "<u4")])
WINDOW_DTYPE = np.dtype([("t_aligned_seq_pos", "<u4"), ("t_aligned_seq_length", "<u4"), ("t_strand", "<i4"),
                         ("t_index", "<i4"), ("t_length", "<u4"), ("q_aligned_seq_pos", "<u4"),
                         ("q_aligned_seq_length", "<u4"), ("q_length", "<u4")])
CANDIDATE_DTYPE = np.dtype([("score", "<i4"), ("t_pos", "<u4"), ("t_aligned_seq_pos", "<u4"), ("t_end", "<u4"),
                            ("t_index", "<i4"), ("t_strand", "<i4"), ("pct_similarity", "<f4"), ("n_blocks", "<i4"),
                            ("blk_off", "<i8")])
# blasr_hip_hit: one alignment MapRead keeps (blasr_hip_map_reads_*)
HIT_DTYPE = np.dtype([("read", "<i4"), ("interval", "<i4"), ("t_strand", "<i4"), ("t_index", "<i4"),
                      ("t_aligned_seq_pos", "<u4"), ("t_aligned_seq_length", "<u4"), ("t_length", "<u4"),
                      ("q_aligned_seq_pos", "<u4"), ("q_aligned_seq_length", "<u4"), ("q_length", "<u4"),
                      ("q_pos", "<i4"), ("t_pos", "<i4"), ("score", "<i4"), ("n_match", "<i4"), ("n_mismatch", "<i4"),
                      ("n_ins", "<i4"), ("n_del", "<i4"), ("n_cells", "<i4"), ("pct_similarity", "<f4"),
                      ("n_ops", "<i4"), ("ops_off", "<i8"), ("q_off", "<i8"), ("t_off", "<i8")])
assert INTERVAL_DTYPE.itemsize == 24 and READ_DTYPE.itemsize == 24 and WINDOW_DTYPE.itemsize == 32
assert CANDIDATE_DTYPE.itemsize == 40 and HIT_DTYPE.itemsize == 104


class _KbandParams(C.Structure):
    _fields_ = [("matrix", C.c_int32 * 25), ("hp_ins_open", C.c_int32), ("hp_ins_extend", C.c_int32),
                ("ins_open", C.c_int32), ("ins_extend", C.c_int32), ("del_", C.c_int32)]


class _KbandLinearParams(C.Structure):
    _fields_ = [("matrix", C.c_int32 * 25), ("ins", C.c_int32), ("del_", C.c_int32), ("fn_ins", C.c_int32),
                ("fn_del", C.c_int32), ("align_type", C.c_int32)]


class _KbandBatch(C.Structure):
    _fields_ = [("n", C.c_int32), ("seq", C.c_void_p), ("q_off", C.c_void_p), ("q_len", C.c_void_p),
                ("t_off", C.c_void_p), ("t_len", C.c_void_p), ("k", C.c_void_p)]


class _PrintParams(C.Structure):
    _fields_ = [("max_score", C.c_int32), ("min_pct_identity", C.c_float), ("min_align_length", C.c_int32),
                ("min_map_qv", C.c_int32)]


class _ExtendParams(C.Structure):
    _fields_ = [("matrix", C.c_int32 * 25), ("ins", C.c_int32), ("del_", C.c_int32), ("band", C.c_int32),
                ("min_extend_n_bases", C.c_int32), ("max_n_drops", C.c_int32)]


class _ExtendBatch(C.Structure):
    _fields_ = [("n", C.c_int32), ("seq", C.c_void_p), ("q_off", C.c_void_p), ("q_len", C.c_void_p),
                ("t_off", C.c_void_p), ("t_len", C.c_void_p), ("q_pos", C.c_void_p), ("t_pos", C.c_void_p),
                ("reverse", C.c_void_p)]


def ops_to_alignment(ops):
    """Alignment::ArrowPathToAlignment (Alignment.h:190-254) on run-length ops: blocks, and one gap list per
    slot (list 0 leads); a gap list that runs to the end of the path is emptied.  Returns
    (blocks (n,3), gapCounts, gapItems (m,2))."""
    blocks, gap_counts, gap_items = [], [], []
    q = t = 0
    ops = np.asarray(ops, dtype=np.uint32)
    types, lens = (ops & 3).tolist(), (ops >> 2).tolist()
    i, n = 0, len(types)
    before_first = True
    while i < n:
        if not before_first and types[i] == OP_M:
            blocks.append((q, t, lens[i]))
            q += lens[i]
            t += lens[i]
            i += 1
        gap_counts.append(0)
        first_item = len(gap_items)
        while i < n and types[i] != OP_M:
            if types[i] == OP_I:
                gap_items.append((GAP_TARGET, lens[i]))
                q += lens[i]
            else:
                gap_items.append((GAP_QUERY, lens[i]))
                t += lens[i]
            gap_counts[-1] += 1
            i += 1
        if i == n:
            del gap_items[first_item:]
            gap_counts[-1] = 0
        before_first = False
    return (np.array(blocks, np.int32).reshape(-1, 3), np.array(gap_counts, np.int32),
            np.array(gap_items, np.int32).reshape(-1, 2))


class _WindowParams(C.Structure):
    _fields_ = [("approximate_max_insertion_rate", C.c_float), ("extend_ends", C.c_int32),
                ("do_global_alignment", C.c_int32), ("reserved", C.c_int32)]

# SMRTDistanceMatrix (ScoreMatrices.h:22-28): the matrix Blasr.cpp installs at :807-809
SMRT_MATRIX = np.array(
    [[-5, 6, 6, 6, 0], [6, -5, 6, 6, 0], [6, 6, -5, 6, 0], [6, 6, 6, -5, 0], [0, 0, 0, 0, 0]], dtype=np.int32)


class BlasrHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("blasr_hip error %d: %s" % (code, message))
        self.code = code


class _ScoreParams(C.Structure):
    _fields_ = [("matrix", C.c_int32 * 25), ("ins", C.c_int32), ("del_", C.c_int32),
                ("affine_open", C.c_int32), ("affine_extend", C.c_int32)]


class _RefineBatch(C.Structure):
    _fields_ = [("n", C.c_int32), ("seq", C.c_void_p), ("q_off", C.c_void_p), ("q_len", C.c_void_p),
                ("t_off", C.c_void_p), ("t_len", C.c_void_p), ("blocks", C.c_void_p), ("blk_off", C.c_void_p),
                ("n_blocks", C.c_void_p)]


class _SdpParams(C.Structure):
    _fields_ = [("word_size", C.c_int32), ("sdp_ins", C.c_int32), ("sdp_del", C.c_int32), ("indel_rate", C.c_float),
                ("align_type", C.c_int32), ("detailed", C.c_int32), ("extend_front", C.c_int32),
                ("prefix_length", C.c_int32), ("recurse", C.c_int32), ("recurse_over", C.c_int32),
                ("max_per_position", C.c_int32), ("sw_align_type", C.c_int32)]


class _SdpBatch(C.Structure):
    _fields_ = [("n", C.c_int32), ("seq", C.c_void_p), ("q_off", C.c_void_p), ("q_len", C.c_void_p),
                ("t_off", C.c_void_p), ("t_len", C.c_void_p)]


SDP_RESULT_DTYPE = np.dtype([
    ("status", "<i4"), ("score", "<i4"), ("q_pos", "<i4"), ("t_pos", "<i4"), ("n_blocks", "<i4"),
    ("stats_score", "<i4"), ("n_match", "<i4"), ("n_mismatch", "<i4"), ("n_ins", "<i4"), ("n_del", "<i4"),
    ("blk_off", "<i8")])
assert SDP_RESULT_DTYPE.itemsize == 48


class _MapParams(C.Structure):
    _fields_ = [("window", _WindowParams), ("sdp", _SdpParams), ("score", _ScoreParams), ("band", C.c_int32),
                ("n_candidates", C.c_int32), ("max_gap", C.c_int32), ("max_score", C.c_int32),
                ("min_pct_identity", C.c_float), ("sort_refined", C.c_int32), ("extend_alignments", C.c_int32),
                ("extend_band_size", C.c_int32), ("max_extend_dropoff", C.c_int32),
                ("refine_between_anchors_only", C.c_int32), ("indel", C.c_int32), ("sdp_bypass_threshold", C.c_float),
                ("use_guided_align", C.c_int32), ("iv_chains", C.c_void_p), ("iv_chain_off", C.c_void_p)]


class _RbaoParams(C.Structure):
    _fields_ = [("sdp", _SdpParams), ("score", _ScoreParams), ("band", C.c_int32), ("indel", C.c_int32),
                ("sdp_bypass_threshold", C.c_float), ("reserved", C.c_int32)]


class _AnchorParams(C.Structure):
    _fields_ = [("min_match_length", C.c_uint32), ("min_prefix_match_length", C.c_uint32), ("expand", C.c_int32),
                ("max_anchors_per_position", C.c_int32), ("advance_exact_matches", C.c_int32),
                ("max_lcp_length", C.c_int32), ("use_lookup_table", C.c_int32), ("stop_mapping_once_unique", C.c_int32),
                ("forward_only", C.c_int32), ("sorted", C.c_int32)]


class _IntervalParams(C.Structure):
    _fields_ = [("n_candidates", C.c_int32), ("min_interval", C.c_int32), ("indel_rate", C.c_float),
                ("max_p_value", C.c_float), ("remove_contained_intervals", C.c_int32), ("threads", C.c_int32)]


class _PipelineParams(C.Structure):
    _fields_ = [("anchor", _AnchorParams), ("interval", _IntervalParams), ("map", _MapParams),
                ("reads_per_block", C.c_int32), ("overlap", C.c_int32)]


# blasr_hip_anchor: MatchPos t, q, l
ANCHOR_DTYPE = np.dtype([("t", "<u4"), ("q", "<u4"), ("l", "<u4")])


class _MapFetch(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("seq_capacity", C.c_int64), ("seq_used", C.c_int64), ("ops", C.c_void_p),
                ("ops_capacity", C.c_int64), ("ops_used", C.c_int64)]


# params.indelRate * 3 as the reference computes it (float x int in single precision, Blasr.cpp:1717)
DEFAULT_INDEL_RATE = float(np.float32(0.3) * np.float32(3))


class SdpParams:
    """What AlignIntervals passes to SDPAlign (alignment/Blasr.cpp:1716-1722), defaults of
    MappingParameters.h:192-361."""

    def __init__(self, k=11, sdp_ins=5, sdp_del=10, indel_rate=DEFAULT_INDEL_RATE, align_type=0, detailed=1,
                 extend_front=0, prefix=50, recurse=2, recurse_over=1000, max_per_pos=0, sw_align_type=0):
        self.sw_align_type = sw_align_type
        self.k, self.sdp_ins, self.sdp_del, self.indel_rate = k, sdp_ins, sdp_del, indel_rate
        self.align_type, self.detailed, self.extend_front = align_type, detailed, extend_front
        self.prefix, self.recurse, self.recurse_over, self.max_per_pos = prefix, recurse, recurse_over, max_per_pos

    def as_struct(self):
        s = _SdpParams()
        s.word_size, s.sdp_ins, s.sdp_del, s.indel_rate = self.k, self.sdp_ins, self.sdp_del, self.indel_rate
        s.align_type, s.detailed, s.extend_front = self.align_type, self.detailed, self.extend_front
        s.prefix_length, s.recurse, s.recurse_over = self.prefix, self.recurse, self.recurse_over
        s.max_per_position = self.max_per_pos
        s.sw_align_type = self.sw_align_type
        return s


RESULT_DTYPE = np.dtype([
    ("status", "<i4"), ("q_pos", "<i4"), ("t_pos", "<i4"), ("dp_score", "<i4"), ("n_cells", "<i4"),
    ("score", "<i4"), ("n_match", "<i4"), ("n_mismatch", "<i4"), ("n_ins", "<i4"), ("n_del", "<i4"),
    ("n_ops", "<i4"), ("reserved", "<i4"), ("ops_off", "<i8")])
assert RESULT_DTYPE.itemsize == 56


class ScoreParams:
    """InitializeScoreFunction + SMRTDistanceMatrix defaults (MappingParameters.h:587-592, :338-342)."""

    def __init__(self, matrix=None, ins=5, dele=5, affine_open=50, affine_extend=0):
        self.matrix = np.ascontiguousarray(SMRT_MATRIX if matrix is None else matrix, dtype=np.int32).reshape(5, 5)
        self.ins, self.dele, self.affine_open, self.affine_extend = ins, dele, affine_open, affine_extend

    def as_struct(self):
        s = _ScoreParams()
        for i, v in enumerate(self.matrix.reshape(-1)):
            s.matrix[i] = int(v)
        s.ins, s.del_, s.affine_open, s.affine_extend = self.ins, self.dele, self.affine_open, self.affine_extend
        return s


_lib = None


def load_library():
    """Load libblasr_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BlasrHipError(ERR_NO_DEVICE, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
                                % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.blasr_hip_last_error.restype = C.c_char_p
        lib.blasr_hip_stream.restype = C.c_void_p
        lib.blasr_hip_held_bytes.restype = C.c_int64
        lib.blasr_hip_held_bytes.argtypes = [C.c_void_p]
        for name in ("blasr_hip_refine_device", "blasr_hip_refine_host"):
            getattr(lib, name).argtypes = [C.c_void_p, C.POINTER(_RefineBatch), C.c_int32, C.POINTER(_ScoreParams),
                                           C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        lib.blasr_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.blasr_hip_destroy.argtypes = [C.c_void_p]
        lib.blasr_hip_last_error.argtypes = [C.c_void_p]
        lib.blasr_hip_stream.argtypes = [C.c_void_p]
        lib.blasr_hip_set_workspace_limit.argtypes = [C.c_void_p, C.c_int64]
        lib.blasr_hip_refine_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        lib.blasr_hip_sdp_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        lib.blasr_hip_sdp_last_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        lib.blasr_hip_refine_last_fill4.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64),
                                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        for name in ("blasr_hip_sdp_device", "blasr_hip_sdp_host"):
            getattr(lib, name).argtypes = [C.c_void_p, C.POINTER(_SdpBatch), C.POINTER(_SdpParams),
                                           C.POINTER(_ScoreParams), C.c_void_p, C.c_void_p, C.c_int64,
                                           C.POINTER(C.c_int64)]
        lib.blasr_hip_upload_genome.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        lib.blasr_hip_windows_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.POINTER(_WindowParams), C.c_void_p, C.c_int64, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        lib.blasr_hip_windows_host.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                               C.c_int64, C.POINTER(_WindowParams), C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        for name in ("blasr_hip_kband_device", "blasr_hip_kband_host"):
            getattr(lib, name).argtypes = [C.c_void_p, C.POINTER(_KbandBatch), C.POINTER(_KbandParams), C.c_void_p,
                                           C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        for name in ("blasr_hip_kband_linear_device", "blasr_hip_kband_linear_host"):
            getattr(lib, name).argtypes = [C.c_void_p, C.POINTER(_KbandBatch), C.POINTER(_KbandLinearParams), C.c_void_p,
                                           C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        lib.blasr_hip_alignment_is_printed.argtypes = [C.c_int32, C.c_float, C.c_uint32, C.c_int32, C.POINTER(_PrintParams)]
        lib.blasr_hip_extend_default_params.argtypes = [C.POINTER(_ExtendParams)]
        for name in ("blasr_hip_extend_device", "blasr_hip_extend_host"):
            getattr(lib, name).argtypes = [C.c_void_p, C.POINTER(_ExtendBatch), C.POINTER(_ExtendParams), C.c_void_p,
                                           C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        lib.blasr_hip_sort_candidates.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.blasr_hip_remove_low_quality.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                                     C.c_int32, C.c_int32]
        lib.blasr_hip_remove_overlapping.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_float]
        lib.blasr_hip_refine_from_sdp_device.argtypes = [C.c_void_p, C.POINTER(_SdpBatch), C.c_void_p, C.c_void_p,
                                                         C.c_int32, C.c_void_p, C.c_int32, C.POINTER(_ScoreParams),
                                                         C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        lib.blasr_hip_map_reads_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                   C.POINTER(_MapParams), C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                                   C.c_void_p, C.POINTER(_MapFetch)]
        lib.blasr_hip_map_reads_host.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                                 C.c_void_p, C.POINTER(_MapParams), C.c_void_p, C.c_int64,
                                                 C.POINTER(C.c_int64), C.c_void_p, C.POINTER(_MapFetch)]
        lib.blasr_hip_map_default_params.argtypes = [C.POINTER(_MapParams)]
        lib.blasr_hip_map_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        lib.blasr_hip_map_device_arenas.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        lib.blasr_hip_adopt_genome_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        lib.blasr_hip_anchor_default_params.argtypes = [C.POINTER(_AnchorParams)]
        lib.blasr_hip_upload_suffix_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                      C.c_int64, C.c_int32]
        lib.blasr_hip_anchor_reads_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                                      C.POINTER(_AnchorParams), C.c_void_p, C.c_int64,
                                                      C.POINTER(C.c_int64), C.c_void_p]
        lib.blasr_hip_anchor_reads_host.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                                    C.POINTER(_AnchorParams), C.c_void_p, C.c_int64,
                                                    C.POINTER(C.c_int64), C.c_void_p]
        lib.blasr_hip_anchor_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        lib.blasr_hip_build_suffix_array.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.blasr_hip_sa_write.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
        lib.blasr_hip_interval_default_params.argtypes = [C.POINTER(_IntervalParams)]
        lib.blasr_hip_count_tuples.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        lib.blasr_hip_find_intervals.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                                 C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.POINTER(_IntervalParams), C.c_void_p, C.c_void_p, C.c_int64,
                                                 C.POINTER(C.c_int64), C.c_void_p]
        lib.blasr_hip_find_intervals_chains.argtypes = lib.blasr_hip_find_intervals.argtypes + [
            C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
        lib.blasr_hip_align_between_anchors_host.argtypes = [
            C.c_void_p, C.POINTER(_SdpBatch), C.c_int64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
            C.POINTER(_RbaoParams), C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        _lib = lib
    return _lib


# --- next-2: first pieces of the output epilogue (host code, no device involved) --------------------
CLIP_HARD, CLIP_SOFT, CLIP_SUBREAD, CLIP_NONE = 0, 1, 2, 3  # SAMOutput::Clipping (printers/SAMPrinter.h:26)


class _CigarParams(C.Structure):
    _fields_ = [("clipping", C.c_int32), ("t_strand", C.c_int32), ("read_length", C.c_uint32),
                ("low_quality_prefix", C.c_uint32), ("low_quality_suffix", C.c_uint32), ("subread_start", C.c_uint32),
                ("subread_end", C.c_uint32), ("q_align_start", C.c_uint32), ("q_align_end", C.c_uint32)]


class _M4Fields(C.Structure):
    _fields_ = [("q_name", C.c_char_p), ("t_name", C.c_char_p), ("score", C.c_int32), ("pct_similarity", C.c_float),
                ("q_strand", C.c_int32), ("q_align_start", C.c_uint32), ("q_align_end", C.c_uint32),
                ("q_length", C.c_uint32), ("t_strand", C.c_int32), ("t_align_start", C.c_uint32),
                ("t_align_end", C.c_uint32), ("t_length", C.c_uint32), ("map_qv", C.c_int32)]


def ref_contig_location(t_aligned_seq_pos, t_aligned_seq_length, t_strand, seq_start_pos, genome_length):
    """AssignRefContigLocation (Blasr.cpp:2183-2220): (contig index, position of the window inside the contig)."""
    lib = load_library()
    sp = np.ascontiguousarray(seq_start_pos, dtype=np.uint32)
    contig, pos = C.c_int32(0), C.c_uint32(0)
    rc = lib.blasr_hip_ref_contig_location(C.c_uint32(t_aligned_seq_pos), C.c_uint32(t_aligned_seq_length), int(t_strand),
                                           sp.ctypes.data_as(C.c_void_p), len(sp), C.c_uint32(genome_length),
                                           C.byref(contig), C.byref(pos))
    if rc < 0:
        raise BlasrHipError(rc, "ref_contig_location")
    return contig.value, pos.value


def cigar(q_aligned_seq, t_aligned_seq, q_pos, t_pos, ops=None, blocks=None, clipping=CLIP_NONE, t_strand=0,
          read_length=0, low_quality_prefix=0, low_quality_suffix=0, subread_start=0, subread_end=0, q_align_start=0,
          q_align_end=0):
    """CreateCIGARString (printers/SAMPrinter.h:346-416) from the refine stage's op list (packed words or
    ops_to_alignment's input) or from blocks.  Returns (cigar string, (prefixSoft, suffixSoft, prefixHard, suffixHard))."""
    lib = load_library()
    q = np.ascontiguousarray(np.frombuffer(q_aligned_seq, dtype=np.uint8) if isinstance(q_aligned_seq, bytes) else q_aligned_seq, dtype=np.uint8)
    t = np.ascontiguousarray(np.frombuffer(t_aligned_seq, dtype=np.uint8) if isinstance(t_aligned_seq, bytes) else t_aligned_seq, dtype=np.uint8)
    o = None if ops is None else np.ascontiguousarray(ops, dtype=np.uint32)
    b = None if blocks is None else np.ascontiguousarray(blocks, dtype=np.int32).reshape(-1)
    p = _CigarParams(clipping, t_strand, read_length, low_quality_prefix, low_quality_suffix, subread_start, subread_end,
                     q_align_start, q_align_end)
    cap = 16 * (len(q) + len(t) + 8)
    out = C.create_string_buffer(cap)
    n = C.c_int64(0)
    clips = (C.c_uint32 * 4)()
    rc = lib.blasr_hip_cigar(q.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), int(q_pos), int(t_pos),
                             None if o is None else o.ctypes.data_as(C.c_void_p), C.c_int64(0 if o is None else len(o)),
                             None if b is None else b.ctypes.data_as(C.c_void_p), C.c_int64(0 if b is None else len(b) // 3),
                             C.byref(p), out, C.c_int64(cap), C.byref(n), clips)
    if rc < 0:
        raise BlasrHipError(rc, "cigar")
    return out.value.decode(), tuple(int(v) for v in clips)


def m4_line(q_name, t_name, score, pct_similarity, q_strand, q_align_start, q_align_end, q_length, t_strand,
            t_align_start, t_align_end, t_length, map_qv):
    """IntervalAlignmentPrinter::Print (printers/IntervalAlignmentPrinter.h:9-38): one -m 4 line."""
    lib = load_library()
    f = _M4Fields(q_name.encode(), t_name.encode(), score, pct_similarity, q_strand, q_align_start, q_align_end, q_length,
                  t_strand, t_align_start, t_align_end, t_length, map_qv)
    out = C.create_string_buffer(len(q_name) + len(t_name) + 256)
    n = C.c_int64(0)
    rc = lib.blasr_hip_m4_line(C.byref(f), out, C.c_int64(len(out)), C.byref(n))
    if rc < 0:
        raise BlasrHipError(rc, "m4_line")
    return out.value.decode()


def order_gaps_by_type(ops):
    """Alignment::OrderGapsByType (Alignment.h:255-317, -forPicard) on an op list."""
    lib = load_library()
    o = np.ascontiguousarray(ops, dtype=np.uint32)
    out = np.zeros(max(len(o), 1), dtype=np.uint32)
    n = C.c_int64(0)
    rc = lib.blasr_hip_order_gaps_by_type(o.ctypes.data_as(C.c_void_p), C.c_int64(len(o)), out.ctypes.data_as(C.c_void_p),
                                          C.c_int64(len(out)), C.byref(n))
    if rc < 0:
        raise BlasrHipError(rc, "order_gaps_by_type")
    return out[: n.value].copy()


class _SamFields(C.Structure):
    _fields_ = [("q_name", C.c_char_p), ("t_name", C.c_char_p), ("read_group_id", C.c_char_p), ("cigar", C.c_char_p),
                ("read_seq", C.c_void_p), ("read_qual", C.c_void_p), ("clips", C.c_uint32 * 4), ("clipping", C.c_int32),
                ("q_strand", C.c_int32), ("t_strand", C.c_int32), ("is_primary", C.c_int32), ("read_length", C.c_uint32),
                ("low_quality_prefix", C.c_uint32), ("low_quality_suffix", C.c_uint32), ("subread_start", C.c_uint32),
                ("subread_end", C.c_uint32), ("q_align_start", C.c_uint32), ("q_align_end", C.c_uint32),
                ("t_align_start", C.c_uint32), ("t_end", C.c_uint32), ("t_length", C.c_uint32), ("q_length", C.c_uint32),
                ("q_aligned_seq_pos", C.c_uint32), ("q_aligned_seq_length", C.c_uint32), ("map_qv", C.c_int32),
                ("score", C.c_int32), ("edit_dist", C.c_int32), ("hole_number", C.c_int32), ("qs", C.c_int32),
                ("qe", C.c_int32), ("num_passes", C.c_int32), ("subread_context", C.c_int32),
                ("read_quality", C.c_float), ("snr", C.c_float * 4)]


def sam_line(q_name, t_name, read_group_id, cigar_string, clips, read_seq, read_qual=None, **kw):
    """SAMOutput::PrintAlignment (printers/SAMPrinter.h:418-548): one SAM record.  kw: the remaining fields of
    blasr_hip_sam_fields (include/blasr_hip.h) by name; snr is a sequence of four."""
    lib = load_library()
    seq = np.ascontiguousarray(read_seq, dtype=np.uint8)
    qual = None if read_qual is None else np.ascontiguousarray(read_qual, dtype=np.uint8)
    f = _SamFields()
    f.q_name, f.t_name, f.read_group_id, f.cigar = q_name.encode(), t_name.encode(), read_group_id.encode(), cigar_string.encode()
    f.read_seq = seq.ctypes.data
    f.read_qual = None if qual is None else qual.ctypes.data
    f.clips = (C.c_uint32 * 4)(*[int(v) for v in clips])
    f.snr = (C.c_float * 4)(*[float(v) for v in kw.pop("snr", (0, 0, 0, 0))])
    for k, v in kw.items():
        setattr(f, k, v)
    cap = 2 * len(seq) + len(cigar_string) + len(q_name) + len(t_name) + len(read_group_id) + 512
    out = C.create_string_buffer(cap)
    n = C.c_int64(0)
    rc = lib.blasr_hip_sam_line(C.byref(f), out, C.c_int64(cap), C.byref(n))
    if rc < 0:
        raise BlasrHipError(rc, "sam_line")
    return out.value.decode()


def sam_unaligned_line(q_name, read_group_id, read_seq, subread_start=0, subread_end=None, clipping=CLIP_NONE, **kw):
    """SAMOutput::PrintUnalignedRead (printers/SAMPrinter.h:578-664): the record of a (sub)read without an alignment."""
    lib = load_library()
    seq = np.ascontiguousarray(read_seq, dtype=np.uint8)
    f = _SamFields()
    f.q_name, f.read_group_id = q_name.encode(), read_group_id.encode()
    f.read_seq = seq.ctypes.data
    f.read_length = len(seq)
    f.subread_start, f.subread_end = int(subread_start), len(seq) if subread_end is None else int(subread_end)
    f.clipping = int(clipping)
    f.hole_number = int(kw.pop("hole_number", -1))
    f.snr = (C.c_float * 4)(*[float(v) for v in kw.pop("snr", (0, 0, 0, 0))])
    f.qs, f.qe = f.subread_start + 1, f.subread_end + 1  # the subread's own (SMRTSequence.h:102-103) unless given
    for k, v in kw.items():
        setattr(f, k, v)
    cap = len(seq) + len(q_name) + len(read_group_id) + 512
    out = C.create_string_buffer(cap)
    n = C.c_int64(0)
    rc = lib.blasr_hip_sam_unaligned_line(C.byref(f), out, C.c_int64(cap), C.byref(n))
    if rc < 0:
        raise BlasrHipError(rc, "sam_unaligned_line")
    return out.value.decode()


class _MapQvAlignment(C.Structure):
    _fields_ = [("q_aligned_seq", C.c_void_p), ("t_aligned_seq", C.c_void_p), ("ops", C.c_void_p), ("n_ops", C.c_int64),
                ("q_pos", C.c_int32), ("t_pos", C.c_int32), ("q_align_start", C.c_uint32), ("q_align_end", C.c_uint32),
                ("q_strand", C.c_int32), ("q_length", C.c_uint32), ("n_match", C.c_int32), ("reserved", C.c_int32)]


class _MapQvParams(C.Structure):
    _fields_ = [("matrix", C.c_int32 * 25), ("ins", C.c_int32), ("del_", C.c_int32), ("affine_open", C.c_int32),
                ("affine_extend", C.c_int32), ("affine_align", C.c_int32), ("min_fraction_overlapping", C.c_float)]


SMRT_LOG_PROB_MATRIX = [0, 15, 15, 15, 15, 15, 0, 15, 15, 15, 15, 15, 0, 15, 15, 15, 15, 15, 0, 15, 15, 15, 15, 15, 0]


def map_qvs(alignments, substitution_qv=None, matrix=None, ins=5, dele=5, affine_open=50, affine_extend=0, affine_align=True,
            min_fraction_overlapping=0.75):
    """StoreMapQVs (Blasr.cpp:2761-2950) for the alignments of one read: a list of dicts with q, t (aligned
    sequences), ops, q_pos, t_pos, q_align_start, q_align_end, q_strand, q_length, n_match.
    Returns (mapQV list, probScore list)."""
    lib = load_library()
    n = len(alignments)
    arr = (_MapQvAlignment * max(n, 1))()
    keep = []
    for i, a in enumerate(alignments):
        q = np.ascontiguousarray(a["q"], dtype=np.uint8)
        t = np.ascontiguousarray(a["t"], dtype=np.uint8)
        o = np.ascontiguousarray(a["ops"], dtype=np.uint32)
        keep.append((q, t, o))
        arr[i] = _MapQvAlignment(q.ctypes.data, t.ctypes.data, o.ctypes.data, len(o), a["q_pos"], a["t_pos"],
                                 a["q_align_start"], a["q_align_end"], a["q_strand"], a["q_length"], a["n_match"], 0)
    p = _MapQvParams((C.c_int32 * 25)(*(SMRT_LOG_PROB_MATRIX if matrix is None else [int(v) for v in np.asarray(matrix).reshape(-1)])),
                     ins, dele, affine_open, affine_extend, 1 if affine_align else 0, min_fraction_overlapping)
    qv = None if substitution_qv is None else np.ascontiguousarray(substitution_qv, dtype=np.uint8)
    out = np.zeros(max(n, 1), dtype=np.int32)
    prob = np.zeros(max(n, 1), dtype=np.float32)
    rc = lib.blasr_hip_map_qvs(arr, n, C.byref(p), None if qv is None else qv.ctypes.data_as(C.c_void_p),
                               out.ctypes.data_as(C.c_void_p), prob.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise BlasrHipError(rc, "map_qvs")
    return [int(v) for v in out[:n]], [float(v) for v in prob[:n]]


# --- a-10: host-side ordering / pruning of a read's candidates (no device involved) ---------------
def sort_candidates(cands, order=None):
    """std::sort(alignmentPtrs, SortAlignmentPointersByScore()) (Blasr.cpp:226-236, :2547, :2178-2180).
    cands: CANDIDATE_DTYPE array; returns the new order (indices into cands)."""
    lib = load_library()
    cands = np.ascontiguousarray(cands, dtype=CANDIDATE_DTYPE)
    order = np.arange(len(cands), dtype=np.int32) if order is None else np.ascontiguousarray(order, dtype=np.int32).copy()
    n = lib.blasr_hip_sort_candidates(cands.ctypes.data, len(order), order.ctypes.data)
    if n < 0:
        raise BlasrHipError(n, "sort_candidates")
    return order


def remove_low_quality(cands, blocks, order, n_candidates=10, max_gap=0, max_score=-200):
    """RemoveLowQualityAlignments (Blasr.cpp:2022-2079).  Returns (order, cands): cands[i].score may have been
    set to max_score by the maxGap rule."""
    lib = load_library()
    cands = np.ascontiguousarray(cands, dtype=CANDIDATE_DTYPE).copy()
    order = np.ascontiguousarray(order, dtype=np.int32).copy()
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    n = lib.blasr_hip_remove_low_quality(cands.ctypes.data, blocks.ctypes.data if blocks.size else None, len(order),
                                         order.ctypes.data, n_candidates, max_gap, max_score)
    if n < 0:
        raise BlasrHipError(n, "remove_low_quality")
    return order[:n], cands


def remove_overlapping(cands, order, min_pct_identity=0.0):
    """RemoveOverlappingAlignments (Blasr.cpp:2081-2161)."""
    lib = load_library()
    cands = np.ascontiguousarray(cands, dtype=CANDIDATE_DTYPE)
    order = np.ascontiguousarray(order, dtype=np.int32).copy()
    n = lib.blasr_hip_remove_overlapping(cands.ctypes.data, len(order), order.ctypes.data, float(min_pct_identity))
    if n < 0:
        raise BlasrHipError(n, "remove_overlapping")
    return order[:n]


class RefineBatch:
    """Structure-of-arrays batch of refine candidates (host memory)."""

    def __init__(self):
        self._seq, self._blocks = [], []
        self.q_off, self.q_len, self.t_off, self.t_len, self.blk_off, self.n_blocks = [], [], [], [], [], []
        self._seq_bytes = 0
        self._n_blocks = 0

    def __len__(self):
        return len(self.q_off)

    def add(self, query, target, guide_blocks):
        """query/target: ASCII uint8 arrays (already trimmed as Blasr.cpp:850-859 does);
        guide_blocks: (n,3) int array of (qPos, tPos, length)."""
        q = np.ascontiguousarray(query, dtype=np.uint8)
        t = np.ascontiguousarray(target, dtype=np.uint8)
        b = np.ascontiguousarray(guide_blocks, dtype=np.int32).reshape(-1, 3)
        self.q_off.append(self._seq_bytes)
        self.q_len.append(len(q))
        self._seq_bytes += len(q)
        self.t_off.append(self._seq_bytes)
        self.t_len.append(len(t))
        self._seq_bytes += len(t)
        self._seq += [q, t]
        self.blk_off.append(self._n_blocks)
        self.n_blocks.append(len(b))
        self._n_blocks += len(b)
        self._blocks.append(b)

    def arrays(self):
        seq = np.concatenate(self._seq) if self._seq else np.zeros(0, np.uint8)
        blocks = np.concatenate(self._blocks).reshape(-1) if self._blocks else np.zeros(0, np.int32)
        return dict(seq=np.ascontiguousarray(seq), q_off=np.array(self.q_off, np.int64),
                    q_len=np.array(self.q_len, np.int32), t_off=np.array(self.t_off, np.int64),
                    t_len=np.array(self.t_len, np.int32), blocks=np.ascontiguousarray(blocks, dtype=np.int32),
                    blk_off=np.array(self.blk_off, np.int64), n_blocks=np.array(self.n_blocks, np.int32))

    def ops_bound(self):
        """Upper bound of ops: a path has at most qLen + tLen steps."""
        return int(sum(self.q_len) + sum(self.t_len) + 2 * len(self))


class RefineResult:
    """One refined candidate, in the reference's own terms."""

    def __init__(self, rec, ops):
        self.status = int(rec["status"])
        self.qPos, self.tPos = int(rec["q_pos"]), int(rec["t_pos"])
        self.dpScore, self.nCells, self.score = int(rec["dp_score"]), int(rec["n_cells"]), int(rec["score"])
        self.nMatch, self.nMismatch = int(rec["n_match"]), int(rec["n_mismatch"])
        self.nIns, self.nDel = int(rec["n_ins"]), int(rec["n_del"])
        self.ops = ops  # uint32 array, (len << 2) | type

    @property
    def pctSimilarity(self):
        # AlignmentUtils.h:566-577: nMatch*2.0 / (2 * alignment columns) * 100, stored as float
        cols = self.nMatch + self.nMismatch + self.nIns + self.nDel
        if cols == 0:
            return np.float32(0)
        return np.float32((self.nMatch * 2.0) / (cols + cols) * 100)

    def blocks_and_gaps(self):
        """Rebuild Alignment.blocks / Alignment.gaps (Alignment.h:190-254 conventions):
        blocks relative to (qPos,tPos); gaps[0] empty, gaps[i+1] between block i and i+1,
        last list empty."""
        blocks, gap_counts, gap_items = [], [0], []
        q = t = 0
        types, lens = self.ops & 3, self.ops >> 2
        for ty, ln in zip(types.tolist(), lens.tolist()):
            if ty == OP_M:
                blocks.append((q, t, ln))
                gap_counts.append(0)
                q += ln
                t += ln
            elif ty == OP_I:
                gap_items.append((GAP_TARGET, ln))
                gap_counts[-1] += 1
                q += ln
            else:
                gap_items.append((GAP_QUERY, ln))
                gap_counts[-1] += 1
                t += ln
        if not blocks:
            gap_counts = [0] if self.status == 0 else []
        return (np.array(blocks, np.int32).reshape(-1, 3), np.array(gap_counts, np.int32),
                np.array(gap_items, np.int32).reshape(-1, 2))

    def as_dict(self):
        blocks, gc, gi = self.blocks_and_gaps()
        return dict(qPos=self.qPos, tPos=self.tPos, dpScore=self.dpScore, nCells=self.nCells, score=self.score,
                    nMatch=self.nMatch, nMismatch=self.nMismatch, nIns=self.nIns, nDel=self.nDel,
                    pct=float(self.pctSimilarity), blocks=blocks, gapCounts=gc, gapItems=gi)


class Context:
    """One blasr_hip_ctx (one per host thread, like MappingBuffers)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.blasr_hip_create(device, C.byref(h))
        if rc != OK:
            raise BlasrHipError(rc, "blasr_hip_create(device=%d) failed: no usable gfx950 device" % device)
        self.handle = h

    def close(self):
        if self.handle:
            self.lib.blasr_hip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=()):
        if rc != OK and rc not in allow:
            raise BlasrHipError(rc, self.lib.blasr_hip_last_error(self.handle).decode())

    @property
    def stream(self):
        return self.lib.blasr_hip_stream(self.handle)

    def held_bytes(self):
        return int(self.lib.blasr_hip_held_bytes(self.handle))

    def set_workspace_limit(self, nbytes):
        self._check(self.lib.blasr_hip_set_workspace_limit(self.handle, int(nbytes)))

    def last_timing(self):
        ms = (C.c_float * 4)()
        cnt = (C.c_int64 * 3)()
        self._check(self.lib.blasr_hip_refine_last_timing(self.handle, ms, cnt))
        return dict(geometry_ms=ms[0], fill_ms=ms[1], traceback_ms=ms[2], compact_ms=ms[3], cells=cnt[0],
                    fill_bytes=cnt[1], fill_launches=cnt[2])

    def refine(self, batch, band=16, score=None, ops_capacity=None):
        """RefineAlignments for a host batch; returns a list of RefineResult."""
        score = score or ScoreParams()
        n = len(batch)
        if n == 0:
            return []
        a = batch.arrays()
        sb = _RefineBatch()
        sb.n = n
        for k in ("seq", "q_off", "q_len", "t_off", "t_len", "blocks", "blk_off", "n_blocks"):
            setattr(sb, k, a[k].ctypes.data)
        cap = batch.ops_bound() if ops_capacity is None else int(ops_capacity)
        results = np.zeros(n, dtype=RESULT_DTYPE)
        ops = np.zeros(max(cap, 1), dtype=np.uint32)
        used = C.c_int64(0)
        sc = score.as_struct()
        rc = self.lib.blasr_hip_refine_host(self.handle, C.byref(sb), band, C.byref(sc), results.ctypes.data,
                                            ops.ctypes.data, cap, C.byref(used))
        self._check(rc)
        out = []
        for i in range(n):
            r = results[i]
            o = ops[int(r["ops_off"]): int(r["ops_off"]) + int(r["n_ops"])].copy()
            out.append(RefineResult(r, o))
        return out

    def sdp(self, pairs, sdp=None, score=None, blocks_capacity=None):
        """SDPAlign + ComputeAlignmentStats for a list of (query, target) uint8 arrays (host memory).
        Returns a list of dicts: score, qPos, tPos, blocks (n,3), stats."""
        sdp, score = sdp or SdpParams(), score or ScoreParams()
        n = len(pairs)
        if n == 0:
            return []
        seqs, q_off, q_len, t_off, t_len, at = [], [], [], [], [], 0
        for q, t in pairs:
            q = np.ascontiguousarray(q, dtype=np.uint8)
            t = np.ascontiguousarray(t, dtype=np.uint8)
            q_off.append(at); q_len.append(len(q)); at += len(q)
            t_off.append(at); t_len.append(len(t)); at += len(t)
            seqs += [q, t]
        seq = np.ascontiguousarray(np.concatenate(seqs)) if at else np.zeros(1, np.uint8)
        q_off, t_off = np.array(q_off, np.int64), np.array(t_off, np.int64)
        q_len, t_len = np.array(q_len, np.int32), np.array(t_len, np.int32)
        sb = _SdpBatch()
        sb.n = n
        sb.seq, sb.q_off, sb.q_len = seq.ctypes.data, q_off.ctypes.data, q_len.ctypes.data
        sb.t_off, sb.t_len = t_off.ctypes.data, t_len.ctypes.data
        cap = int(np.minimum(q_len, t_len).sum() + n) if blocks_capacity is None else int(blocks_capacity)
        results = np.zeros(n, dtype=SDP_RESULT_DTYPE)
        blocks = np.zeros(3 * max(cap, 1), dtype=np.int32)
        used = C.c_int64(0)
        sp, sc = sdp.as_struct(), score.as_struct()
        rc = self.lib.blasr_hip_sdp_host(self.handle, C.byref(sb), C.byref(sp), C.byref(sc), results.ctypes.data,
                                         blocks.ctypes.data, cap, C.byref(used))
        self._check(rc)
        out = []
        for r in results:
            b0 = int(r["blk_off"])
            out.append(dict(status=int(r["status"]), score=int(r["score"]), qPos=int(r["q_pos"]),
                            tPos=int(r["t_pos"]), blocks=blocks[3 * b0: 3 * (b0 + int(r["n_blocks"]))].reshape(-1, 3).copy(),
                            stats=dict(score=int(r["stats_score"]), nMatch=int(r["n_match"]),
                                       nMismatch=int(r["n_mismatch"]), nIns=int(r["n_ins"]), nDel=int(r["n_del"]))))
        return out

    # --- a-1: candidate windows from the resident genome (AlignIntervals, Blasr.cpp:1208-1336) ---
    def upload_genome(self, genome, contig_starts=None):
        genome = np.ascontiguousarray(genome, dtype=np.uint8)
        self._genome_keepalive = genome
        self.genome_length = len(genome)
        cs, nc = None, 0
        if contig_starts is not None:
            cs = np.ascontiguousarray(contig_starts, dtype=np.uint32)
            nc = len(cs) - 1
        self._check(self.lib.blasr_hip_upload_genome(self.handle, genome.ctypes.data, len(genome),
                                                     cs.ctypes.data if cs is not None else None, nc))

    @staticmethod
    def _window_params(rate, extend_ends, do_global):
        wp = _WindowParams()
        wp.approximate_max_insertion_rate, wp.extend_ends, wp.do_global_alignment = rate, int(extend_ends), int(do_global)
        return wp

    def windows(self, intervals, reads, read_pool, rate=1.3, extend_ends=True, do_global=False, seq_capacity=None):
        """intervals: INTERVAL_DTYPE array, reads: READ_DTYPE array, read_pool: uint8 ASCII.  Returns
        (seq, q_off, q_len, t_off, t_len, windows) -- the first five are a blasr_hip_sdp_batch."""
        intervals = np.ascontiguousarray(intervals, dtype=INTERVAL_DTYPE)
        reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
        read_pool = np.ascontiguousarray(read_pool, dtype=np.uint8)
        n = len(intervals)
        if seq_capacity is None:
            seq_capacity = int(sum(int(reads[iv["read"]]["length"]) * 4 + int(iv["end"] - iv["start"]) for iv in intervals)) + 64
        seq = np.zeros(max(seq_capacity, 1), dtype=np.uint8)
        q_off, t_off = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64)
        q_len, t_len = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        win = np.zeros(max(n, 1), dtype=WINDOW_DTYPE)
        used = C.c_int64(0)
        wp = self._window_params(rate, extend_ends, do_global)
        rc = self.lib.blasr_hip_windows_host(self.handle, n, intervals.ctypes.data, len(reads), reads.ctypes.data,
                                             read_pool.ctypes.data, len(read_pool), C.byref(wp), seq.ctypes.data,
                                             seq_capacity, q_off.ctypes.data, q_len.ctypes.data, t_off.ctypes.data,
                                             t_len.ctypes.data, win.ctypes.data, C.byref(used))
        self._check(rc)
        return seq[: used.value], q_off[:n], q_len[:n], t_off[:n], t_len[:n], win[:n]

    def windows_device(self, n, intervals_ptr, reads_ptr, read_pool_ptr, seq_ptr, seq_capacity, q_off_ptr, q_len_ptr,
                       t_off_ptr, t_len_ptr, windows_ptr, rate=1.3, extend_ends=True, do_global=False):
        """Raw device pointers (e.g. torch.Tensor.data_ptr()); returns the bytes written to seq."""
        used = C.c_int64(0)
        wp = self._window_params(rate, extend_ends, do_global)
        self._check(self.lib.blasr_hip_windows_device(self.handle, n, intervals_ptr, reads_ptr, read_pool_ptr,
                                                      C.byref(wp), seq_ptr, seq_capacity, q_off_ptr, q_len_ptr,
                                                      t_off_ptr, t_len_ptr, windows_ptr, C.byref(used)))
        return used.value

    def adopt_genome_device(self, genome_ptr, length, contig_starts=None):
        """The genome already sits in device memory (raw address) and stays the caller's."""
        cs, nc = None, 0
        if contig_starts is not None:
            cs = np.ascontiguousarray(contig_starts, dtype=np.uint32)
            nc = len(cs) - 1
        self._check(self.lib.blasr_hip_adopt_genome_device(self.handle, C.c_void_p(genome_ptr), int(length),
                                                           cs.ctypes.data if cs is not None else None, nc))
        self.genome_length = int(length)

    # --- next-3: -rbao, AlignIntervals between the chained anchors (Blasr.cpp:1354-1712) ---
    def align_between_anchors(self, seq, q_off, q_len, t_off, t_len, windows, chains, chain_off, sdp=None, score=None,
                              band=16, indel=5, sdp_bypass_threshold=1000000.0, blocks_capacity=None):
        """blasr_hip_align_between_anchors_host over what Context.windows returned and the intervals' chained anchors
        (hip.find_intervals(chains=True)).  Returns (results SDP_RESULT_DTYPE, blocks (m, 3) int32) like sdp_align."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        q_off, t_off = np.ascontiguousarray(q_off, np.int64), np.ascontiguousarray(t_off, np.int64)
        q_len, t_len = np.ascontiguousarray(q_len, np.int32), np.ascontiguousarray(t_len, np.int32)
        windows = np.ascontiguousarray(windows, dtype=WINDOW_DTYPE)
        chains = np.ascontiguousarray(chains, dtype=np.uint32)
        chain_off = np.ascontiguousarray(chain_off, dtype=np.int64)
        n = len(q_off)
        sb = _SdpBatch()
        sb.n, sb.seq, sb.q_off, sb.q_len = n, seq.ctypes.data, q_off.ctypes.data, q_len.ctypes.data
        sb.t_off, sb.t_len = t_off.ctypes.data, t_len.ctypes.data
        rp = _RbaoParams()
        rp.sdp = (sdp or SdpParams()).as_struct()
        rp.score = (score or ScoreParams()).as_struct()
        rp.band, rp.indel, rp.sdp_bypass_threshold = band, indel, sdp_bypass_threshold
        cap = int(np.minimum(q_len, t_len).sum() + 64) if blocks_capacity is None else int(blocks_capacity)
        res = np.zeros(max(n, 1), dtype=SDP_RESULT_DTYPE)
        blocks = np.zeros((max(cap, 1), 3), dtype=np.int32)
        used = C.c_int64(0)
        self._check(self.lib.blasr_hip_align_between_anchors_host(
            self.handle, C.byref(sb), len(seq), windows.ctypes.data, C.c_uint32(self.genome_length), chains.ctypes.data,
            chain_off.ctypes.data, C.byref(rp), res.ctypes.data, blocks.ctypes.data, cap, C.byref(used)))
        return res[:n], blocks[: used.value]

    # --- MapRead from the intervals on (Blasr.cpp:2530-2644), C++ driver inside the library ---
    def map_params(self, **kw):
        """blasr_hip_map_params with the defaults of MappingParameters.h; keyword overrides: band, n_candidates,
        max_gap, max_score, min_pct_identity, sort_refined, extend_alignments (-extend), extend_band_size,
        max_extend_dropoff, sdp=SdpParams, score=ScoreParams, rate, extend_ends, do_global."""
        p = _MapParams()
        self.lib.blasr_hip_map_default_params(C.byref(p))
        if "sdp" in kw:
            p.sdp = kw.pop("sdp").as_struct()
        if "score" in kw:
            p.score = kw.pop("score").as_struct()
        if "rate" in kw:
            p.window.approximate_max_insertion_rate = kw.pop("rate")
        if "extend_ends" in kw:
            p.window.extend_ends = int(kw.pop("extend_ends"))
        if "do_global" in kw:
            p.window.do_global_alignment = int(kw.pop("do_global"))
        # MakeSane (MappingParameters.h:533-545) sets bandSize = 16 only WITH useGuidedAlign; under -noUseGuidedAlign the
        # reference leaves it 0 and RefineAlignment takes k from the SDP alignment's drift (Blasr.cpp:991-1001)
        if not kw.get("use_guided_align", 1):
            self.lib.blasr_hip_map_set_no_guided_align(C.byref(p), C.c_int32(1 if "band" in kw else 0))
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def map_last_timing(self):
        ms = (C.c_float * 6)()
        cnt = (C.c_int64 * 6)()
        self._check(self.lib.blasr_hip_map_last_timing(self.handle, ms, cnt))
        return dict(windows_ms=ms[0], sdp_ms=ms[1], filter1_ms=ms[2], refine_ms=ms[3], filter2_ms=ms[4], fetch_ms=ms[5],
                    intervals=cnt[0], refined=cnt[1], aligned_reads=cnt[2], aligned_bases=cnt[3], dp_cells=cnt[4],
                    hits=cnt[5])

    def map_reads(self, reads, read_pool, intervals, params=None, fetch=True, chains=None):
        """blasr_hip_map_reads_host: reads READ_DTYPE, read_pool uint8, intervals INTERVAL_DTYPE (host arrays).
        Returns (hits HIT_DTYPE array, read_hit_off, seq arena, ops arena); with fetch the hits' q_off / t_off /
        ops_off index the two arenas.  chains: (chains, chain_off) of hip.find_intervals(chains=True), for
        params.refine_between_anchors_only (-rbao)."""
        reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
        read_pool = np.ascontiguousarray(read_pool, dtype=np.uint8)
        intervals = np.ascontiguousarray(intervals, dtype=INTERVAL_DTYPE)
        params = params or self.map_params()
        n_reads, n_iv = len(reads), len(intervals)
        if chains is not None:
            ch = np.ascontiguousarray(chains[0], dtype=np.uint32)
            choff = np.ascontiguousarray(chains[1], dtype=np.int64)
            assert len(choff) == n_iv + 1
            params.iv_chains, params.iv_chain_off = ch.ctypes.data, choff.ctypes.data
        cap = max(n_iv, 1)
        hits = np.zeros(cap, dtype=HIT_DTYPE)
        off = np.zeros(n_reads + 1, dtype=np.int64)
        n_hits = C.c_int64(0)
        f, seq, ops = None, np.zeros(0, np.uint8), np.zeros(0, np.uint32)
        if fetch:
            owner = np.clip(intervals["read"].astype(np.int64), 0, max(n_reads - 1, 0))
            longest = reads["length"].astype(np.int64)[owner] if n_reads and n_iv else np.zeros(0, np.int64)
            seq_cap = int((longest * 3 + (intervals["end"].astype(np.int64) - intervals["start"])).clip(0).sum()) + 64
            seq, ops = np.zeros(seq_cap, np.uint8), np.zeros(seq_cap, np.uint32)
            f = _MapFetch(seq.ctypes.data, seq_cap, 0, ops.ctypes.data, seq_cap, 0)
        rc = self.lib.blasr_hip_map_reads_host(self.handle, n_reads, reads.ctypes.data, read_pool.ctypes.data,
                                               len(read_pool), n_iv, intervals.ctypes.data, C.byref(params),
                                               hits.ctypes.data, cap, C.byref(n_hits), off.ctypes.data,
                                               C.byref(f) if f is not None else None)
        self._check(rc)
        if f is not None:
            seq, ops = seq[: f.seq_used], ops[: f.ops_used]
        return hits[: n_hits.value], off, seq, ops

    def map_reads_device(self, n_reads, reads_ptr, read_pool_ptr, n_intervals, intervals_ptr, params, hits, read_hit_off=None):
        """Device-resident reads / pool / intervals (raw addresses); hits: a HIT_DTYPE host array to fill.
        Returns the number of hits; their offsets index the context's device arenas."""
        n_hits = C.c_int64(0)
        rc = self.lib.blasr_hip_map_reads_device(self.handle, n_reads, C.c_void_p(reads_ptr), C.c_void_p(read_pool_ptr),
                                                 n_intervals, C.c_void_p(intervals_ptr), C.byref(params),
                                                 hits.ctypes.data, len(hits), C.byref(n_hits),
                                                 read_hit_off.ctypes.data if read_hit_off is not None else None, None)
        self._check(rc)
        return n_hits.value

    def refine_from_sdp_device(self, dev_ptrs, sdp_results_ptr, blocks_ptr, n_sel, sel_ptr, band, score, results_ptr,
                               ops_ptr, ops_capacity):
        """RefineAlignment's trimming + refine on the SDP stage's own device records (raw addresses)."""
        sb = _SdpBatch()
        sb.n = int(dev_ptrs["n"])
        for k in ("seq", "q_off", "q_len", "t_off", "t_len"):
            setattr(sb, k, int(dev_ptrs[k]))
        used = C.c_int64(0)
        sc = score.as_struct()
        rc = self.lib.blasr_hip_refine_from_sdp_device(self.handle, C.byref(sb), C.c_void_p(sdp_results_ptr),
                                                       C.c_void_p(blocks_ptr), int(n_sel),
                                                       C.c_void_p(sel_ptr) if sel_ptr else None, band, C.byref(sc),
                                                       C.c_void_p(results_ptr), C.c_void_p(ops_ptr), int(ops_capacity),
                                                       C.byref(used))
        self._check(rc)
        return used.value

    # --- next-1, device half: MapReadToGenome over the suffix array (MapBySuffixArray.h:217-311) ---
    def borrow_index(self, other):
        """blasr_hip_borrow_index: anchor in the genome and the suffix array `other` (a Context on the same device) holds."""
        self.lib.blasr_hip_borrow_index.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.lib.blasr_hip_borrow_index(self.handle, other.handle))
        self._index_owner = other  # (keeps it alive)

    def upload_suffix_array(self, sa):
        """sa: what read_suffix_array returns (index, start_pos_table, end_pos_table, lookup_prefix_length)."""
        idx = np.ascontiguousarray(sa["index"], dtype=np.uint32)
        sp = np.ascontiguousarray(sa["start_pos_table"], dtype=np.uint32)
        ep = np.ascontiguousarray(sa["end_pos_table"], dtype=np.uint32)
        self._check(self.lib.blasr_hip_upload_suffix_array(self.handle, idx.ctypes.data, C.c_int64(len(idx)),
                                                           sp.ctypes.data if len(sp) else None,
                                                           ep.ctypes.data if len(ep) else None, C.c_int64(len(sp)),
                                                           int(sa["lookup_prefix_length"])))

    def build_suffix_array(self, lookup_prefix_length=8, download=True):
        """The suffix array of the uploaded genome and its lookup table, built on the device (what the reference's
        sawriter computes); it stays resident for anchor_reads.  With download: the dict read_suffix_array returns."""
        n = self.genome_length
        tl = (1 << (2 * lookup_prefix_length)) if lookup_prefix_length else 0
        idx = np.zeros(n if download else 0, np.uint32)
        sp, ep = np.zeros(tl if download else 0, np.uint32), np.zeros(tl if download else 0, np.uint32)
        self._check(self.lib.blasr_hip_build_suffix_array(self.handle, lookup_prefix_length,
                                                          idx.ctypes.data if download else None,
                                                          sp.ctypes.data if download and tl else None,
                                                          ep.ctypes.data if download and tl else None))
        if download:
            return dict(index=idx, start_pos_table=sp, end_pos_table=ep, lookup_prefix_length=lookup_prefix_length)
        return None

    def write_resident_suffix_array(self, path):
        """blasr_hip_sa_write_resident: the index resident on the device as sawriter's .sa file."""
        self.lib.blasr_hip_sa_write_resident.argtypes = [C.c_void_p, C.c_char_p]
        self._check(self.lib.blasr_hip_sa_write_resident(self.handle, path.encode()))

    def count_tuples(self, tuple_size=8):
        """The tuple count table of the uploaded genome, counted on the device (= hip.count_tuples of the same genome)."""
        table = np.zeros(1 << (2 * tuple_size), dtype=np.int32)
        self.lib.blasr_hip_count_tuples_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        self._check(self.lib.blasr_hip_count_tuples_device(self.handle, tuple_size, table.ctypes.data))
        return table

    def pipeline_params(self, reads_per_block=None, anchor=None, interval=None, map=None, overlap=0):
        p = _PipelineParams()
        self.lib.blasr_hip_pipeline_default_params.argtypes = [C.POINTER(_PipelineParams)]
        self.lib.blasr_hip_pipeline_default_params(C.byref(p))
        if anchor is not None:
            p.anchor = anchor
        if interval is not None:
            p.interval = interval
        if map is not None:
            p.map = map
        if reads_per_block:
            p.reads_per_block = int(reads_per_block)
        p.overlap = int(overlap)
        return p

    def map_from_reads_device(self, genome_host, seq_start_pos, count_table, reads, reads_ptr, read_pool_ptr, params, hits,
                              read_hit_off=None, intervals_capacity=0, fetch_capacity=0):
        """blasr_hip_map_from_reads_device: anchoring in sub-blocks overlapped with the host's interval finding, then the
        path.  genome_host / seq_start_pos / count_table: host arrays (the interval finder's); reads: host READ_DTYPE
        array, reads_ptr / read_pool_ptr: the same table and the pool on the device; hits: HIT_DTYPE host array to fill.
        Returns (number of hits, intervals INTERVAL_DTYPE array or None); with fetch_capacity (bytes of sequence = number of
        ops) also the two host arenas the hits' q_off / t_off / ops_off index."""
        genome_host = np.ascontiguousarray(genome_host, dtype=np.uint8)
        starts = np.ascontiguousarray(seq_start_pos, dtype=np.uint32)
        table = np.ascontiguousarray(count_table, dtype=np.int32)
        reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
        k = (len(table).bit_length() - 1) // 2
        ivs = np.zeros(intervals_capacity, dtype=INTERVAL_DTYPE) if intervals_capacity else None
        n_hits, n_iv = C.c_int64(0), C.c_int64(0)
        f = self.lib.blasr_hip_map_from_reads_device
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.POINTER(_PipelineParams), C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        fetch = None
        if fetch_capacity:
            seq, ops = np.zeros(fetch_capacity, np.uint8), np.zeros(fetch_capacity, np.uint32)
            fetch = _MapFetch(seq.ctypes.data, fetch_capacity, 0, ops.ctypes.data, fetch_capacity, 0)
        self._check(f(self.handle, genome_host.ctypes.data, starts.ctypes.data, len(starts), table.ctypes.data, k, len(reads),
                      reads.ctypes.data, C.c_void_p(reads_ptr), C.c_void_p(read_pool_ptr), C.byref(params), hits.ctypes.data,
                      len(hits), C.byref(n_hits), read_hit_off.ctypes.data if read_hit_off is not None else None,
                      C.byref(fetch) if fetch is not None else None,
                      ivs.ctypes.data if ivs is not None else None, intervals_capacity, C.byref(n_iv)))
        if fetch is not None:
            return n_hits.value, (ivs[: n_iv.value] if ivs is not None else None), seq[: fetch.seq_used], ops[: fetch.ops_used]
        return n_hits.value, (ivs[: n_iv.value] if ivs is not None else None)

    def pipeline_last_timing(self):
        ms = (C.c_float * 4)()
        cnt = (C.c_int64 * 2)()
        self.lib.blasr_hip_pipeline_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        self._check(self.lib.blasr_hip_pipeline_last_timing(self.handle, ms, cnt))
        return dict(anchoring_ms=ms[0], wait_for_intervals_ms=ms[1], path_ms=ms[2], host_intervals_ms=ms[3], anchors=cnt[0],
                    intervals=cnt[1])

    def anchor_params(self, **kw):
        """blasr_hip_anchor_params with the reference's defaults; keyword overrides by field name."""
        p = _AnchorParams()
        self.lib.blasr_hip_anchor_default_params(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, int(v))
        return p

    def anchor_reads(self, reads, read_pool, params=None, capacity=None):
        """blasr_hip_anchor_reads_host: (anchors ANCHOR_DTYPE array, list_off); list 2r = read r forward, 2r+1 = its
        reverse complement.  The capacity grows once when the first guess was short."""
        reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
        read_pool = np.ascontiguousarray(read_pool, dtype=np.uint8)
        params = params or self.anchor_params()
        n = len(reads)
        cap = int(capacity) if capacity is not None else 4 * int(reads["length"].astype(np.int64).sum()) + 1024
        off = np.zeros(2 * n + 1, dtype=np.int64)
        need = C.c_int64(0)
        for _ in range(2):
            out = np.zeros(max(cap, 1), dtype=ANCHOR_DTYPE)
            rc = self.lib.blasr_hip_anchor_reads_host(self.handle, n, reads.ctypes.data, read_pool.ctypes.data,
                                                      C.c_int64(len(read_pool)), C.byref(params), out.ctypes.data,
                                                      C.c_int64(cap), C.byref(need), off.ctypes.data)
            if rc == ERR_CAPACITY and capacity is None and need.value > cap:
                cap = need.value
                continue
            break
        self._check(rc)
        return out[: need.value], off

    def anchor_reads_device(self, reads, read_pool_ptr, params, anchors_ptr, capacity):
        """Device-resident read pool and anchor array (raw addresses); reads: host READ_DTYPE array.  Returns
        (number of anchors, list_off)."""
        reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
        off = np.zeros(2 * len(reads) + 1, dtype=np.int64)
        need = C.c_int64(0)
        self._check(self.lib.blasr_hip_anchor_reads_device(self.handle, len(reads), reads.ctypes.data,
                                                           C.c_void_p(read_pool_ptr), C.byref(params),
                                                           C.c_void_p(anchors_ptr), C.c_int64(capacity), C.byref(need),
                                                           off.ctypes.data))
        return need.value, off

    def find_intervals_device(self, genome, seq_start_pos, count_table, reads, anchors_ptr, list_off, params=None, chains=False):
        """blasr_hip_find_intervals_device: the anchors stay on the device (anchors_ptr: raw address of the sorted anchor
        array blasr_hip_anchor_reads_device wrote); (intervals INTERVAL_DTYPE, read_interval_off, p_values) like
        hip.find_intervals; with chains=True (blasr_hip_find_intervals_device_chains) also (chains (m, 3) uint32, chain_off)."""
        genome = np.ascontiguousarray(genome, dtype=np.uint8)
        starts = np.ascontiguousarray(seq_start_pos, dtype=np.uint32)
        table = np.ascontiguousarray(count_table, dtype=np.int32)
        k = (len(table).bit_length() - 1) // 2
        reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
        list_off = np.ascontiguousarray(list_off, dtype=np.int64)
        params = params or interval_params()
        n = len(reads)
        cap = max(n * max(int(params.n_candidates), 1), 1)
        out = np.zeros(cap, dtype=INTERVAL_DTYPE)
        pv = np.zeros(cap, dtype=np.float32)
        off = np.zeros(n + 1, dtype=np.int64)
        need = C.c_int64(0)
        self.lib.blasr_hip_find_intervals_device.argtypes = None
        self._check(self.lib.blasr_hip_find_intervals_device(
            self.handle, C.c_void_p(genome.ctypes.data), C.c_int64(len(genome)), C.c_void_p(starts.ctypes.data),
            C.c_int32(len(starts)), C.c_void_p(table.ctypes.data), C.c_int32(k), C.c_int32(n), C.c_void_p(reads.ctypes.data),
            C.c_void_p(anchors_ptr), C.c_void_p(list_off.ctypes.data), C.byref(params), C.c_void_p(out.ctypes.data),
            C.c_void_p(pv.ctypes.data), C.c_int64(cap), C.byref(need), C.c_void_p(off.ctypes.data)))
        if chains:
            ccap = max(int(list_off[-1] - list_off[0]), 1)
            ch = np.zeros((ccap, 3), dtype=np.uint32)
            coff = np.zeros(cap + 1, dtype=np.int64)
            cused = C.c_int64(0)
            self.lib.blasr_hip_find_intervals_device_chains.argtypes = None
            self._check(self.lib.blasr_hip_find_intervals_device_chains(
                self.handle, C.c_void_p(genome.ctypes.data), C.c_int64(len(genome)), C.c_void_p(starts.ctypes.data),
                C.c_int32(len(starts)), C.c_void_p(table.ctypes.data), C.c_int32(k), C.c_int32(n), C.c_void_p(reads.ctypes.data),
                C.c_void_p(anchors_ptr), C.c_void_p(list_off.ctypes.data), C.byref(params), C.c_void_p(out.ctypes.data),
                C.c_void_p(pv.ctypes.data), C.c_int64(cap), C.byref(need), C.c_void_p(off.ctypes.data),
                C.c_void_p(ch.ctypes.data), C.c_int64(ccap), C.byref(cused), C.c_void_p(coff.ctypes.data)))
            return out[: need.value], off, pv[: need.value], ch[: cused.value], coff[: need.value + 1]
        return out[: need.value], off, pv[: need.value]

    def anchor_last_timing(self):
        ms = (C.c_float * 4)()
        cnt = (C.c_int64 * 4)()
        self._check(self.lib.blasr_hip_anchor_last_timing(self.handle, ms, cnt))
        return dict(search_ms=ms[0], emit_ms=ms[1], sort_ms=ms[2], wall_ms=ms[3], positions=cnt[0], anchors=cnt[1],
                    pieces=cnt[2], probes=cnt[3])

    # --- a-11: AffineKBandAlign (the -noUseGuidedAlign refiner, Blasr.cpp:675-746) ---
    def _kband_call(self, fn, pairs, ks, params, ops_capacity):
        n = len(pairs)
        seqs, q_off, q_len, t_off, t_len, at = [], [], [], [], [], 0
        for q, t in pairs:
            q = np.ascontiguousarray(q, dtype=np.uint8)
            t = np.ascontiguousarray(t, dtype=np.uint8)
            q_off.append(at); q_len.append(len(q)); at += len(q)
            t_off.append(at); t_len.append(len(t)); at += len(t)
            seqs += [q, t]
        seq = np.ascontiguousarray(np.concatenate(seqs)) if at else np.zeros(1, np.uint8)
        q_off, t_off = np.array(q_off, np.int64), np.array(t_off, np.int64)
        q_len, t_len = np.array(q_len, np.int32), np.array(t_len, np.int32)
        ks = np.ascontiguousarray(ks, dtype=np.int32)
        kb = _KbandBatch()
        kb.n = n
        kb.seq, kb.q_off, kb.q_len = seq.ctypes.data, q_off.ctypes.data, q_len.ctypes.data
        kb.t_off, kb.t_len, kb.k = t_off.ctypes.data, t_len.ctypes.data, ks.ctypes.data
        cap = int(q_len.sum() + t_len.sum() + 2 * ks.sum() + 8 * n) if ops_capacity is None else int(ops_capacity)
        results = np.zeros(n, dtype=RESULT_DTYPE)
        ops = np.zeros(max(cap, 1), dtype=np.uint32)
        used = C.c_int64(0)
        self._check(fn(self.handle, C.byref(kb), C.byref(params), results.ctypes.data, ops.ctypes.data, cap,
                       C.byref(used)))
        out = []
        for r in results:
            o = ops[int(r["ops_off"]): int(r["ops_off"]) + int(r["n_ops"])].copy()
            blocks, gc, gi = ops_to_alignment(o)
            out.append(dict(status=int(r["status"]), score=int(r["dp_score"]), nCells=int(r["n_cells"]),
                            qPos=int(r["q_pos"]), tPos=int(r["t_pos"]), ops=o, blocks=blocks, gapCounts=gc, gapItems=gi))
        return out

    def kband(self, pairs, ks, matrix=None, indel=5, ops_capacity=None):
        """AffineKBandAlign, Global.  pairs: list of (query, target) uint8 arrays, ks: band half-widths.
        Penalties as PairwiseLocalAlign derives them from params.indel (Blasr.cpp:695-699).  Returns a list of
        dicts: status, score, nCells, ops, blocks, gapCounts, gapItems."""
        if len(pairs) == 0:
            return []
        kp = _KbandParams()
        m = np.ascontiguousarray(SMRT_MATRIX if matrix is None else matrix, dtype=np.int32).reshape(-1)
        for i, v in enumerate(m):
            kp.matrix[i] = int(v)
        kp.hp_ins_open, kp.hp_ins_extend, kp.ins_open, kp.ins_extend, kp.del_ = indel + 2, indel - 3, indel + 2, indel - 1, indel
        return self._kband_call(self.lib.blasr_hip_kband_host, pairs, ks, kp, ops_capacity)

    def kband_linear(self, pairs, ks, align_type=1, matrix=None, ins=5, dele=5, fn_ins=5, fn_del=5, ops_capacity=None):
        """KBandAlign with a DistanceMatrixScoreFunction (KBandAlign.h:74-403); align_type Global 1, QueryFit 2,
        TargetFit 3, Fit 7.  Returns dicts with score, qPos, tPos, nCells, blocks, gap lists."""
        if len(pairs) == 0:
            return []
        kp = _KbandLinearParams()
        m = np.ascontiguousarray(SMRT_MATRIX if matrix is None else matrix, dtype=np.int32).reshape(-1)
        for i, v in enumerate(m):
            kp.matrix[i] = int(v)
        kp.ins, kp.del_, kp.fn_ins, kp.fn_del, kp.align_type = ins, dele, fn_ins, fn_del, align_type
        return self._kband_call(self.lib.blasr_hip_kband_linear_host, pairs, ks, kp, ops_capacity)

    def extend(self, tasks, band=10, max_n_drops=10, matrix=None, ins=5, dele=5, min_extend_n_bases=1,
               ops_capacity=None):
        """ExtendAlignmentForward / ExtendAlignmentReverse (ExtendAlign.h:365-445), the -extend step of AlignIntervals
        (Blasr.cpp:1749-1940).  tasks: list of (query, query_pos, target, ref_pos, reverse).  Returns dicts with
        status, score (the function's return value), qPos, tPos, nCells, ops, blocks, gap lists."""
        n = len(tasks)
        if n == 0:
            return []
        ep = _ExtendParams()
        self._check(self.lib.blasr_hip_extend_default_params(C.byref(ep)))
        if matrix is not None:
            for i, v in enumerate(np.ascontiguousarray(matrix, dtype=np.int32).reshape(-1)):
                ep.matrix[i] = int(v)
        ep.ins, ep.del_, ep.band, ep.min_extend_n_bases, ep.max_n_drops = ins, dele, band, min_extend_n_bases, max_n_drops
        seqs, q_off, q_len, t_off, t_len, at = [], [], [], [], [], 0
        for q, _, t, _, _ in tasks:
            q = np.ascontiguousarray(q, dtype=np.uint8)
            t = np.ascontiguousarray(t, dtype=np.uint8)
            q_off.append(at); q_len.append(len(q)); at += len(q)
            t_off.append(at); t_len.append(len(t)); at += len(t)
            seqs += [q, t]
        seq = np.ascontiguousarray(np.concatenate(seqs)) if at else np.zeros(1, np.uint8)
        q_off, t_off = np.array(q_off, np.int64), np.array(t_off, np.int64)
        q_len, t_len = np.array(q_len, np.int32), np.array(t_len, np.int32)
        q_pos = np.array([x[1] for x in tasks], np.int32)
        t_pos = np.array([x[3] for x in tasks], np.int32)
        rev = np.array([1 if x[4] else 0 for x in tasks], np.uint8)
        eb = _ExtendBatch()
        eb.n = n
        eb.seq, eb.q_off, eb.q_len = seq.ctypes.data, q_off.ctypes.data, q_len.ctypes.data
        eb.t_off, eb.t_len = t_off.ctypes.data, t_len.ctypes.data
        eb.q_pos, eb.t_pos, eb.reverse = q_pos.ctypes.data, t_pos.ctypes.data, rev.ctypes.data
        cap = int(q_len.sum() + t_len.sum() + 8 * n) if ops_capacity is None else int(ops_capacity)
        results = np.zeros(n, dtype=RESULT_DTYPE)
        ops = np.zeros(max(cap, 1), dtype=np.uint32)
        used = C.c_int64(0)
        self._check(self.lib.blasr_hip_extend_host(self.handle, C.byref(eb), C.byref(ep), results.ctypes.data,
                                                   ops.ctypes.data, cap, C.byref(used)))
        out = []
        for r in results:
            o = ops[int(r["ops_off"]): int(r["ops_off"]) + int(r["n_ops"])].copy()
            blocks, gc, gi = ops_to_alignment(o)
            out.append(dict(status=int(r["status"]), score=int(r["dp_score"]), nCells=int(r["n_cells"]),
                            qPos=int(r["q_pos"]), tPos=int(r["t_pos"]), ops=o, blocks=blocks, gapCounts=gc, gapItems=gi))
        return out

    def last_fill4(self):
        """fillp_kernel launches of the last refine call: ms (HIP events), launches, algorithmic bytes, cells."""
        ms, n, b, c = C.c_float(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(self.lib.blasr_hip_refine_last_fill4(self.handle, C.byref(ms), C.byref(n), C.byref(b), C.byref(c)))
        return dict(ms=ms.value, launches=n.value, bytes=b.value, cells=c.value)

    def last_kernels(self):
        """Every heavy kernel of the last refine call on its own (blasr_hip_refine_last_kernels): name as rocprofv3
        prints it, HIP-event ms, launches, algorithmic bytes, band cells and candidates of exactly those launches."""
        rec = np.zeros(8, dtype=KERNEL_RECORD_DTYPE)
        self.lib.blasr_hip_refine_last_kernels.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.lib.blasr_hip_kernel_name.argtypes = [C.c_int, C.c_int]
        self.lib.blasr_hip_kernel_name.restype = C.c_char_p
        n = self.lib.blasr_hip_refine_last_kernels(self.handle, rec.ctypes.data, len(rec))
        if n < 0:
            self._check(n)
        return [dict(name=self.lib.blasr_hip_kernel_name(int(r["kind"]), int(r["ext0"])).decode(), kind=int(r["kind"]),
                     ms=float(r["ms"]), launches=int(r["launches"]), bytes=int(r["algorithmic_bytes"]),
                     cells=int(r["cells"]), candidates=int(r["candidates"])) for r in rec[:n]]

    def sdp_last_kernels(self):
        """Every heavy kernel of the last SDP call on its own (blasr_hip_sdp_last_kernels): HIP-event ms on the stream it
        ran on, launches (brackets), how many kernels ran beside it, algorithmic bytes / fragments / invocations."""
        rec = np.zeros(16, dtype=SDP_KERNEL_RECORD_DTYPE)
        self.lib.blasr_hip_sdp_last_kernels.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.lib.blasr_hip_sdp_kernel_name.argtypes = [C.c_int]
        self.lib.blasr_hip_sdp_kernel_name.restype = C.c_char_p
        n = self.lib.blasr_hip_sdp_last_kernels(self.handle, rec.ctypes.data, len(rec))
        if n < 0:
            self._check(n)
        return [dict(name=self.lib.blasr_hip_sdp_kernel_name(int(r["kind"])).decode(), kind=int(r["kind"]), ms=float(r["ms"]),
                     launches=int(r["launches"]), concurrent=int(r["concurrent"]), bytes=int(r["algorithmic_bytes"]),
                     fragments=int(r["fragments"]), invocations=int(r["invocations"])) for r in rec[:n]]

    def sdp_last_timing(self):
        ms = (C.c_float * 4)()
        cnt = (C.c_int64 * 3)()
        self._check(self.lib.blasr_hip_sdp_last_timing(self.handle, ms, cnt))
        by = (C.c_int64 * 4)()
        self._check(self.lib.blasr_hip_sdp_last_bytes(self.handle, by))
        return dict(keys_ms=ms[0], match_ms=ms[1], chain_ms=ms[2], blocks_ms=ms[3], invocations=cnt[0],
                    fragments=cnt[1], sw_fills=cnt[2], keys_bytes=by[0], match_bytes=by[1], chain_bytes=by[2],
                    blocks_bytes=by[3])

    def sdp_device(self, dev_ptrs, n, sdp, score, results_ptr, blocks_ptr, blocks_capacity):
        """Device-resident variant (raw device addresses); returns the number of block triplets produced."""
        sb = _SdpBatch()
        sb.n = n
        for k in ("seq", "q_off", "q_len", "t_off", "t_len"):
            setattr(sb, k, int(dev_ptrs[k]))
        used = C.c_int64(0)
        sp, sc = sdp.as_struct(), score.as_struct()
        rc = self.lib.blasr_hip_sdp_device(self.handle, C.byref(sb), C.byref(sp), C.byref(sc), C.c_void_p(results_ptr),
                                           C.c_void_p(blocks_ptr), int(blocks_capacity), C.byref(used))
        self._check(rc)
        return used.value

    def refine_device(self, dev_ptrs, n, band, score, results_ptr, ops_ptr, ops_capacity):
        """Device-resident variant: dev_ptrs maps the blasr_hip_refine_batch field names to raw
        device addresses (e.g. torch tensors' data_ptr()).  Returns ops used."""
        sb = _RefineBatch()
        sb.n = n
        for k in ("seq", "q_off", "q_len", "t_off", "t_len", "blocks", "blk_off", "n_blocks"):
            setattr(sb, k, int(dev_ptrs[k]))
        used = C.c_int64(0)
        sc = score.as_struct()
        rc = self.lib.blasr_hip_refine_device(self.handle, C.byref(sb), band, C.byref(sc), C.c_void_p(results_ptr),
                                              C.c_void_p(ops_ptr), int(ops_capacity), C.byref(used))
        self._check(rc)
        return used.value


# --- next-4: the reference's on-disk inputs (host code, no device involved) -----------------------------
# --- next-1, host half: anchors -> candidate intervals (host code of the library) -------------------
def alignment_is_printed(score, pct_similarity, t_aligned_seq_length, map_qv, max_score=-200, min_pct_identity=0.0,
                         min_align_length=0, min_map_qv=0):
    """The gates of PrintAlignment (Blasr.cpp:2957-2978)."""
    p = _PrintParams(int(max_score), float(min_pct_identity), int(min_align_length), int(min_map_qv))
    return bool(load_library().blasr_hip_alignment_is_printed(int(score), float(pct_similarity), int(t_aligned_seq_length),
                                                              int(map_qv), C.byref(p)))


def interval_params(**kw):
    """blasr_hip_interval_params with the reference's defaults; keyword overrides by field name."""
    p = _IntervalParams()
    load_library().blasr_hip_interval_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def count_tuples(genome, tuple_size=8):
    """TupleCountTable of the genome (Blasr.cpp:4479-4482): int32 array of 4^tuple_size counts."""
    genome = np.ascontiguousarray(genome, dtype=np.uint8)
    table = np.zeros(1 << (2 * tuple_size), dtype=np.int32)
    rc = load_library().blasr_hip_count_tuples(genome.ctypes.data, len(genome), tuple_size, table.ctypes.data)
    if rc != OK:
        raise BlasrHipError(rc, "count_tuples")
    return table


def write_count_table(path, count_table):
    """TupleCountTable::Write: the .ctab file of a count table (count_tuples' output)."""
    table = np.ascontiguousarray(count_table, dtype=np.int32)
    k = (len(table).bit_length() - 1) // 2
    lib = load_library()
    lib.blasr_hip_ctab_write.argtypes = [C.c_char_p, C.c_void_p, C.c_int32]
    if lib.blasr_hip_ctab_write(path.encode(), table.ctypes.data, k) != OK:
        raise BlasrHipError(ERR_INVALID, "ctab_write %s" % path)


def read_count_table(path):
    """TupleCountTable::Read: (counts int32 array, tuple size) of a .ctab file."""
    lib = load_library()
    lib.blasr_hip_ctab_read.restype = C.c_int64
    lib.blasr_hip_ctab_read.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
    k = C.c_int32(0)
    n = lib.blasr_hip_ctab_read(path.encode(), None, 0, C.byref(k))
    if n < 0:
        raise BlasrHipError(int(n), "ctab_read %s" % path)
    table = np.zeros(n, dtype=np.int32)
    if lib.blasr_hip_ctab_read(path.encode(), table.ctypes.data, n, C.byref(k)) != n:
        raise BlasrHipError(ERR_INVALID, "ctab_read %s" % path)
    return table, k.value


def find_intervals(genome, seq_start_pos, count_table, reads, anchors, list_off, params=None, stale=None, chains=False):
    """blasr_hip_find_intervals: (intervals INTERVAL_DTYPE, read_interval_off, p_values); with chains=True
    (blasr_hip_find_intervals_chains) two more: the intervals' chained anchors as an (m, 3) uint32 array of (t, q, l) and
    chain_off (intervals + 1 entries) -- WeightedInterval::matches, what -rbao reads."""
    lib = load_library()
    genome = np.ascontiguousarray(genome, dtype=np.uint8)
    starts = np.ascontiguousarray(seq_start_pos, dtype=np.uint32)
    table = np.ascontiguousarray(count_table, dtype=np.int32)
    k = (len(table).bit_length() - 1) // 2
    reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
    anchors = np.ascontiguousarray(anchors, dtype=ANCHOR_DTYPE)
    list_off = np.ascontiguousarray(list_off, dtype=np.int64)
    params = params or interval_params()
    st = np.ascontiguousarray(stale, dtype=ANCHOR_DTYPE) if stale is not None else None
    n = len(reads)
    cap = max(n * max(int(params.n_candidates), 1), 1)
    out = np.zeros(cap, dtype=INTERVAL_DTYPE)
    pv = np.zeros(cap, dtype=np.float32)
    off = np.zeros(n + 1, dtype=np.int64)
    need = C.c_int64(0)
    rc = lib.blasr_hip_find_intervals(genome.ctypes.data, len(genome), starts.ctypes.data, len(starts), table.ctypes.data,
                                      k, n, reads.ctypes.data, anchors.ctypes.data if len(anchors) else None,
                                      list_off.ctypes.data, st.ctypes.data if st is not None else None, C.byref(params),
                                      out.ctypes.data, pv.ctypes.data, cap, C.byref(need), off.ctypes.data)
    if rc != OK:
        raise BlasrHipError(rc, "find_intervals")
    if chains:
        ccap = max(len(anchors), 1)
        ch = np.zeros((ccap, 3), dtype=np.uint32)
        coff = np.zeros(cap + 1, dtype=np.int64)
        cused = C.c_int64(0)
        rc = lib.blasr_hip_find_intervals_chains(
            genome.ctypes.data, len(genome), starts.ctypes.data, len(starts), table.ctypes.data, k, n, reads.ctypes.data,
            anchors.ctypes.data if len(anchors) else None, list_off.ctypes.data, st.ctypes.data if st is not None else None,
            C.byref(params), out.ctypes.data, pv.ctypes.data, cap, C.byref(need), off.ctypes.data, ch.ctypes.data,
            C.c_int64(ccap), C.byref(cused), coff.ctypes.data)
        if rc != OK:
            raise BlasrHipError(rc, "find_intervals_chains")
        return out[: need.value], off, pv[: need.value], ch[: cused.value], coff[: need.value + 1]
    return out[: need.value], off, pv[: need.value]


def read_fasta(path=None, text=None):
    """FASTAReader::ReadAllSequences: [(title, sequence uint8 array)] of a FASTA file (or of `text`, bytes)."""
    lib = load_library()
    h = C.c_void_p()
    if path is not None:
        rc = lib.blasr_hip_fasta_read(path.encode(), C.byref(h))
    else:
        rc = lib.blasr_hip_fasta_parse(text, C.c_int64(len(text)), C.byref(h))
    if rc != OK:
        raise BlasrHipError(rc, "fasta_read %s" % path)
    lib.blasr_hip_fasta_count.restype = C.c_int64
    out = []
    for i in range(lib.blasr_hip_fasta_count(h)):
        t, tl, s, sl = C.c_char_p(), C.c_int64(0), C.c_void_p(), C.c_int64(0)
        lib.blasr_hip_fasta_record(h, C.c_int64(i), C.byref(t), C.byref(tl), C.byref(s), C.byref(sl))
        title = C.string_at(t, tl.value)
        seq = np.frombuffer(C.string_at(s, sl.value), dtype=np.uint8).copy() if sl.value else np.zeros(0, np.uint8)
        out.append((title, seq))
    lib.blasr_hip_fasta_free(h)
    return out


def read_sam_reads(path=None, text=None):
    """The reads of a SAM file (or text) as the reference takes a `.sam` read file (blasr_hip_sam_reads_read): list of
    (QNAME bytes, SEQ uint8 array, QUAL characters uint8 array -- empty for "*").  Raises BlasrHipError where the
    reference exits (a line it cannot parse)."""
    return read_fastq(path, text, _sam=True)


def read_fastq(path=None, text=None, _sam=False):
    """A FASTQ file (or text) as the reference's FASTQReader::GetNext reads it in this fork (blasr_hip_fastq_read: NOT a
    FASTQ parser, see include/blasr_hip.h): list of (title bytes, bases uint8 array, quality characters uint8 array).
    Raises BlasrHipError where the reference exits or reads undefined memory."""
    lib = load_library()
    h = C.c_void_p()
    if text is not None:
        rc = (lib.blasr_hip_sam_reads_parse if _sam else lib.blasr_hip_fastq_parse)(text, C.c_int64(len(text)), C.byref(h))
    else:
        rc = (lib.blasr_hip_sam_reads_read if _sam else lib.blasr_hip_fastq_read)(path.encode(), C.byref(h))
    lib.blasr_hip_fasta_error.restype = C.c_char_p
    lib.blasr_hip_fasta_error.argtypes = [C.c_void_p]
    if rc != OK:
        msg = lib.blasr_hip_fasta_error(h).decode() if h else "fastq_read"
        if h:
            lib.blasr_hip_fasta_free(h)
        raise BlasrHipError(rc, msg)
    lib.blasr_hip_fasta_count.restype = C.c_int64
    out = []
    for i in range(lib.blasr_hip_fasta_count(h)):
        t, tl, s, sl, q, ql = C.c_char_p(), C.c_int64(), C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        lib.blasr_hip_fasta_record(h, C.c_int64(i), C.byref(t), C.byref(tl), C.byref(s), C.byref(sl))
        lib.blasr_hip_fastq_quality(h, C.c_int64(i), C.byref(q), C.byref(ql))
        title = C.string_at(t, tl.value)
        seq = np.frombuffer(C.string_at(s, sl.value), dtype=np.uint8).copy() if sl.value else np.zeros(0, np.uint8)
        qual = np.frombuffer(C.string_at(q, ql.value), dtype=np.uint8).copy() if ql.value else np.zeros(0, np.uint8)
        out.append((title, seq, qual))
    lib.blasr_hip_fasta_free(h)
    return out


def read_genome(path):
    """A genome FASTA as Blasr.cpp holds it: (genome uint8 array: contigs back to back, an N after each, upper case;
    seqStartPos uint32 array; names)."""
    lib = load_library()
    h = C.c_void_p()
    rc = lib.blasr_hip_fasta_read(path.encode(), C.byref(h))
    if rc != OK:
        raise BlasrHipError(rc, "fasta_read %s" % path)
    lib.blasr_hip_fasta_count.restype = C.c_int64
    lib.blasr_hip_fasta_concatenate.restype = C.c_int64
    lib.blasr_hip_fasta_name.restype = C.c_int64
    n = lib.blasr_hip_fasta_count(h)
    starts = np.zeros(n + 1, dtype=np.uint32)
    total = lib.blasr_hip_fasta_concatenate(h, 1, None, C.c_int64(0), starts.ctypes.data_as(C.c_void_p))
    genome = np.zeros(max(total, 1), dtype=np.uint8)
    got = lib.blasr_hip_fasta_concatenate(h, 1, genome.ctypes.data_as(C.c_void_p), C.c_int64(total), None)
    names = []
    for i in range(n):
        buf = C.create_string_buffer(4096)
        lib.blasr_hip_fasta_name(h, C.c_int64(i), buf, C.c_int64(4096))
        names.append(buf.value.decode())
    lib.blasr_hip_fasta_free(h)
    if got < 0:
        raise BlasrHipError(int(got), "fasta_concatenate")
    return genome[:total], starts, names


def write_suffix_array(path, sa):
    """SuffixArray::Write as sawriter calls it: the index array and the lookup table of `sa` (a read_suffix_array dict)."""
    idx = np.ascontiguousarray(sa["index"], dtype=np.uint32)
    sp = np.ascontiguousarray(sa["start_pos_table"], dtype=np.uint32)
    ep = np.ascontiguousarray(sa["end_pos_table"], dtype=np.uint32)
    rc = load_library().blasr_hip_sa_write(path.encode(), idx.ctypes.data, len(idx), sp.ctypes.data if len(sp) else None,
                                           ep.ctypes.data if len(ep) else None, len(sp), int(sa["lookup_prefix_length"]))
    if rc != OK:
        raise BlasrHipError(rc, "sa_write %s" % path)


def read_suffix_array(path):
    """SuffixArray::Read: dict(index, start_pos_table, end_pos_table, lookup_prefix_length) of a .sa file."""
    lib = load_library()
    h = C.c_void_p()
    rc = lib.blasr_hip_sa_read(path.encode(), C.byref(h))
    if rc != OK:
        raise BlasrHipError(rc, "sa_read %s" % path)
    idx, sp, ep = C.c_void_p(), C.c_void_p(), C.c_void_p()
    n, ltl, lpl = C.c_int64(0), C.c_int64(0), C.c_int32(0)
    lib.blasr_hip_sa_arrays(h, C.byref(idx), C.byref(n), C.byref(sp), C.byref(ep), C.byref(ltl), C.byref(lpl))
    take = lambda p, k: np.frombuffer(C.string_at(p, 4 * k), dtype=np.uint32).copy() if k else np.zeros(0, np.uint32)
    out = dict(index=take(idx, n.value), start_pos_table=take(sp, ltl.value), end_pos_table=take(ep, ltl.value),
               lookup_prefix_length=lpl.value)
    lib.blasr_hip_sa_free(h)
    return out
