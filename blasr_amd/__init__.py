"""blasr_amd -- MI355X-native refine stage of the BLASR long-read mapper.

The product is ``libblasr_hip.so`` (hand-written HIP for gfx950 behind the C
ABI of ``include/blasr_hip.h``).  This package is the thin Python host side
used by the tests and ``bench.py``: ctypes bindings plus helpers that mirror
the reference's call surface (``RefineAlignments``, alignment/Blasr.cpp:2164).
It contains no CPU implementation of the algorithms and never imports
``oracle/``: if the HIP library or a gfx950 device is missing, calls raise.
"""
import os as _os

# The library's streams need more hardware queues than the HIP runtime's default of four (blasr_hip_create says why); the
# variable is read when the runtime starts, which a Python host does on its first torch.cuda / HIP call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .hip import (  # noqa: F401,E402
    BlasrHipError,
    Context,
    LIB_PATH,
    RefineBatch,
    RefineResult,
    ScoreParams,
    SdpParams,
    load_library,
    remove_low_quality,
    remove_overlapping,
    sort_candidates,
    cigar,
    m4_line,
    map_qvs,
    order_gaps_by_type,
    sam_line,
    ref_contig_location,
    CANDIDATE_DTYPE,
    HIT_DTYPE,
    INTERVAL_DTYPE,
    READ_DTYPE,
    WINDOW_DTYPE,
)

from .mapread import Hit, candidate_intervals, m4_record, map_reads, sam_record  # noqa: F401,E402

__all__ = ["BlasrHipError", "Context", "LIB_PATH", "RefineBatch", "RefineResult", "ScoreParams", "SdpParams", "load_library"]
