"""CPU-side checks of the C ABI: the shared library loads, exports every symbol
include/blasr_hip.h declares, and refuses to run without a gfx950 device (there
is no CPU fallback to fall into)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import blasr_amd
from blasr_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "blasr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(blasr_hip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = blasr_amd.load_library()
    names = _declared_functions()
    assert len(names) >= 9
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(hip.EXPORTS) == names
    assert lib.blasr_hip_abi_version() == 6


def test_result_record_layout_matches_header():
    text = open(os.path.join(ROOT, "include", "blasr_hip.h")).read()
    body = text[text.index("typedef struct blasr_hip_refine_result {"):text.index("} blasr_hip_refine_result;")]
    fields = re.findall(r"int(32|64)_t\s+([a-z_]+);", body)
    assert [f[1] for f in fields] == list(hip.RESULT_DTYPE.names)
    assert sum(int(f[0]) // 8 for f in fields) == hip.RESULT_DTYPE.itemsize


def test_hit_record_layout_matches_header():
    text = open(os.path.join(ROOT, "include", "blasr_hip.h")).read()
    body = text[text.index("typedef struct blasr_hip_hit {"):text.index("} blasr_hip_hit;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for ty, names in re.findall(r"\b(u?int32_t|int64_t|float)\s+([a-z_, ]+);", body):
        fields += [(ty, n.strip()) for n in names.split(",")]
    assert [f[1] for f in fields] == list(hip.HIT_DTYPE.names)
    assert sum(8 if f[0] == "int64_t" else 4 for f in fields) == hip.HIT_DTYPE.itemsize
    p = hip._MapParams()
    assert blasr_amd.load_library().blasr_hip_map_default_params(C.byref(p)) == 0
    assert (p.band, p.n_candidates, p.max_gap, p.max_score, p.sdp.word_size, p.score.affine_open) == (16, 10, 0, -200, 11, 50)
    assert p.sdp.indel_rate == np.float32(0.3) * np.float32(3)  # params.indelRate * 3 in single precision


def test_anchor_records_match_header():
    text = open(os.path.join(ROOT, "include", "blasr_hip.h")).read()
    body = text[text.index("typedef struct blasr_hip_anchor_params {"):text.index("} blasr_hip_anchor_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [n for _, n in re.findall(r"\b(u?int32_t)\s+([a-z_]+);", body)]
    assert names == [f[0] for f in hip._AnchorParams._fields_] and C.sizeof(hip._AnchorParams) == 4 * len(names)
    assert hip.ANCHOR_DTYPE.itemsize == 12 and hip.ANCHOR_DTYPE.names == ("t", "q", "l")
    p = hip._AnchorParams()
    assert blasr_amd.load_library().blasr_hip_anchor_default_params(C.byref(p)) == 0
    # MappingParameters.h:244 (minMatchLength 14), :243 (lookupTableLength 8), :309 (stopMappingOnceUnique), AnchorParameters.h:34-37
    assert (p.min_match_length, p.min_prefix_match_length, p.expand, p.max_anchors_per_position, p.advance_exact_matches,
            p.max_lcp_length, p.use_lookup_table, p.stop_mapping_once_unique, p.forward_only, p.sorted) == (
        14, 8, 0, 1000, 0, 0, 1, 1, 0, 1)


def test_no_device_is_a_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(blasr_amd.BlasrHipError) as e:
        blasr_amd.Context(0)
    assert e.value.code == hip.ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    """The shipped path must not import, link or shell out to oracle/."""
    pkg = os.path.join(ROOT, "blasr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                code = "\n".join(l for l in src.splitlines() if "never imports" not in l and "no CPU" not in l)
                assert "libblasr_oracle" not in code and "blasr_oracle.h" not in code, f
                assert "oracle/" not in code or f == "__init__.py", f


def test_batch_packing_roundtrip():
    b = blasr_amd.RefineBatch()
    b.add(np.frombuffer(b"ACGT", np.uint8), np.frombuffer(b"ACGTT", np.uint8), [[0, 0, 4]])
    b.add(np.frombuffer(b"GG", np.uint8), np.frombuffer(b"GGA", np.uint8), [[0, 0, 1], [1, 1, 1]])
    a = b.arrays()
    assert a["seq"].tobytes() == b"ACGTACGTTGGGGA"
    assert a["q_off"].tolist() == [0, 9] and a["t_off"].tolist() == [4, 11]
    assert a["blk_off"].tolist() == [0, 1] and a["n_blocks"].tolist() == [1, 2]
    assert a["blocks"].tolist() == [0, 0, 4, 0, 0, 1, 1, 1, 1]
    assert b.ops_bound() == 4 + 5 + 2 + 3 + 4


def test_ops_decode_to_blocks_and_gaps():
    rec = np.zeros(1, dtype=hip.RESULT_DTYPE)[0]
    ops = np.array([(5 << 2) | 0, (2 << 2) | 1, (1 << 2) | 2, (3 << 2) | 0], dtype=np.uint32)
    blocks, gc, gi = blasr_amd.RefineResult(rec, ops).blocks_and_gaps()
    assert blocks.tolist() == [[0, 0, 5], [7, 6, 3]]
    assert gc.tolist() == [0, 2, 0]
    assert gi.tolist() == [[hip.GAP_TARGET, 2], [hip.GAP_QUERY, 1]]


def test_no_exception_crosses_the_abi_when_host_memory_runs_out():
    """no exception crosses the C ABI.  A child process caps its address space (RLIMIT_AS) and asks the host
    side of the interval finder -- std::vectors per read, a pool of std::threads -- for more than that: the call comes
    back with BLASR_HIP_ERR_NOMEM (every entry point is a function-try-block, the pool's threads catch in their own body
    and fall back to the caller's thread when no thread can be had), the process does not abort."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, resource, sys
import numpy as np
sys.path.insert(0, %r)
import blasr_amd
from blasr_amd import hip
lib = blasr_amd.load_library()
n = 12000000                                  # reads: the call sizes several vectors of vectors by it
genome = np.frombuffer(b"ACGT" * 64, dtype=np.uint8)
starts = np.array([0, len(genome)], dtype=np.uint32)
table = np.zeros(1 << 16, dtype=np.int32)
reads = np.zeros(n, dtype=hip.READ_DTYPE)
list_off = np.zeros(2 * n + 1, dtype=np.int64)
out = np.zeros(16, dtype=hip.INTERVAL_DTYPE)
off = np.zeros(n + 1, dtype=np.int64)
need = C.c_int64(0)
params = hip.interval_params(threads=8)
used = int(open("/proc/self/statm").read().split()[0]) * resource.getpagesize()
resource.setrlimit(resource.RLIMIT_AS, (used + (96 << 20), used + (96 << 20)))
rc = lib.blasr_hip_find_intervals(genome.ctypes.data, len(genome), starts.ctypes.data, len(starts), table.ctypes.data, 8, n,
                                  reads.ctypes.data, None, list_off.ctypes.data, None, C.byref(params), out.ctypes.data, None, 16,
                                  C.byref(need), off.ctypes.data)
print("rc", rc)
''' % ROOT
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stderr.decode()[-2000:])  # not SIGABRT from std::terminate
    assert "rc -4" in p.stdout.decode(), (p.stdout.decode(), p.stderr.decode()[-2000:])
