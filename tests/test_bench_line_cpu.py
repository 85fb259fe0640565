"""The driver reads the LAST stdout line of bench.py out of an 8 KB tail: that line must be one compact JSON object
(< 4 KB) whatever the run measured; everything else goes to earlier lines and to the detail file.  These tests feed
bench.py's formatter full records (a committed one of round 4's default line -- 20 KB, the one the driver could not parse --
and a synthetic worst case) without a GPU."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def full_record():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_v6_bench_default_line.json")))


def test_the_round_4_record_fits_the_drivers_line():
    rec = full_record()
    assert len(json.dumps(rec)) > 8192  # what went wrong in round 4
    text = bench.compact_line(rec, "gpurun_out/bench_detail.json")
    assert len(text) < bench.LINE_LIMIT <= 4096 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "build_id"):
        assert k in line, k
    assert "workload" in line["config"] and "model" not in line["config"]
    rl = line["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"):
        assert k in rl, k
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-4
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["parity_mismatches"] == 0 and cb["sample"]
    # the other configurations: a five-field stub each
    assert set(line["other_configs"]) == set(rec["other_configs"])
    for name, stub in line["other_configs"].items():
        assert set(stub) == {"ms_per_step", "value", "roofline_frac", "roofline_kernel", "parity_mismatches", "parity_checked"}
        assert abs(stub["ms_per_step"] - rec["other_configs"][name]["ms_per_step"]) < 1e-3 * stub["ms_per_step"]
    assert abs(line["value"] - rec["value"]) < 1e-4 * rec["value"]


def test_a_worst_case_record_still_fits():
    rec = full_record()
    rec["config"]["workload"] = "w" * 3000
    rec["cpu_baseline"]["sample"] = "s" * 3000
    rec["n_gpus"] = 8
    rec["per_rank_ms_per_step"] = [dict(step=600.123456, windows=1.0, sdp=2.0, filters_after_sdp=3.0, refine=4.0, filters=5.0,
                                        candidates_refined=186000.0)] * 8
    rec["other_configs"]["a failed one"] = {"error": "exit 1: " + "x" * 2000}
    rec["roofline_of_the_largest_refine_kernel"] = dict(rec["roofline"])
    text = bench.compact_line(rec, "gpurun_out/bench_detail.json")
    assert len(text) < 4096
    line = json.loads(text)
    assert line["roofline"] and line["cpu_baseline"]["parity_checked"] == rec["cpu_baseline"]["parity_checked"]


def test_emit_prints_the_compact_line_last_and_writes_the_detail(tmp_path):
    rec = full_record()
    path = str(tmp_path / "sub" / "detail.json")
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(rec, path)
    lines = buf.getvalue().splitlines()
    assert len(lines[-1]) < 4096 and json.loads(lines[-1])["value"] > 0
    # exactly one line starts like a JSON object: the driver's (and the tests') way of finding it
    assert sum(1 for ln in lines if ln.startswith("{")) == 1
    assert any(ln.startswith("[bench detail] other_configs: ") for ln in lines[:-1])
    assert json.load(open(path)) == rec


def test_no_record_means_an_error_not_an_oversized_line():
    rec = full_record()
    rec["other_configs"] = {"configuration %d %s" % (i, "n" * 60): rec["other_configs"]["configs[1]"] for i in range(40)}
    try:
        bench.compact_line(rec)
    except ValueError as e:
        assert "limit" in str(e)
    else:
        raise AssertionError("a line beyond the limit was returned")


def test_emit_falls_back_to_a_minimal_line_instead_of_losing_the_run(tmp_path):
    """A finished run always ends with a line the driver can parse: what does not fit is dropped down to the contract's fields."""
    rec = full_record()
    rec["other_configs"] = {"configuration %d %s" % (i, "n" * 60): rec["other_configs"]["configs[1]"] for i in range(40)}
    rec["metric"] = rec["metric"] + " " + "m" * 5000
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(rec, str(tmp_path / "detail.json"))
    last = buf.getvalue().splitlines()[-1]
    assert len(last) < bench.LINE_LIMIT
    line = json.loads(last)
    assert line["truncated"] and abs(line["value"] - rec["value"]) < 1e-4 * rec["value"]
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "config",
              "roofline", "cpu_baseline", "build_id"):
        assert k in line, k
    assert abs(line["roofline"]["frac"] - rec["roofline"]["frac"]) < 1e-5 and line["cpu_baseline"]["parity_mismatches"] == 0
    # a malformed optional part is no reason either
    rec2 = full_record()
    rec2["other_configs"] = ["not", "a", "dict"]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(rec2, None)
    assert json.loads(buf.getvalue().splitlines()[-1])["truncated"].startswith("compact_line failed")
