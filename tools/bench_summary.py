"""One line out of a bench.py JSON: ms/step, value, fill stage ms, fillp launch ms, parity mismatches."""
import json,sys
txt=open(sys.argv[1]).read().strip()
try:
    d=json.loads(txt)  # a detail record (bench.py --detail-out)
except ValueError:
    d=json.loads(txt.splitlines()[-1])  # bench.py's stdout: the driver's line is the last
print(d["ms_per_step"], d["value"], d["stage_ms_per_step"]["refine_fill"], d["roofline"]["avg_launch_ms"], (d.get("cpu_baseline") or {}).get("parity_mismatches"))
