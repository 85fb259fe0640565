"""One line out of a bench.py JSON: ms/step, value, fill stage ms, fillp launch ms, parity mismatches."""
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["stage_ms_per_step"]["refine_fill"], d["roofline"]["avg_launch_ms"], (d.get("cpu_baseline") or {}).get("parity_mismatches"))
