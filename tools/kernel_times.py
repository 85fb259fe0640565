import json,sys
txt=open(sys.argv[1]).read().strip()
try:
    d=json.loads(txt)  # a detail record (bench.py --detail-out)
except ValueError:
    d=json.loads(txt.splitlines()[-1])  # bench.py's stdout: the driver's line is the last
print(sys.argv[1], round(d["ms_per_step"],1), "fill", round(d["stage_ms_per_step"]["refine_fill"],1), "tb", round(d["stage_ms_per_step"]["refine_traceback_stats"],1), "sdp", round(d["stage_ms_per_step"]["sdp"],1), "parity", (d.get("cpu_baseline") or {}).get("parity_mismatches"))
for r in d["kernel_rooflines"]: print("   ", r["kernel"], round(r["avg_launch_ms"],2))
