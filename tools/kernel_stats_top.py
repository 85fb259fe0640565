"""Top kernels of a rocprofv3 --kernel-trace --stats CSV, with short names:  python tools/kernel_stats_top.py k_kernel_stats.csv [N] [filter]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
flt = sys.argv[3] if len(sys.argv) > 3 else ""
for r in rows:
    name = r["Name"]
    short = re.sub(r"\(anonymous namespace\)::|blasr_hip::|void |rocprim::ROCPRIM_\d+_NS::detail::", "", name).split("(")[0][:70]
    if flt and not re.search(flt, name):
        continue
    print("%-72s calls %6s total %9.2f ms avg %9.3f ms %5s%%" % (short, r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                                float(r["AverageNs"]) / 1e6, r["Percentage"][:5]))
    n -= 1
    if n <= 0:
        break
