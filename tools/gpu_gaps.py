"""Idle gaps of the GPU inside the last pass of a bench run, from a rocprofv3 --kernel-trace CSV:
python tools/gpu_gaps.py <kernel_trace.csv> <ms_per_step> [n_gaps]
The last pass is taken as the ms_per_step before the end of the last blasr_hip kernel."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]) for r in rows)
t1 = max(e[1] for e in ev if "blasr_hip" in e[2] or "rocprim" in e[2])
t0 = t1 - int(float(sys.argv[2]) * 1e6)
seg = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy, cur_end, gaps, prev = 0, t0, [], "(pass start)"
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n, cur_end - t0))
        busy += e - s
        cur_end, prev = e, n
    elif e > cur_end:
        busy += e - cur_end
        cur_end, prev = e, n
print("window %.2f ms, busy %.2f ms, idle %.2f ms in %d gaps, %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), len(seg)))
for g in sorted(gaps, reverse=True)[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%8.3f ms at %8.2f  after %-50s before %s" % (g[0] / 1e6, g[3] / 1e6, g[1], g[2]))
