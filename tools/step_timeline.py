"""Timeline of the LAST step of a bench run from a rocprofv3 --kernel-trace CSV: every kernel of the library (and
rocPRIM / copy kernels between them) in start order with its start offset inside the step, duration and the idle gap
before it; consecutive launches of the same kernel are merged.  Then the same per SDP level.
    python tools/step_timeline.py <kernel_trace.csv> <ms_per_step> [min_ms_to_print]"""
import csv
import sys


def short(n):
    n = n.replace("void ", "").replace("blasr_hip::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:60]


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1]))]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Kernel_Name"]) for r in rows)
    lib = [e for e in ev if "blasr_hip" in e[3] or "rocprim" in e[3] or "(anonymous namespace)::" in e[3]]
    t1 = max(e[1] for e in lib)
    t0 = t1 - int(float(sys.argv[2]) * 1e6)
    floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
    seg = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    merged = []
    for s, e, n, _ in seg:
        if merged and merged[-1][2] == n and s - merged[-1][1] < 200000:
            merged[-1][1] = max(merged[-1][1], e)
            merged[-1][3] += 1
            merged[-1][4] += e - s
        else:
            merged.append([s, e, n, 1, e - s])
    end = t0
    level = -1
    per_level = {}
    for s, e, n, calls, busy in merged:
        if n.startswith("sdp_classify"):
            level += 1
        if n.startswith("sdp_") or n.startswith("sw_"):
            d = per_level.setdefault(level, {})
            d[n] = d.get(n, 0) + busy / 1e6
        gap = (s - end) / 1e6
        if busy / 1e6 >= floor or gap >= floor:
            print("%9.2f  %8.3f ms  x%-4d gap %7.3f  %s" % ((s - t0) / 1e6, busy / 1e6, calls, max(gap, 0), n))
        end = max(end, e)
    print()
    for lv in sorted(per_level):
        tot = sum(per_level[lv].values())
        print("SDP level launch %d: %.1f ms" % (lv, tot))
        for n, v in sorted(per_level[lv].items(), key=lambda kv: -kv[1])[:14]:
            print("      %8.2f  %s" % (v, n))


if __name__ == "__main__":
    main()
