#!/usr/bin/env python3
"""What a launch of fillh8_kernel costs under a placement of its candidates, from the per-slot profile the library dumps
(BLASR_HIP_DUMP_HYBRID=<file>: per slot {narrow passes of 40 columns, narrow rows, wide rows, wide steps of 192 columns, rows,
cells}).  A wavefront holds eight candidates: its narrow rows last as long as the candidate with the most passes, its wide
rows are taken one after the other.  Model: cost = max(passes) * P + sum(wide steps) * S + sum(wide rows) * R (wavefront
instructions); the device runs `slots` wavefronts side by side, taken in list order.

    python tools/hybrid_order.py gpurun_out/r06_dump/hybrid_c2.bin
"""
import heapq
import sys

import numpy as np

P, S, R = 207.0, 125.0, 35.0


def makespan(costs, slots=4096):
    """wavefronts handed out in order to `slots` places: when the last one ends"""
    h = [0.0] * min(slots, len(costs))
    heapq.heapify(h)
    for c in costs:
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)


def report(name, groups, prof):
    passes, wsteps, wrows = prof[:, 0], prof[:, 3], prof[:, 2]
    costs, idle = [], 0.0
    for g in groups:
        g = [i for i in g if i >= 0]
        if not g:
            continue
        mx = passes[g].max()
        idle += (mx * len(g) - passes[g].sum()) + mx * (8 - len(g))
        costs.append(mx * P + wsteps[g].sum() * S + wrows[g].sum() * R)
    costs = np.array(costs)
    total = costs.sum()
    ideal = passes.sum() / 8 * P + wsteps.sum() * S + wrows.sum() * R
    print("%-34s wavefronts %6d  work %.3g (ideal %.3g, +%.1f%%)  idle narrow lane-passes %.1f%%  makespan/avg-per-slot %.3f  longest/mean %.2f" % (
        name, len(costs), total, ideal, 100 * (total / ideal - 1), 100 * idle / (passes.sum() + idle),
        makespan(list(costs)) / (total / 4096), costs.max() / costs.mean()))
    return total, makespan(list(costs))


def main():
    prof = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 6)
    n = len(prof)
    live = np.nonzero(prof[:, 4] > 0)[0]
    print("%d slots, %d candidates; narrow passes %.3g (%.2f per narrow row), wide rows %.3g, wide steps %.3g" % (
        n, len(live), prof[:, 0].sum(), prof[:, 0].sum() / max(prof[:, 1].sum(), 1), prof[:, 2].sum(), prof[:, 3].sum()))
    G = 8
    report("as launched", [list(range(w * G, min((w + 1) * G, n))) for w in range((n + G - 1) // G)], prof)
    # alternatives over the live candidates
    cells, passes = prof[live, 5], prof[live, 0]
    W = (len(live) + G - 1) // G

    def deal(order):  # p -> wavefront p mod W
        groups = [[] for _ in range(W)]
        for p, i in enumerate(order):
            groups[p % W].append(live[i])
        return groups

    def chunks(order):  # consecutive eights
        return [list(live[order[w * G:(w + 1) * G]]) for w in range(W)]
    report("dealt by cells", deal(np.argsort(-cells, kind="stable")), prof)
    report("consecutive by passes", chunks(np.argsort(-passes, kind="stable")), prof)
    o = np.argsort(-passes, kind="stable")
    g = chunks(o)
    wcost = [passes[[np.where(live == i)[0][0] for i in grp]].max() * P + prof[grp, 3].sum() * S + prof[grp, 2].sum() * R for grp in g]
    g2 = [g[i] for i in np.argsort(-np.array(wcost), kind="stable")]
    report("consecutive by passes, heaviest first", g2, prof)
    # a queue: every 8-lane group takes the next candidate when it is done (narrow work never idles; wide work serialises per wavefront)
    total = passes.sum() / 8 * P + prof[:, 3].sum() * S + prof[:, 2].sum() * R
    print("%-34s work %.3g" % ("a queue of candidates (ideal)", total))


if __name__ == "__main__":
    main()
