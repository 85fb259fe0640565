#!/usr/bin/env python3
"""next-3 measured: blasr_hip_extend_device on the shapes AlignIntervals hands ExtendAlignmentForward / Reverse under
-extend (two tasks per alignment: up to 500 bases of read end against up to 500 of genome, band 10, 10 drops), inputs
resident in HBM, with the reference's own templates (oracle/_ref/libblasr_ref.so, else the oracle; one host thread)
timed beside it on a bounded sample of the same tasks and every result of that sample compared.  One JSON line.

    python tools/extend_bench.py [--tasks 200000] [--steps 5] [--cpu-tasks 2000] [--clean]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", type=int, default=200000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-tasks", type=int, default=2000)
    ap.add_argument("--clean", action="store_true", help="1 %% error: extensions run the full 500 rows")
    args = ap.parse_args()
    import torch
    import blasr_amd
    from blasr_amd import hip, synth
    import blasr_testlib as T

    rng = np.random.default_rng(3)
    n_unique = min(args.tasks, 4000)  # distinct read ends; the batch repeats them
    err = (0.01, 0.01, 0.01) if args.clean else (0.11, 0.04, 0.01)
    uniq = []
    while len(uniq) < n_unique:
        q, t = synth.make_pair(rng, int(rng.integers(20, 520)), ins=err[0], dele=err[1], sub=err[2], flank=(0, 40))
        if len(q) == 0 or len(t) == 0:
            continue
        q, t = q[:500], t[:500]
        if len(uniq) % 2:
            uniq.append((q[::-1].copy(), len(q) - 1, t[::-1].copy(), len(t) - 1, True))
        else:
            uniq.append((q, 0, t, 0, False))
    seqs, q_off, q_len, t_off, t_len, at = [], [], [], [], [], 0
    for q, _, t, _, _ in uniq:
        q_off.append(at); q_len.append(len(q)); at += len(q)
        t_off.append(at); t_len.append(len(t)); at += len(t)
        seqs += [q, t]
    pick = np.arange(args.tasks) % n_unique
    dev = torch.device("cuda:0")
    d = dict(seq=torch.from_numpy(np.concatenate(seqs)).to(dev),
             q_off=torch.from_numpy(np.array(q_off, np.int64)[pick]).to(dev),
             q_len=torch.from_numpy(np.array(q_len, np.int32)[pick]).to(dev),
             t_off=torch.from_numpy(np.array(t_off, np.int64)[pick]).to(dev),
             t_len=torch.from_numpy(np.array(t_len, np.int32)[pick]).to(dev),
             q_pos=torch.from_numpy(np.array([u[1] for u in uniq], np.int32)[pick]).to(dev),
             t_pos=torch.from_numpy(np.array([u[3] for u in uniq], np.int32)[pick]).to(dev),
             reverse=torch.from_numpy(np.array([1 if u[4] else 0 for u in uniq], np.uint8)[pick]).to(dev))
    cap = int((np.array(q_len)[pick].astype(np.int64) + np.array(t_len)[pick]).sum()) + 8 * args.tasks
    results = torch.zeros(args.tasks * hip.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    ops = torch.zeros(cap, dtype=torch.int32, device=dev)
    ctx = blasr_amd.Context(0)
    lib = ctx.lib
    ep = hip._ExtendParams()
    lib.blasr_hip_extend_default_params(C.byref(ep))
    eb = hip._ExtendBatch()
    eb.n = args.tasks
    for key in ("seq", "q_off", "q_len", "t_off", "t_len", "q_pos", "t_pos", "reverse"):
        setattr(eb, key, d[key].data_ptr())
    used = C.c_int64(0)

    def step():
        rc = lib.blasr_hip_extend_device(ctx.handle, C.byref(eb), C.byref(ep), results.data_ptr(), ops.data_ptr(), cap,
                                         C.byref(used))
        ctx._check(rc)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    res = np.frombuffer(results.cpu().numpy().tobytes(), dtype=hip.RESULT_DTYPE)
    host_ops = ops.cpu().numpy().view(np.uint32)
    cells = int(res["n_cells"].astype(np.int64).sum())
    # the reference on one host thread, bounded sample, compared
    checker = T.load_ref() or T.load_oracle()
    kind = "reference" if T.load_ref() is not None else "port"
    n_cpu = min(args.cpu_tasks, args.tasks)
    t0 = time.perf_counter()
    want = [T.extend_align(checker, *uniq[pick[i]][:4], 10, 10, uniq[pick[i]][4]) for i in range(n_cpu)]
    cpu_s = time.perf_counter() - t0
    bad = 0
    for i in range(n_cpu):
        r = res[i]
        o = host_ops[int(r["ops_off"]): int(r["ops_off"]) + int(r["n_ops"])]
        blocks, gc, gi = hip.ops_to_alignment(o)
        w = want[i]
        ok = ((int(r["dp_score"]), int(r["q_pos"]), int(r["t_pos"])) == (w["score"], w["qPos"], w["tPos"])
              and np.array_equal(blocks, w["blocks"]) and np.array_equal(gc, w["gapCounts"])
              and np.array_equal(gi, w["gapItems"]))
        bad += not ok
    ext = res["n_ops"] > 0
    print(json.dumps(dict(metric="extensions_per_second", value=args.tasks / (ms * 1e-3), ms_per_step=ms,
                          tasks=args.tasks, cells=cells, cells_per_s=cells / (ms * 1e-3),
                          extended=int(ext.sum()), ops=int(used.value), mismatches=bad, compared=n_cpu,
                          cpu=dict(kind=kind, cores=1, tasks_per_s=n_cpu / cpu_s, sample=n_cpu),
                          clean=bool(args.clean))))
    ctx.close()


if __name__ == "__main__":
    main()
