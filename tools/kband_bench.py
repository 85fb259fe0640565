"""The k-band refiner on its own: N synthetic (query, target) pairs of one length through blasr_hip_kband_host at one
band half-width; prints the kernels' HIP-event times and cells/s.  `python tools/kband_bench.py --reads 512 --len 30000 --k 2700`
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=512)
    ap.add_argument("--len", type=int, default=30000)
    ap.add_argument("--k", type=int, default=2700)
    ap.add_argument("--waves", type=int, default=0, help="wavefronts per task (0: the library's choice)")
    ap.add_argument("--workspace-gb", type=float, default=100.0)
    ap.add_argument("--check", type=int, default=0, help="compare this many results with the oracle")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--k-spread", type=float, default=0.0, help="k of task i = k * exp(N(0, spread)): the bands of a real batch differ")
    args = ap.parse_args()
    if args.waves:
        os.environ["BLASR_HIP_KBAND_WAVES"] = str(args.waves)
    import blasr_amd
    from blasr_amd import synth
    rng = np.random.default_rng(7)
    distinct = min(args.reads, 64)  # the kernels' time does not depend on the bases; the generator's does
    base = [synth.make_pair(rng, args.len, ins=0.10, dele=0.05, sub=0.01, flank=(0, 0)) for _ in range(distinct)]
    pairs = [base[i % distinct] for i in range(args.reads)]
    ks = [args.k] * len(pairs)
    if args.k_spread > 0:
        ks = [max(10, int(args.k * float(np.exp(rng.normal(0, args.k_spread))))) for _ in pairs]
    ctx = blasr_amd.Context(0)
    ctx.set_workspace_limit(int(args.workspace_gb * (1 << 30)))
    out = []
    for it in range(args.repeat):
        t0 = time.time()
        res = ctx.kband(pairs, ks)
        wall = time.time() - t0
        recs = ctx.last_kernels()
        for r in recs:
            r["gcells_per_s"] = r["cells"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else None
        out.append(dict(wall_s=wall, kernels=recs))
    assert all(r["status"] == 0 for r in res)
    if args.check:
        from blasr_testlib import affine_kband_align, load_oracle
        oracle = load_oracle()
        for i in range(min(args.check, distinct)):
            w = affine_kband_align(oracle, pairs[i][0], pairs[i][1], args.k)
            assert res[i]["score"] == w["score"] and np.array_equal(res[i]["blocks"], w["blocks"]), i
    print(json.dumps(dict(reads=args.reads, len=args.len, k=args.k, runs=out)))


if __name__ == "__main__":
    main()
