"""One scattered invocation at a time through the sweep kernels: the latency of the serial chain per fragment, kernel by kernel.
   python tools/chain_bench.py            (on the GPU box)
Global SDPAlign calls on random rectangles (no recursion) as the recursion issues them; BLASR_HIP_CHAIN_SCATTER picks the
kernel for every scattered invocation: 2 = sdp_chain_bits_kernel (a wavefront each), 3 = sdp_chain_rows_kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import blasr_amd
from blasr_amd import synth

ctx = blasr_amd.Context(0)
cases = [(7, 8000, 40000), (7, 20000, 60000), (5, 3000, 12000), (5, 6000, 10000), (5, 1500, 2500), (7, 3000, 6000)]
out = []
for k, qn, tn in cases:
    q, t = synth.random_genome(qn, 100 + qn), synth.random_genome(tn, 200 + tn)
    sp = blasr_amd.SdpParams(k=k, align_type=1, prefix=0, recurse=0)
    row = {"k": k, "q": qn, "t": tn}
    ref = None
    for mode in ("2", "3"):
        os.environ["BLASR_HIP_CHAIN_SCATTER"] = mode
        ctx.sdp([(q, t)], sdp=sp)
        best = None
        for _ in range(3):
            got = ctx.sdp([(q, t)], sdp=sp)
            ks = [r for r in ctx.sdp_last_kernels() if "chain" in r["name"]]
            ms = sum(r["ms"] for r in ks)
            best = ms if best is None else min(best, ms)
            frags = sum(r["fragments"] for r in ks)
        row["mode%s_ms" % mode] = round(best, 3)
        row["fragments"] = frags
        row["mode%s_us_per_fragment" % mode] = round(best * 1e3 / max(frags, 1), 3)
        sig = (got[0]["score"], len(got[0]["blocks"]))
        assert ref is None or ref == sig, (ref, sig)
        ref = sig
    row["fragments_per_row"] = round(row["fragments"] / qn, 2)
    out.append(row)
    print(json.dumps(row), flush=True)
