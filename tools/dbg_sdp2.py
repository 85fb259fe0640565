import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, blasr_amd
from blasr_amd import synth
from blasr_testlib import load_oracle
o = load_oracle(); ctx = blasr_amd.Context(0)
if len(sys.argv) > 1: os.environ["BLASR_HIP_SDP_DEBUG"] = "/tmp/sdp_dbg.txt"
rng = np.random.default_rng(5)
for L in (300, 800):
    q, t = synth.make_pair(rng, L)
    w = o.sdp_align(q,t)
    for rep in range(1):
        g = ctx.sdp([(q,t)])[0]
        print(L, len(q), len(t), "rep", rep, "got blocks", len(g["blocks"]), "want", len(w["blocks"]), "score", g["score"], w["score"], ctx.sdp_last_timing()["invocations"], ctx.sdp_last_timing()["fragments"], ctx.sdp_last_timing()["sw_fills"])
