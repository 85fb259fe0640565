import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, blasr_amd, collections
from blasr_amd import synth
from blasr_testlib import load_oracle
o = load_oracle(); ctx = blasr_amd.Context(0)
rng = np.random.default_rng(5)
pairs = [synth.make_pair(rng, L) for L in (300, 800, 1200)]
want = [o.sdp_align(q,t) for q,t in pairs]
stat = collections.Counter()
for rep in range(60):
    i = rep % 3
    g = ctx.sdp([pairs[i]])[0]
    tm = ctx.sdp_last_timing()
    ok = np.array_equal(g["blocks"], want[i]["blocks"])
    stat[(i, ok, tm["invocations"], tm["fragments"], tm["sw_fills"], len(g["blocks"]))] += 1
for k,v in sorted(stat.items()): print(k, v)
