import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, blasr_amd
from blasr_amd import synth
from blasr_testlib import load_oracle
o = load_oracle(); ctx = blasr_amd.Context(0)
rng = np.random.default_rng(5)
os.environ["BLASR_HIP_SDP_DEBUG"] = "/tmp/sdp_dbg.txt"
q0, t0 = synth.make_pair(rng, 300)
q, t = synth.make_pair(rng, 800)
w = o.sdp_align(q,t)
del os.environ["BLASR_HIP_SDP_DEBUG"]
os.environ["BLASR_HIP_SDP_TRACE"] = "1"
for rep in range(2):
    ctx.sdp([(q0,t0)]); r = ctx.sdp([(q,t)])[0]; print("rep", rep, len(r["blocks"]), file=sys.stderr)
sys.exit(0)
for attempt in range(1):
    g = ctx.sdp([(q,t)])[0]
    ok = np.array_equal(g["blocks"], w["blocks"])
    print("attempt", attempt, "blocks got/want", len(g["blocks"]), len(w["blocks"]), ok, ctx.sdp_last_timing()["invocations"], ctx.sdp_last_timing()["sw_fills"])
    if not ok: break
gb = set(map(tuple, (g["blocks"] + [g["qPos"], g["tPos"], 0]).tolist())); wb = set(map(tuple, (w["blocks"] + [w["qPos"], w["tPos"], 0]).tolist()))
print("missing", sorted(wb - gb)[:12]); print("extra", sorted(gb - wb)[:12])
lines = open("/tmp/sdp_dbg.txt").read().split("\n")
nf, end = map(int, lines[0].split())
d = np.array([[int(v) for v in l.split()] for l in lines[1:1+nf]])
print("nf", nf, "chainEnd", end)
# expected fragments by brute force (k=11 middle; prefix/suffix k=5 zones 50)
def kmers(s, k):
    out = {}
    for i in range(len(s)-k+1):
        w_ = bytes(s[i:i+k])
        if all(c in b"ACGT" for c in w_): out.setdefault(w_, []).append(i)
    return out
exp = {}
tk = kmers(t, 11)
for x in range(len(q)-10):
    for y in tk.get(bytes(q[x:x+11]), []): exp[(x,y)] = ("M", 11)
P = 50
tp = kmers(t[:P], 5)
for x in range(min(len(q),P)-4):
    for y in tp.get(bytes(q[x:x+5]), []):
        exp.setdefault((x,y), ("P",5))
ts = kmers(t[len(t)-P:], 5); qs0 = len(q)-P
for x in range(P-4):
    for y in ts.get(bytes(q[qs0+x:qs0+x+5]), []):
        exp.setdefault((qs0+x, len(t)-P+y), ("S",5))
keys = sorted(exp)
got = [(int(a),int(b)) for a,b in d[:,:2]]
print("frag count exp", len(keys), "got", len(got), "same set", keys == got)
xyl = np.array([[x,y,exp[(x,y)][1]] for x,y in keys], dtype=np.int32)
print("len mismatch", int((xyl[:,2] != d[:,2]).sum()) if len(keys)==len(got) else "n/a")
chain = o.sdp_chain(len(q), xyl, 11, 5, 10, -5, 0)
# device chain by backtracking
dc = []; f = end
while f != -1: dc.append(f); f = int(d[f,4])
dc = dc[::-1]
print("chain len oracle", len(chain), "device", len(dc), "equal", list(chain)==dc)
# brute-force predCl / above
k=11
bad_cl = 0
for i,(x,y) in enumerate(keys):
    best=None
    for j,(px,py) in enumerate(keys[:i]):
        if px < x-k-1 or px >= x: continue
        dg = py-px
        if dg <= y-x:
            if best is None or (dg, px) > best[0]: best=((dg,px), j)
    e = best[1] if best else -1
    if e != d[i,5]: bad_cl += 1
print("predCl mismatches", bad_cl)
order = sorted(range(len(keys)), key=lambda i:(keys[i][1], keys[i][0]))
ab = [-1]*len(keys)
for a_,b_ in zip(order[:-1], order[1:]):
    ax,ay = keys[a_]; bx,by = keys[b_]
    if ax <= bx and ax + xyl[a_,2] > bx and ay < by: ab[b_] = a_
print("above mismatches", int((np.array(ab) != d[:,6]).sum()))
for i in range(min(len(chain), len(dc))):
    if chain[i] != dc[i]:
        print("first chain diff at", i, chain[i], dc[i]); break
