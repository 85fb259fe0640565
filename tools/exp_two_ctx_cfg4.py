"""What N contexts on N host threads (the reference's -nproc N: one MappingBuffers / one context per mapping thread) make of
a configs[4] block: the reads in N contiguous shares, each mapped by its own context, against one context mapping all of them.
    python tools/exp_two_ctx_cfg4.py [reads] [noguided]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blasr_amd
from blasr_amd import hip, synth
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
noguided = len(sys.argv) > 2
genome, starts, copies = synth.make_repeat_genome([synth.ECOLI_K12_LENGTH], 1, dev, families=0)
wl = synth.make_map_workload(genome, None, n, 30000, 3, dev, ins=0.10, dele=0.05, sub=0.01, max_unanchored=300, spurious=8.0,
                             loose=0.0, spurious_span=(0.15, 0.9, 0.55, 1.3))
G = int(genome.numel())
reads, iv = wl["reads"], wl["intervals"]
def sub(lo, hi):
    m = (iv["read"] >= lo) & (iv["read"] < hi)
    s = iv[m].copy(); s["read"] -= lo
    r = reads[lo:hi].copy()
    base = int(r["off"][0]); r["off"] -= base
    pool = wl["pool"][base: base + int(r["off"][-1] + r["length"][-1])]
    return r, s, pool
def make(lo, hi, limit):
    r, s, pool = sub(lo, hi)
    ctx = blasr_amd.Context(0); ctx.set_workspace_limit(limit)
    ctx.adopt_genome_device(genome.data_ptr(), G, None)
    rt = torch.from_numpy(r.view(np.uint8)).to(dev); st = torch.from_numpy(s.view(np.uint8)).to(dev)
    hits = np.zeros(len(s), dtype=hip.HIT_DTYPE)
    params = ctx.map_params(band=0, use_guided_align=0) if noguided else ctx.map_params()
    def step():
        return ctx.map_reads_device(len(r), rt.data_ptr(), pool.data_ptr(), len(s), st.data_ptr(), params, hits)
    return step, (ctx, rt, st, pool, hits)
torch.cuda.synchronize()
for parts in (1, 2, 3, 4):
    cuts = [n * i // parts for i in range(parts + 1)]
    steps = [make(cuts[i], cuts[i + 1], (200 << 30) // parts) for i in range(parts)]
    def run():
        ths = [threading.Thread(target=s[0]) for s in steps]
        for t in ths: t.start()
        for t in ths: t.join()
    run()
    t0 = time.perf_counter(); run(); run(); dt = (time.perf_counter() - t0) / 2
    print("contexts %d: %.1f ms per pass" % (parts, dt * 1e3), flush=True)
    for s in steps: s[1][0].close()
    del steps
