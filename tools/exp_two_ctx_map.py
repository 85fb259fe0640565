"""Potential of two internal pipelines: the configs[2] block mapped as two halves by two host threads, each through
its own context, against one context mapping all of it."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blasr_amd
from blasr_amd import hip, synth
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
genome, starts, copies = synth.make_repeat_genome(list(synth.HG19_CHROM_LENGTHS), 1, dev, families=4000)
wl = synth.make_map_workload(genome, copies, n, 10000, 3, dev)
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
    ctx.adopt_genome_device(genome.data_ptr(), G, starts.astype(np.uint32))
    rt = torch.from_numpy(r.view(np.uint8)).to(dev); st = torch.from_numpy(s.view(np.uint8)).to(dev)
    hits = np.zeros(len(s), dtype=hip.HIT_DTYPE)
    params = ctx.map_params()
    def step():
        return ctx.map_reads_device(len(r), rt.data_ptr(), pool.data_ptr(), len(s), st.data_ptr(), params, hits)
    return step, (ctx, rt, st, pool, hits)
torch.cuda.synchronize()
for parts in (1, 2, 3):
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
