"""Experiment: one batch through one context vs its two halves through two contexts (two host threads,
two HIP streams) at the same time.  Prints ms per pass for both."""
import sys, threading, time
import torch
sys.path.insert(0, ".")
import blasr_amd
from blasr_amd import synth


SDP_LOCK, REF_LOCK = threading.Lock(), threading.Lock()
USE_LOCKS = False


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class Half:
    def __init__(self, reads, seed, dev):
        self.ctx = blasr_amd.Context(0)
        self.ctx.set_workspace_limit(120 << 30)
        self.w = w = synth.make_sdp_workload(reads, 10000, seed, dev)
        self.n = n = int(w["q_len"].numel())
        self.sdp_ptrs = {k: w[k].data_ptr() for k in ("seq", "q_off", "q_len", "t_off", "t_len")}
        self.sdp_res = torch.zeros(n * 48, dtype=torch.uint8, device=dev)
        self.blk_cap = int(torch.minimum(w["q_len"], w["t_len"]).sum().item() // 4 + n)
        self.blocks = torch.zeros(3 * self.blk_cap, dtype=torch.int32, device=dev)
        self.ref_res = torch.zeros(n * 56, dtype=torch.uint8, device=dev)
        self.ops_cap = int((w["q_len"].sum().item() + w["t_len"].sum().item()) // 2 + 1024)
        self.ops = torch.zeros(self.ops_cap, dtype=torch.int32, device=dev)
        self.score, self.par = blasr_amd.ScoreParams(), blasr_amd.SdpParams()
        self.stream = torch.cuda.Stream()

    def step(self, what="both"):
        n, w = self.n, self.w
        with torch.cuda.stream(self.stream):
            if what == "refine":
                self.ctx.refine_device(self.ptrs, n, 16, self.score, self.ref_res.data_ptr(), self.ops.data_ptr(),
                                       self.ops_cap)
                return
            with (SDP_LOCK if USE_LOCKS else _Null()):
                self.ctx.sdp_device(self.sdp_ptrs, n, self.par, self.score, self.sdp_res.data_ptr(),
                                    self.blocks.data_ptr(), self.blk_cap)
            r = self.sdp_res.view(torch.int32).view(n, 12)
            q_pos, t_pos, n_blk = r[:, 2].to(torch.int64), r[:, 3].to(torch.int64), r[:, 4].contiguous()
            blk_off = self.sdp_res.view(torch.int64).view(n, 6)[:, 5].contiguous()
            last = (blk_off + n_blk - 1).clamp_min(0)
            b3 = self.blocks.view(-1, 3)
            has = n_blk > 0
            q_len2 = torch.where(has, b3[last, 0] + b3[last, 2], torch.zeros_like(n_blk)).contiguous()
            t_len2 = torch.where(has, b3[last, 1] + b3[last, 2], torch.zeros_like(n_blk)).contiguous()
            q_off2 = (w["q_off"] + q_pos).contiguous()
            t_off2 = (w["t_off"] + t_pos).contiguous()
            ptrs = dict(seq=w["seq"].data_ptr(), q_off=q_off2.data_ptr(), q_len=q_len2.data_ptr(),
                        t_off=t_off2.data_ptr(), t_len=t_len2.data_ptr(), blocks=self.blocks.data_ptr(),
                        blk_off=blk_off.data_ptr(), n_blocks=n_blk.data_ptr())
            self.stream.synchronize()
            self.ptrs, self.keep = ptrs, (q_len2, t_len2, q_off2, t_off2, blk_off, n_blk)
            if what == "sdp":
                return
            with (REF_LOCK if USE_LOCKS else _Null()):
                self.ctx.refine_device(ptrs, n, 16, self.score, self.ref_res.data_ptr(), self.ops.data_ptr(),
                                       self.ops_cap)


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    what = sys.argv[3] if len(sys.argv) > 3 else "both"
    one = Half(total, 3, dev)
    one.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        one.step(what)
    torch.cuda.synchronize()
    print("one context, %d reads, %s: %.1f ms per pass" % (total, what, (time.perf_counter() - t0) / 2 * 1e3))
    print("  fill4", one.ctx.last_fill4())
    del one
    torch.cuda.empty_cache()
    halves = [Half(total // parts, 3 + i, dev) for i in range(parts)]

    for h in halves:
        h.step()

    def run(h, k):
        for _ in range(k):
            h.step(what)

    global USE_LOCKS
    USE_LOCKS = len(sys.argv) > 4
    for k in (1, 3):
        ths = [threading.Thread(target=run, args=(h, k)) for h in halves]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        if k == 3:
            print("%d contexts, %d reads each, concurrently%s: %.1f ms per pass of all" %
                  (parts, total // parts, " (stage locks)" if USE_LOCKS else "", (time.perf_counter() - t0) / 3 * 1e3))
            for h in halves:
                print("  fill4", h.ctx.last_fill4(), {k2: round(v, 1) for k2, v in h.ctx.last_timing().items()},
                      {k2: round(v, 1) for k2, v in h.ctx.sdp_last_timing().items() if k2.endswith("ms")})


main()
