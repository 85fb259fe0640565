#!/usr/bin/env python3
"""The drop-in as a maintainer would run it, with a number: oracle/_ref/blasrmc_gpu -useGPU -gpuBatch B -nproc {1, 4}
(the reference's own Blasr.cpp + INTEGRATION.md's patch, one library call per block of B reads) beside the unpatched
oracle/_ref/blasrmc -nproc N on the same FASTA files and the same .sa index: wall time, reads/s, and the -m 4 lines
compared as sorted sets.  The genome is synthetic (i.i.d. with planted repeat families), its suffix array is built on
the device and written with blasr_hip_sa_write (byte-identical to sawriter's file, tests/test_sa_build_gpu.py).

    python tools/dropin_bench.py [--genome-len 200000000] [--reads 20000] [--read-len 10000] [--nproc 32] [--batch 4096]
One JSON line."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-len", type=int, default=200000000)
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--families", type=int, default=300)
    ap.add_argument("--nproc", type=int, default=0, help="threads of the unpatched binary (0 = min(32, cores))")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--gpu-nproc", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--switches", nargs="*", default=[], help="more switches for BOTH binaries, without their dash (e.g. rbao, extend)")
    ap.add_argument("--reference-mapping-s", type=float, default=0.0,
                    help="tuning runs: do not run the unpatched binary again, take this as its mapping time (no lines compared)")
    args = ap.parse_args()
    import torch

    import blasr_amd
    from blasr_amd import hip, synth
    dev = torch.device("cuda", 0)
    nproc = args.nproc or min(32, os.cpu_count() or 1)
    genome, starts, copies = synth.make_repeat_genome([args.genome_len], 1, dev, families=args.families)
    wl = synth.make_map_workload(genome, copies, args.reads, args.read_len, args.seed, dev, ins=0.11, dele=0.04, sub=0.01,
                                 max_unanchored=300, spurious=0.0, loose=0.0, spurious_span=None)
    g = genome.cpu().numpy()
    out = dict(genome_bases=int(len(g)), reads=args.reads, read_len=args.read_len, batch=args.batch, switches=args.switches)
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "genome.fasta")
        with open(fa, "wb") as f:
            f.write(b">chr1\n")
            f.write(g.tobytes())
            f.write(b"\n")
        t0 = time.perf_counter()
        ctx = blasr_amd.Context(0)
        ctx.upload_genome(np.concatenate([g, np.frombuffer(b"N", np.uint8)]), None)
        hip.write_suffix_array(fa + ".sa", ctx.build_suffix_array(13))
        ctx.close()
        out["index_build_and_write_s"] = round(time.perf_counter() - t0, 1)
        reads = wl["reads"]
        pool = wl["pool"][: int(reads["off"][-1] + reads["length"][-1])].cpu().numpy()
        with open(os.path.join(tmp, "reads.fasta"), "wb") as f:
            for r in range(len(reads)):
                s = pool[int(reads["off"][r]): int(reads["off"][r]) + int(reads["length"][r])]
                f.write(b">read%d/%d/0_%d\n" % (r, r, len(s)))
                f.write(s.tobytes())
                f.write(b"\n")
        open(os.path.join(tmp, "none.fasta"), "w").write(">none/0/0_4\nACGT\n")
        del genome, wl
        torch.cuda.empty_cache()

        trace = []

        def run(exe, reads_file, extra, tag):
            path = os.path.join(tmp, tag + ".m4")
            t0 = time.perf_counter()
            p = subprocess.run([exe, os.path.join(tmp, reads_file), fa, "-sa", fa + ".sa", "-m", "4", "-bestn", "1", "-out", path] + extra +
                               ["-" + x for x in args.switches],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, BLASR_HIP_DROPIN_TRACE="1"))
            trace.extend(x for x in p.stdout.decode().splitlines() if x.startswith("[dropin]") or x.startswith("[rbao]"))
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return dt, None, p.stdout.decode()[-400:]
            return dt, sorted(open(path).read().splitlines()), ""

        ref_exe = os.path.join(ROOT, "oracle", "_ref", "blasrmc")
        gpu_exe = os.path.join(ROOT, "oracle", "_ref", "blasrmc_gpu")
        t_none, _, _ = run(ref_exe, "none.fasta", ["-nproc", "1"], "none")
        if args.reference_mapping_s > 0:
            t_ref, ref_lines, err = t_none + args.reference_mapping_s, None, "not run (--reference-mapping-s)"
        else:
            t_ref, ref_lines, err = run(ref_exe, "reads.fasta", ["-nproc", str(nproc)], "ref")
        out["index_load_s"] = round(t_none, 1)
        out["reference"] = dict(nproc=nproc, wall_s=round(t_ref, 1), mapping_s=round(t_ref - t_none, 1),
                                reads_per_s=round(args.reads / max(t_ref - t_none, 1e-9), 1),
                                lines=None if ref_lines is None else len(ref_lines), error=err)
        out["dropin"] = []
        for n in args.gpu_nproc:
            t_g0, _, _ = run(gpu_exe, "none.fasta", ["-useGPU", "-gpuBatch", str(args.batch), "-nproc", "1"], "gnone")
            t_g, lines, err = run(gpu_exe, "reads.fasta", ["-useGPU", "-gpuBatch", str(args.batch), "-nproc", str(n)], "gpu%d" % n)
            d = dict(nproc=n, wall_s=round(t_g, 1), zero_read_run_s=round(t_g0, 1), mapping_s=round(t_g - t_g0, 1),
                     reads_per_s=round(args.reads / max(t_g - t_g0, 1e-9), 1), lines=None if lines is None else len(lines), error=err)
            if lines is not None and ref_lines is not None:
                d["lines_that_differ"] = len(set(lines) ^ set(ref_lines))
            if lines is not None:
                d["speedup_over_reference"] = round((t_ref - t_none) / max(t_g - t_g0, 1e-9), 1)
            d["trace"] = trace[-40:] if os.environ.get("BLASR_HIP_RBAO_TRACE") else trace[-12:]
            del trace[:]
            out["dropin"].append(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
