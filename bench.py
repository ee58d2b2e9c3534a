#!/usr/bin/env python3
"""Headline bench: contigs/s over eref + generateGraph + matching on the 1M-contig synthetic
(BASELINE.json `metric`), inputs resident in HBM, one process per GPU.

  python bench.py [--gpus N --steps K --warmup W] [--contigs 1000000]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over the whole synthetic sample:
  eref:   zero the count table, count every read of both FASTQ sides, scan every phage ref
  (generateGraph and matching stages are added to the step as they land; `config.stages` names
   what the printed number covers.)
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant
kernel (live HIP-event timing on the kernel's own stream) and `cpu_baseline` (the oracle, i.e.
the CPU restatement of the reference algorithm, timed on a bounded sample on this host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20261003
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
READ_LEN = 150


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contigs", type=int, default=1_000_000)
    ap.add_argument("--refs", type=int, default=5000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=40000)
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------
# synthetic sample, generated on the device (SURVEY.md section 8(d) shapes)
# ----------------------------------------------------------------------------------------------
def make_sample(torch, dev, n_contigs, n_refs, rank=0, world=1):
    g = torch.Generator(device=dev)
    g.manual_seed(SEED)
    rng = np.random.Generator(np.random.PCG64(SEED))
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    comp[torch.tensor([65, 67, 71, 84], device=dev)] = torch.tensor([84, 71, 67, 65], dtype=torch.uint8, device=dev)

    def dna(n):
        out = torch.empty(n, dtype=torch.uint8, device=dev)
        step = 1 << 28
        for s in range(0, n, step):
            m = min(step, n - s)
            out[s:s + m] = lut[torch.randint(0, 4, (m,), generator=g, device=dev)]
        return out

    scale = n_contigs / 1_000_000
    # phage DB: n_refs refs, length U[20 kb, 60 kb]
    ref_lens = rng.integers(20000, 60001, size=n_refs).astype(np.int64)
    ref_off = np.zeros(n_refs + 1, dtype=np.int64)
    np.cumsum(ref_lens, out=ref_off[1:])
    ref_bases = dna(int(ref_off[-1]))
    # contigs: log-normal lengths (median 800, sigma 1, min 56); the read pool
    c_lens = np.maximum(56, rng.lognormal(np.log(800.0), 1.0, size=n_contigs)).astype(np.int64)
    c_off = np.zeros(n_contigs + 1, dtype=np.int64)
    np.cumsum(c_lens, out=c_off[1:])
    pool = dna(int(c_off[-1]))
    # reads: sum(fq1 bases) = 5e8 per 1M contigs (keeps E3 in the keep-everything regime)
    n_pairs = int(5e8 * scale) // READ_LEN
    n_phage = n_pairs // 10                       # ~12x over 200 "present" refs
    present = rng.choice(n_refs, size=max(1, int(200 * min(1.0, n_refs / 5000))), replace=False)
    ar = torch.arange(READ_LEN, device=dev)

    def cut(src, starts):
        out = torch.empty((len(starts), READ_LEN), dtype=torch.uint8, device=dev)
        step = 1 << 20
        for s in range(0, len(starts), step):
            st = starts[s:s + step]
            out[s:s + len(st)] = src[st[:, None] + ar[None, :]]
        return out

    def with_errors(reads, rate):
        m = torch.rand(reads.shape, generator=g, device=dev) < rate
        sub = lut[torch.randint(0, 4, reads.shape, generator=g, device=dev)]
        return torch.where(m, sub, reads)

    # pool pairs: fragment inside one contig when it fits, else clipped to the pool end
    ins = torch.clamp(torch.normal(400.0, 40.0, (n_pairs,), generator=g, device=dev), READ_LEN, 800).long()
    pool_n = n_pairs - n_phage
    p_start = (torch.rand(pool_n, generator=g, device=dev, dtype=torch.float64) * (len(pool) - 1000)).long()
    pr = torch.from_numpy(ref_off[present]).to(dev)
    pl = torch.from_numpy(ref_lens[present]).to(dev)
    which = torch.randint(0, len(present), (n_phage,), generator=g, device=dev)
    f_start = pr[which] + (torch.rand(n_phage, generator=g, device=dev, dtype=torch.float64)
                           * (pl[which] - 900).double()).long()
    r1 = torch.cat([cut(pool, p_start), with_errors(cut(ref_bases, f_start), 0.005)])
    r2_pool = cut(pool, p_start + ins[:pool_n] - READ_LEN)
    r2_ph = with_errors(cut(ref_bases, f_start + ins[pool_n:] - READ_LEN), 0.005)
    r2 = comp[torch.cat([r2_pool, r2_ph]).flip(1).long()]
    perm = torch.randperm(n_pairs, generator=g, device=dev)
    r1, r2 = r1[perm].contiguous(), r2[perm].contiguous()
    if world > 1:                                  # reads shard by record range across ranks
        lo, hi = n_pairs * rank // world, n_pairs * (rank + 1) // world
        r1, r2 = r1[lo:hi].contiguous(), r2[lo:hi].contiguous()
    n_loc = r1.shape[0]
    read_off = torch.arange(n_loc + 1, device=dev, dtype=torch.int64) * READ_LEN
    del pool
    return dict(n_contigs=n_contigs, n_refs=n_refs, ref_bases=ref_bases,
                ref_off=torch.from_numpy(ref_off).to(dev), ref_total=int(ref_off[-1]), ref_lens=ref_lens,
                r1=r1.reshape(-1), r2=r2.reshape(-1), read_off=read_off, n_reads_side=n_loc,
                n_pairs_total=n_pairs, present=np.sort(present))


# ----------------------------------------------------------------------------------------------
def cpu_baseline(torch, sample, header, n_reads):
    """The oracle (CPU restatement of the reference algorithm, 1 thread) on a bounded sample,
    extrapolated linearly to the whole workload.  Returns contigs/s and a description."""
    from oracle import binding as orc
    cc = orc.header_to_cc(header)
    n_side = min(n_reads // 2, sample["n_reads_side"])
    b1 = sample["r1"][: n_side * READ_LEN].cpu().numpy()
    b2 = sample["r2"][: n_side * READ_LEN].cpu().numpy()
    off = np.arange(n_side + 1, dtype=np.int64) * READ_LEN
    n_ref_s = max(1, sample["n_refs"] // 100)
    ro = sample["ref_off"][: n_ref_s + 1].cpu().numpy()
    rb = sample["ref_bases"][: int(ro[-1])].cpu().numpy()
    idx = [orc.index_ref(rb[ro[i]:ro[i + 1]], cc) for i in range(n_ref_s)]   # cached index: not timed
    table = orc.CountTable()
    t0 = time.perf_counter()
    table.clear()                                   # extract_ref.cpp:1257 (fixed cost, not scaled)
    t_clear = time.perf_counter() - t0
    t0 = time.perf_counter()
    table.count(b1, off, cc)
    table.count(b2, off, cc)
    t_reads = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(n_ref_s):
        orc.scan_ref(idx[i], int(ro[i + 1] - ro[i]), table, 0.9, 0.85)
    t_refs = time.perf_counter() - t0
    table.free()
    total_reads = 2 * sample["n_pairs_total"]
    t_full = t_clear + t_reads * total_reads / (2 * n_side) + t_refs * sample["n_refs"] / n_ref_s
    return dict(value=sample["n_contigs"] / t_full, unit="contigs/s", cores=1, kind="port",
                sample=(f"oracle/eref_oracle.c, 1 thread: {2 * n_side} of {total_reads} reads x{READ_LEN} bp "
                        f"({t_reads:.1f} s) + 4 GiB table memset ({t_clear:.1f} s, fixed) + scan of {n_ref_s} of {sample['n_refs']} refs ({t_refs:.2f} s), "
                        f"extrapolated linearly; stages: eref only (4 GiB byte table as the reference, "
                        f"dead 16 GiB Peaks memset excluded)"),
                reads_per_s=2 * n_side / t_reads)


def main():
    args = parse_args()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from palace_amd import capi
    from oracle import binding as orc       # header helper + cpu_baseline leg only

    hdr = orc.header_from_picks(np.random.Generator(np.random.PCG64(SEED)).integers(0, 6, size=32))
    ctx = capi.Ctx(local)
    ctx.eref_set_coder(hdr)
    sample = make_sample(torch, dev, args.contigs, args.refs, rank, world)
    torch.cuda.synchronize()
    one_min, three_min = capi.window_minimums(0.9, 0.85)
    rows = torch.zeros((sample["n_refs"], 4), dtype=torch.int32, device=dev)
    rows_host = torch.zeros((sample["n_refs"], 4), dtype=torch.int32).pin_memory()
    L = capi.lib()
    P = lambda t: t.data_ptr()
    n_side = sample["n_reads_side"]

    def step(i, timed):
        m = 4 * i
        capi._check(L.palace_eref_table_reset(ctx.h), "reset")
        if timed: ctx.mark(m)
        capi._check(L.palace_eref_count_reads(ctx.h, P(sample["r1"]), P(sample["read_off"]), n_side, None), "count")
        capi._check(L.palace_eref_count_reads(ctx.h, P(sample["r2"]), P(sample["read_off"]), n_side, None), "count")
        if timed: ctx.mark(m + 1)
        capi._check(L.palace_eref_scan_refs(ctx.h, P(sample["ref_bases"]), P(sample["ref_off"]), sample["n_refs"],
                                            sample["ref_total"], one_min, three_min, P(rows)), "scan")
        if timed: ctx.mark(m + 2)
        capi._check(L.palace_d2h(ctx.h, rows_host.data_ptr(), P(rows), rows.numel() * 4), "d2h")

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for w in range(args.warmup):
        step(0, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_step = 1e3 * dt / args.steps
    count_ms = np.mean([ctx.mark_elapsed(4 * i, 4 * i + 1) for i in range(args.steps)]) / 2    # two launches
    scan_ms = np.mean([ctx.mark_elapsed(4 * i + 1, 4 * i + 2) for i in range(args.steps)])
    r = rows_host.numpy()
    reported = int(((r[:, 1] > 0) & (r[:, 1].astype(np.float32) / r[:, 2].astype(np.float32) > 0.75)).sum())

    if rank == 0:
        alg_bytes = (READ_LEN + 6 * (READ_LEN - 31)) * n_side            # per launch (one FASTQ side)
        achieved = alg_bytes / (count_ms * 1e-3) / 1e9
        out = {
            "metric": "contigs/sec eref+generate_graph+matching, 1M-contig synth",
            "value": args.contigs / (ms_step * 1e-3), "unit": "contigs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{args.contigs}-contig synthetic sample: {sample['n_refs']} phage refs "
                                   f"({sample['ref_total']} bp), {2 * sample['n_pairs_total']} reads x {READ_LEN} bp",
                       "stages": ["eref"], "seed": SEED, "refs_reported": reported,
                       "refs_present": int(len(sample["present"]))},
            "roofline": {"bound": "hbm", "kernel": "eref_count_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": count_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "stage_ms": {"eref_count_both_sides": 2 * count_ms, "eref_scan_refs": scan_ms},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, sample, hdr, args.cpu_sample_reads)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
