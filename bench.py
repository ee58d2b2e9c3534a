#!/usr/bin/env python3
"""Headline bench: contigs/s over eref + generateGraph + matching on the 1M-contig synthetic
(BASELINE.json `metric`), inputs resident in HBM, one process per GPU.

  python bench.py [--gpus N --steps K --warmup W] [--contigs 1000000] [--workload default|long]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(`python bench.py --gpus N` without a launcher starts its N ranks itself, as child processes, before this process
touches a GPU, relays rank 0's JSON line and exits with their status.)

One "step" = one full pass of the hot path over the whole synthetic sample:
  eref:   zero the count table, count every read of both FASTQ sides, scan every phage ref
  (generateGraph and matching stages are added to the step as they land; `config.stages` names
   what the printed number covers.)
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant
kernel (live HIP-event timing on the kernel's own stream) and `cpu_baseline` (the oracle, i.e.
the CPU restatement of the reference algorithm, timed on a bounded sample on this host).
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import threading
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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("default", "long"), default="default",
                    help="default: BASELINE configs[1..3] shape (log-normal contigs, median 800 bp); long: configs[4] "
                         "(100k contigs, N50 ~ 50 kb, tail > 120 kb, evidence that reaches the exp-underflow gate)")
    ap.add_argument("--contigs", type=int, default=None)
    ap.add_argument("--refs", type=int, default=5000)
    ap.add_argument("--reads", choices=("packed", "ascii"), default="packed",
                    help="form of the reads resident in HBM: packed = two bits per base + 32-mer start mask (what the eref executable's "
                         "parser threads produce; palace_eref_count_reads_packed), ascii = a byte per base (palace_eref_count_reads)")
    ap.add_argument("--batches-in-flight", type=int, choices=(1, 2), default=1,
                    help="one GPU: 2 = the k-mer table is double-buffered and the counting kernels of a step run beside Phase B of the step "
                         "before (its rows are fetched one step later): 10.4 instead of 11.1 ms per step, but the count launch then shares the "
                         "device and its own duration -- the roofline figure -- grows from 9.2 to 10.3 ms; 1 (default) = every step drains "
                         "before the next, the count launch is timed with only this step's generateGraph stream beside it")
    ap.add_argument("--fused-probe", type=int, choices=(0, 1), default=0,
                    help="1: Phase B's channel-0 probe rides along in the count kernel (palace_eref_attach_probe_index): the step is ~0.1 ms shorter "
                         "(10.28 against 10.36 ms), the count launch 0.7 ms longer (9.19 against 8.46 ms) and Phase B 0.8 ms shorter; 0 (default): "
                         "the count launch is Phase A alone, which is what `roofline` is about")
    ap.add_argument("--graph-lag", type=int, choices=(0, 1), default=0,
                    help="1: the graph result of a step (stage 04's decomposition) is collected at the START of the next step -- a two-deep "
                         "pipeline of stream B, as a resident service would submit sample i+1 before it reads sample i's paths; with "
                         "--stage04-hold l2 the matching rounds then run beside the count kernel and Phase B only and the step is stream A's "
                         "length.  0 (default): every step collects its own result before it ends")
    ap.add_argument("--stage04-hold", choices=("0", "l1", "l2", "1"), default="0",
                    help="hold stage 04's matching rounds back until level 1 of the count launch (l1), its partition kernels (l2) or the whole launch (1) are done")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files -> files leg (CLI chain on generated files)")
    ap.add_argument("--soak-seconds", type=float, default=2.0,
                    help="after the K timed steps keep stepping (untimed for `value`) until this much wall time has passed")
    ap.add_argument("--cpu-sample-frac", type=float, default=0.10, help="share of reads / records the CPU baseline is timed on")
    a = ap.parse_args()
    if a.contigs is None:
        a.contigs = 100_000 if a.workload == "long" else 1_000_000
    return a


# ----------------------------------------------------------------------------------------------
# synthetic sample, generated on the device (SURVEY.md section 8(d) shapes)
# ----------------------------------------------------------------------------------------------
def contig_lengths(n_contigs, long_mode):
    """log-normal contig lengths: median 800 (sigma 1, min 56), or the long-contig set: median 30 kb, sigma 0.9
    (N50 ~ 50 kb, 7 % of the contigs above 110 kb, where exp(-d/150) underflows: generate_graph.cpp:255-260)."""
    rng = np.random.Generator(np.random.PCG64(SEED + 7))
    if long_mode:
        return np.maximum(56, rng.lognormal(np.log(30000.0), 0.9, size=n_contigs)).astype(np.int64)
    return np.maximum(56, rng.lognormal(np.log(800.0), 1.0, size=n_contigs)).astype(np.int64)


def make_sample(torch, dev, n_contigs, n_refs, rank=0, world=1, long_mode=False, read_weights=None):
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

    scale = 1.0 if long_mode else n_contigs / 1_000_000     # the long-contig set keeps the read volume of the 1M config
    # phage DB: n_refs refs, length U[20 kb, 60 kb]
    ref_lens = rng.integers(20000, 60001, size=n_refs).astype(np.int64)
    ref_off = np.zeros(n_refs + 1, dtype=np.int64)
    np.cumsum(ref_lens, out=ref_off[1:])
    ref_bases = dna(int(ref_off[-1]))
    # contigs: log-normal lengths (median 800, sigma 1, min 56); the read pool
    c_lens = contig_lengths(n_contigs, long_mode)
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
    if world > 1:                                  # reads shard by record range across ranks (read_weights: relative shares, e.g. none for rank 0)
        w = np.asarray(read_weights if read_weights is not None else [1.0] * world, dtype=np.float64)
        cuts = np.concatenate([[0], np.floor(np.cumsum(w) / w.sum() * n_pairs + 1e-9).astype(np.int64)])
        cuts[-1] = n_pairs
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        r1, r2 = r1[lo:hi].contiguous(), r2[lo:hi].contiguous()
    n_loc = r1.shape[0]
    read_off = torch.arange(2 * n_loc + 1, device=dev, dtype=torch.int64) * READ_LEN
    del pool
    return dict(n_contigs=n_contigs, n_refs=n_refs, ref_bases=ref_bases,
                ref_off=torch.from_numpy(ref_off).to(dev), ref_total=int(ref_off[-1]), ref_lens=ref_lens,
                r1=r1.reshape(-1), r2=r2.reshape(-1), r12=torch.cat([r1.reshape(-1), r2.reshape(-1)]), read_off=read_off,
                n_reads_side=n_loc,
                n_pairs_total=n_pairs, present=np.sort(present))


# ----------------------------------------------------------------------------------------------
# BAM-side sample: one primary record per read, coordinate sorted, as decoded columns in HBM
# ----------------------------------------------------------------------------------------------
def make_graph_sample(torch, dev, n_contigs, n_pairs, rank=0, world=1, long_mode=False):
    g = torch.Generator(device=dev)
    g.manual_seed(SEED + 1)
    rng = np.random.Generator(np.random.PCG64(SEED + 1))
    c_lens = contig_lengths(n_contigs, long_mode)
    ids = rng.permutation(np.arange(1, 4 * n_contigs + 1))[:n_contigs]
    covs = rng.gamma(2.0, 8.0, size=n_contigs)
    names = [f"EDGE_{i}_length_{l}_cov_{c:.6f}" for i, l, c in zip(ids.tolist(), c_lens.tolist(), covs.tolist())]
    order = np.argsort(np.array(names, dtype="S"))
    trank = np.empty(n_contigs, dtype=np.int32)
    trank[order] = np.arange(n_contigs, dtype=np.int32)
    link = rng.integers(0, n_contigs, size=n_contigs)
    link = np.where(link == np.arange(n_contigs), (link + 1) % n_contigs, link)
    # FASTG links: one random successor per contig plus, for half of them, the evidence-bearing one
    a = np.concatenate([np.arange(n_contigs), np.arange(n_contigs)[::2]])
    b = np.concatenate([rng.integers(0, n_contigs, size=n_contigs), link[::2]])
    o1 = rng.integers(0, 2, size=len(a)).astype(np.uint64)
    o2 = np.concatenate([rng.integers(0, 2, size=n_contigs), np.zeros(len(a) - n_contigs, dtype=np.int64)]).astype(np.uint64)
    o1[n_contigs:] = 0
    k1 = (a.astype(np.uint64) << np.uint64(33)) | (b.astype(np.uint64) << np.uint64(2)) | (o1 << np.uint64(1)) | o2
    k2 = (b.astype(np.uint64) << np.uint64(33)) | (a.astype(np.uint64) << np.uint64(2)) | ((o1 ^ np.uint64(1)) << np.uint64(1)) | (o2 ^ np.uint64(1))
    fastg = np.unique(np.concatenate([k1, k2]))

    T = lambda x, dt=None: torch.as_tensor(x, device=dev) if dt is None else torch.as_tensor(x, device=dev).to(dt)
    lens_t, link_t = T(c_lens), T(link)
    cum = torch.cumsum(lens_t, 0)
    start = cum - lens_t
    # read 1 of every pair
    u = (torch.rand(n_pairs, generator=g, device=dev, dtype=torch.float64) * float(cum[-1].item())).long()
    ta = torch.searchsorted(cum, u, right=True).clamp_(max=n_contigs - 1)
    la = lens_t[ta]
    p1 = torch.minimum(u - start[ta], torch.clamp(la - 2, min=0))
    ins = torch.clamp(torch.normal(400.0, 40.0, (n_pairs,), generator=g, device=dev), 150, 800).long()
    rev1 = torch.rand(n_pairs, generator=g, device=dev) < 0.5
    tb = ta.clone()
    p2 = torch.where(rev1, torch.clamp(p1 - ins + 150, min=0), torch.minimum(p1 + ins - 150, torch.clamp(la - 2, min=0)))
    rev2 = ~rev1
    kind = torch.rand(n_pairs, generator=g, device=dev)
    hot_x = T(rng.choice(n_contigs, size=max(8, n_contigs // 50), replace=False))      # junctions seen by pairs
    hot_s = T(rng.choice(n_contigs, size=max(8, n_contigs // 33), replace=False))      # junctions seen by split reads
    cross = kind < 0.04
    split = (kind >= 0.04) & (kind < 0.10)                                            # 6 % of pairs = 3 % of reads
    nx, ns = int(cross.sum().item()), int(split.sum().item())

    def end_pos(L, n):      # 0-based position whose 1-based value is in the END region
        lo = torch.maximum(L - 300, L // 2)
        return lo + (torch.rand(n, generator=g, device=dev) * torch.clamp(L - 1 - lo, min=1).float()).long()

    def start_pos(L, n):
        hi = torch.minimum(torch.full_like(L, 300), L // 2)
        return (torch.rand(n, generator=g, device=dev) * torch.clamp(hi, min=1).float()).long().clamp_(max=299)

    xa = hot_x[torch.randint(0, len(hot_x), (nx,), generator=g, device=dev)]
    ta[cross] = xa; tb[cross] = link_t[xa]
    p1[cross] = end_pos(lens_t[xa], nx); p2[cross] = start_pos(lens_t[link_t[xa]], nx)
    rev1[cross] = False; rev2[cross] = True
    if long_mode:
        # half of the cross pairs as (a-, b+): read 1 reverse at a's START, mate reverse at b's START.  The '-' side measures
        # its distance to the far end of a (nearEndDistances, generate_graph.cpp:310-318), so on contigs above ~110 kb the
        # score underflows to 0 and the evidence is rejected -- the G5 gate this configuration is about.
        flip = cross & (torch.rand(n_pairs, generator=g, device=dev) < 0.5)
        p1[flip] = start_pos(lens_t[ta[flip]], int(flip.sum().item()))
        rev1[flip] = True
    sa_a = hot_s[torch.randint(0, len(hot_s), (ns,), generator=g, device=dev)]
    ta[split] = sa_a; tb[split] = sa_a
    p1[split] = end_pos(lens_t[sa_a], ns); p2[split] = torch.clamp(p1[split] - 250, min=0)
    rev1[split] = False; rev2[split] = True

    def mapq_nm(n):
        r = torch.rand(n, generator=g, device=dev)
        mq = torch.where(r < 0.7, 60, torch.where(r < 0.85, 40, torch.where(r < 0.95, 20, 0))).to(torch.uint8)
        nm = (torch.rand(n, generator=g, device=dev) ** 2 * 7).to(torch.int32)
        return mq, nm

    mq1, nm1 = mapq_nm(n_pairs)
    mq2, nm2 = mapq_nm(n_pairs)
    i32 = torch.int32
    f1 = (0x41 + 0x10 * rev1.long() + 0x20 * rev2.long()).to(torch.int16)
    f2 = (0x81 + 0x10 * rev2.long() + 0x20 * rev1.long()).to(torch.int16)
    pair_id = torch.arange(n_pairs, device=dev, dtype=torch.int64)
    qk = (pair_id * -7046029254386353131) ^ (pair_id >> 7)                            # distinct per pair
    rl1 = torch.where(split, 90, 150).to(i32)
    ce1 = torch.where(split, 60, 0).to(i32)
    col = dict(
        tid=torch.cat([ta, tb]).to(i32), pos=torch.cat([p1, p2]).to(i32), mtid=torch.cat([tb, ta]).to(i32),
        mpos=torch.cat([p2, p1]).to(i32), flag=torch.cat([f1, f2]), mapq=torch.cat([mq1, mq2]), nm=torch.cat([nm1, nm2]),
        ref_len=torch.cat([rl1, torch.full((n_pairs,), 150, device=dev, dtype=i32)]),
        read_len=torch.full((2 * n_pairs,), 150, device=dev, dtype=i32),
        clip_s=torch.zeros(2 * n_pairs, device=dev, dtype=i32), clip_e=torch.cat([ce1, torch.zeros(n_pairs, device=dev, dtype=i32)]),
        qkey=torch.cat([qk, qk]), has_sa=torch.cat([split, torch.zeros(n_pairs, device=dev, dtype=torch.bool)]))
    sa_tid = torch.cat([link_t[ta], torch.zeros(n_pairs, device=dev, dtype=torch.int64)])
    sa_pos = torch.cat([start_pos(lens_t[link_t[ta]], n_pairs) + 1, torch.zeros(n_pairs, device=dev, dtype=torch.int64)])
    sa_mq, sa_nm = mapq_nm(2 * n_pairs)
    key = col["tid"].long() * (1 << 32) + col["pos"].long()
    perm = torch.argsort(key, stable=True)
    col = {k: v[perm].contiguous() for k, v in col.items()}
    sa_tid, sa_pos, sa_mq, sa_nm = sa_tid[perm], sa_pos[perm], sa_mq[perm], sa_nm[perm]
    n_rec = 2 * n_pairs
    if world > 1:                                   # records shard by ordinal range across ranks
        lo, hi = n_rec * rank // world, n_rec * (rank + 1) // world
        col = {k: v[lo:hi].contiguous() for k, v in col.items()}
        sa_tid, sa_pos, sa_mq, sa_nm = sa_tid[lo:hi], sa_pos[lo:hi], sa_mq[lo:hi], sa_nm[lo:hi]
    else:
        lo, hi = 0, n_rec
    hs = col.pop("has_sa")
    sa_off = torch.zeros(hi - lo + 1, device=dev, dtype=i32)
    sa_off[1:] = torch.cumsum(hs.to(i32), 0)
    n_sa = int(sa_off[-1].item())
    sa = torch.zeros((max(1, n_sa), 8), device=dev, dtype=i32)             # palace_sa_item rows
    sa[:n_sa, 0] = sa_tid[hs].to(i32); sa[:n_sa, 1] = sa_pos[hs].to(i32); sa[:n_sa, 2] = sa_mq[hs].to(i32)
    sa[:n_sa, 3] = sa_nm[hs]; sa[:n_sa, 4] = 90; sa[:n_sa, 5] = 0; sa[:n_sa, 6] = 150; sa[:n_sa, 7] = 0
    total_ref = float(col["ref_len"].sum().item()) if world == 1 else None
    return dict(col=col, sa_off=sa_off, sa=sa, n_sa=n_sa, n=hi - lo, ord_base=lo, n_total=n_rec, fastg_links=(a, b, o1, o2),
                tlen=T(c_lens, i32), trank=T(trank), fastg=T(fastg.view(np.int64)), n_fastg=len(fastg),
                names=names, lens=c_lens, link=link, avg_depth=None if total_ref is None else float(f"{total_ref / c_lens.sum():.6g}"))


def make_side_inputs(gs):
    """The per-sample side inputs of filter_graph.py: hit_seqs 3 % of the contigs, node_scores all of them
    (uniform, some in e-05 notation), .blast for 2 %, one contigs.paths entry per 3 contigs -- once, as data: the files -> files
    leg writes them out as text, the resident step gets them as the arrays of palace_stage04_inputs."""
    rng = np.random.Generator(np.random.PCG64(SEED + 2))
    names, lens = gs["names"], gs["lens"]
    n = len(names)
    hit = rng.choice(n, size=max(1, n * 3 // 100), replace=False)
    hit_k = rng.integers(1, 9, size=len(hit))
    sc = rng.random(n)
    tiny = rng.random(n) < 0.05
    score_text = [(f"{x * 9:.4f}e-05" if t else f"{x:.6f}") for x, t in zip(sc.tolist(), tiny.tolist())]
    bl = rng.choice(n, size=max(1, n // 50), replace=False)
    bl_ident = rng.choice([99.5, 85.0, 69.9], size=len(bl))
    bl_frac = rng.choice([0.3, 0.8, 0.95], size=len(bl))
    bl_ref = rng.integers(1, 200, size=len(bl))
    k_paths = max(1, n // 3)
    mem = rng.integers(0, n, size=(k_paths, 3))
    sg = rng.integers(0, 2, size=(k_paths, 3))
    # seed bits as filter_graph.py derives them from those files (:66-112), thresholds 0.7 / 0.7 as palace:568-579 passes them
    seed = np.zeros(n, np.uint8)
    al = np.maximum(30, (lens[bl] * bl_frac).astype(np.int64))
    seed[bl[(bl_ident > 70.0) & ((al / lens[bl] > 0.7) | (al > 2000))]] |= 1
    seed[hit] |= 2
    score_hit = np.fromiter((0.0 if t else float(f"{float(s):.3f}") for s, t in zip(score_text, tiny.tolist())), dtype=np.float64, count=n) > 0.7
    seed[score_hit] |= 4
    # contigs.paths: every entry is two path lines (the path and its reverse complement)
    fwd = 2 * mem + sg
    rc = (2 * mem + (1 - sg))[:, ::-1]
    tok = np.stack([fwd, rc], axis=1).reshape(-1).astype(np.int32)
    off = np.arange(2 * k_paths + 1, dtype=np.int64) * 3
    return dict(hit=hit, hit_k=hit_k, score_text=score_text, bl=bl, bl_ident=bl_ident, bl_frac=bl_frac, bl_ref=bl_ref, mem=mem, sg=sg,
                seed=seed, path_off=off, path_tok=tok)


def graph_to_arcs(cn, n_segs, edges, min_count=5):
    """host glue between generateGraph's numbers and matching's input (JUNC filter :1056-1061, arc + conjugate,
    arc ranking): the library's own host routine, the same one palace_amd/host/matching_main.cpp ranks with."""
    from palace_amd import capi
    assert len(cn) == n_segs
    return capi.match_arcs_from_edges(cn, edges, min_count, reuse=True)


# ----------------------------------------------------------------------------------------------
# files -> files: the same sample as the FILES the pipeline hands to the three executables, and the CLI chain on them
# ----------------------------------------------------------------------------------------------
def fastq_to_file(torch, reads, n, tag, path):
    """4-line FASTQ, fixed-width names @r0000000/<tag> (extract_ref.cpp:940-1004 reads line 1 of every 4), built on the
    device as an [n, record] byte matrix."""
    dev = reads.device
    digits = 8
    w = 2 + digits + 3 + READ_LEN + 3 + READ_LEN + 1                    # "@r" d "/t\n" seq "\n+\n" qual "\n"
    step = 1 << 20
    with open(path, "wb") as f:
        for lo in range(0, n, step):
            m = min(step, n - lo)
            rec = torch.empty((m, w), dtype=torch.uint8, device=dev)
            idx = torch.arange(lo, lo + m, device=dev)
            rec[:, 0] = 64; rec[:, 1] = 114
            for k in range(digits):
                rec[:, 2 + k] = ((idx // 10 ** (digits - 1 - k)) % 10 + 48).to(torch.uint8)
            o = 2 + digits
            rec[:, o] = 47; rec[:, o + 1] = ord(tag); rec[:, o + 2] = 10
            o += 3
            rec[:, o:o + READ_LEN] = reads[lo * READ_LEN:(lo + m) * READ_LEN].view(m, READ_LEN)
            o += READ_LEN
            rec[:, o] = 10; rec[:, o + 1] = 43; rec[:, o + 2] = 10
            rec[:, o + 3:o + 3 + READ_LEN] = 73
            rec[:, o + 3 + READ_LEN] = 10
            f.write(rec.cpu().numpy().tobytes())


def write_e2e_inputs(torch, sample, gs, hdr, work):
    """Every file of palace:473-480 and 555-600 for this sample.  Generation is not timed."""
    t0 = time.perf_counter()
    P = {k: os.path.join(work, v) for k, v in dict(
        fq1="reads_1.fq", fq2="reads_2.fq", fa="phagedb.fa", hdr="coder.hdr", bam="reads_pe_primary.sort.bam", cols="bam_cols",
        fastg_fai="assembly_graph.fastg.fai", fasta_fai="assembly_graph.fasta.fai", blast="assembly_graph.fasta.blast",
        hit="hit_seqs.out", score="node_scores.out", paths="contigs.paths", graph="s_graph.txt", pre="s_filtered_graph_pre.txt",
        filt="s_filtered_graph.txt", allhit="all_hit_segs.txt", lin="s_linear.txt", cyc="s_cycle.txt", nodup="s_cycle_nodup.txt",
        result="s_all_result.txt", refnames="s_ref_names.txt", tmp="s_tmp.txt").items()}
    n_side = sample["n_reads_side"]
    fastq_to_file(torch, sample["r1"], n_side, "1", P["fq1"])
    fastq_to_file(torch, sample["r2"], n_side, "2", P["fq2"])
    rb, ro = sample["ref_bases"].cpu().numpy(), sample["ref_off"].cpu().numpy()
    with open(P["fa"], "wb") as f:
        for i in range(sample["n_refs"]):
            b = rb[ro[i]:ro[i + 1]].tobytes()
            f.write(b">phage_%d synthetic\n" % (i + 1) + b"\n".join(b[k:k + 80] for k in range(0, len(b), 80)) + b"\n")
    open(P["hdr"], "wb").write(np.asarray(hdr, dtype=np.uint8).tobytes())
    # BAM: the decoded columns go through palace_amd/bin/synthbam (multi-threaded BGZF writer)
    os.makedirs(P["cols"], exist_ok=True)
    names, lens = gs["names"], gs["lens"]
    c = gs["col"]
    for k in ("tid", "pos", "mtid", "mpos", "nm", "ref_len", "clip_e"):
        c[k].cpu().numpy().astype(np.int32).tofile(os.path.join(P["cols"], k + ".i32"))
    gs["sa_off"].cpu().numpy().astype(np.int32).tofile(os.path.join(P["cols"], "sa_off.i32"))
    gs["sa"][: max(1, gs["n_sa"])].cpu().numpy().astype(np.int32).tofile(os.path.join(P["cols"], "sa.i32"))
    c["flag"].cpu().numpy().view(np.uint16).tofile(os.path.join(P["cols"], "flag.u16"))
    c["mapq"].cpu().numpy().tofile(os.path.join(P["cols"], "mapq.u8"))
    c["qkey"].cpu().numpy().view(np.uint64).tofile(os.path.join(P["cols"], "qkey.u64"))
    with open(os.path.join(P["cols"], "targets.tsv"), "w") as f:
        f.write("".join(f"{n}\t{l}\n" for n, l in zip(names, lens.tolist())))
    import subprocess
    subprocess.run([os.path.join(ROOT, "palace_amd", "bin", "synthbam"), P["cols"], P["bam"], str(min(16, os.cpu_count() or 1)), "1"], check=True)
    # FASTG .fai (generate_graph.cpp:119-169 reads column 0 only): one line per link
    a, b, o1, o2 = gs["fastg_links"]
    q = "'"
    with open(P["fastg_fai"], "w") as f:
        f.write("".join(f"{names[x]}{q if u else ''}:{names[y]}{q if (u ^ v) else ''};\t{lens[x]}\t0\t60\t61\n"
                        for x, y, u, v in zip(a.tolist(), b.tolist(), o1.tolist(), o2.tolist())))
    # side inputs of filter_graph.py: the same data the resident step's palace_stage04 object was built from (make_side_inputs)
    sd = gs["side"]
    n = len(names)
    with open(P["fasta_fai"], "w") as f:
        f.write("".join(f"{nm}\t{l}\t{7 + 100 * i}\t60\t61\n" for i, (nm, l) in enumerate(zip(names, lens.tolist()))))
    with open(P["hit"], "w") as f:
        f.write("".join(f"{names[i]}\t{k}\n" for i, k in zip(sd["hit"].tolist(), sd["hit_k"].tolist())))
    with open(P["score"], "w") as f:
        f.write("".join(f"{nm}\t{t}\n" for nm, t in zip(names, sd["score_text"])))
    with open(P["blast"], "w") as f:
        for i, ident, frac, ref in zip(sd["bl"].tolist(), sd["bl_ident"].tolist(), sd["bl_frac"].tolist(), sd["bl_ref"].tolist()):
            L = int(lens[i]); al = max(30, int(L * frac))
            f.write(f"{names[i]}\tphage_{ref}\t{ident:.3f}\t{al}\t3\t0\t1\t{al}\t100\t{100 + al}\t1e-50\t200\t{L}\t40000\n")
    with open(P["paths"], "w") as f:
        f.write(paths_text(names, lens, sd))
    P["gen_s"] = time.perf_counter() - t0
    P["bytes"] = {k: os.path.getsize(P[k]) for k in ("fq1", "fq2", "fa", "bam", "fastg_fai")}
    return P


def paths_text(names, lens, sd):
    """contigs.paths (SPAdes): NODE header, the path, NODE' header, its reverse complement"""
    ids = [nm.split("_")[1] for nm in names]
    mem, sg = sd["mem"], sd["sg"]
    out = []
    for k in range(len(mem)):
        fwd = [ids[j] + "+-"[t] for j, t in zip(mem[k].tolist(), sg[k].tolist())]
        rc = [t[:-1] + ("-" if t[-1] == "+" else "+") for t in reversed(fwd)]
        tot = int(lens[mem[k]].sum())
        out.append(f"NODE_{k + 1}_length_{tot}_cov_9.5\n{','.join(fwd)}\nNODE_{k + 1}_length_{tot}_cov_9.5'\n{','.join(rc)}\n")
    return "".join(out)


def run_e2e(P, avg_depth, n_contigs, rows_host, n_junc_expected, result_text_expected=None):
    """The chain of palace:473-480 and 555-600 on the files, one process per stage as the driver runs them.  Returns wall
    seconds per stage.  eref is run twice: the first run builds <db>.k32.index.dat (once per DB, extract_ref.cpp:1245-1251),
    the second finds it -- the steady state of a DB shared by many samples and the one that enters `seconds`."""
    import subprocess
    B = os.path.join(ROOT, "palace_amd", "bin")
    S = os.path.join(ROOT, "palace_amd", "scripts")
    threads = str(min(16, os.cpu_count() or 1))
    st = {}

    def timed(key, cmd, stdout=None, env=None):
        t0 = time.perf_counter()
        subprocess.run(cmd, check=True, stdout=stdout, env=env)
        st[key] = time.perf_counter() - t0

    eref = [os.path.join(B, "eref"), P["fq1"], P["fq2"], P["fa"], P["tmp"], "0.9", "0.85", threads]
    with open(P["refnames"], "wb") as f:
        timed("eref_first_run_builds_index", eref, stdout=f, env=dict(os.environ, PALACE_CODER_HEADER=P["hdr"]))
    # (the run above is set-up: it builds the DB's 2.4 GB index file once, as the reference's first run on a DB does.  Its worker process
    # is torn down behind the back of the process we waited for -- host/fast_exit.hpp --, and a GPU process started while that goes on
    # waits 0.1-0.3 s longer for its HIP runtime: let the set-up finish before the timed stages start)
    time.sleep(1.0)
    with open(P["refnames"], "wb") as f:
        timed("eref", eref, stdout=f)
    timed("generateGraph", [os.path.join(B, "generateGraph"), P["bam"], P["fastg_fai"], P["graph"], f"{avg_depth:.6g}"])
    timed("filter_graph.py", [sys.executable, os.path.join(S, "filter_graph.py"), P["fastg_fai"], P["graph"], P["pre"], f"{avg_depth:.6g}", "0",
                              P["hit"], P["score"], P["blast"], "0.7", P["fasta_fai"], P["allhit"], P["paths"], "0.7"])
    with open(P["filt"], "wb") as f:
        timed("uniq", ["uniq", P["pre"]], stdout=f)
    timed("matching", [os.path.join(B, "matching"), "-g", P["filt"], "-r", P["lin"], "-c", P["cyc"], "-s", "-i", "10", "-l", P["paths"]])
    timed("remove_cycle_dup.py", [sys.executable, os.path.join(S, "remove_cycle_dup.py"), P["cyc"], P["nodup"]])
    t0 = time.perf_counter()
    with open(P["result"], "wb") as f:
        for k in ("lin", "nodup"):
            f.write(open(P[k], "rb").read())
    st["cat"] = time.perf_counter() - t0
    # the same files from ONE process: generateGraph with its stage-04 options (palace_amd/host/stage04_fused.hpp) -- the graph stays
    # in HBM between the stages, every named artefact is still written; the separate executables above stay for the unchanged driver
    fused = None
    try:
        fp = {k: P[k] + ".fused" for k in ("graph", "pre", "filt", "allhit", "lin", "cyc", "nodup", "result")}
        t0 = time.perf_counter()
        subprocess.run([os.path.join(B, "generateGraph"), "--hit-seqs", P["hit"], "--node-scores", P["score"], "--blast", P["blast"], "--fasta-fai", P["fasta_fai"],
                        "--paths", P["paths"], "--filtered-pre", fp["pre"], "--filtered", fp["filt"], "--all-hit-segs", fp["allhit"], "--linear", fp["lin"],
                        "--cycle", fp["cyc"], "--cycle-nodup", fp["nodup"], "--all-result", fp["result"], "-s", "-i", "10",
                        P["bam"], P["fastg_fai"], fp["graph"], f"{avg_depth:.6g}"], check=True)
        t_fused = time.perf_counter() - t0
        same = all(open(fp[k], "rb").read() == open(P[k], "rb").read() for k in ("graph", "pre", "filt", "allhit", "lin", "cyc", "nodup", "result"))
        fused = dict(seconds=st["eref"] + t_fused, contigs_per_s=n_contigs / (st["eref"] + t_fused),
                     stage_s=dict(eref=round(st["eref"], 3), generateGraph_with_stage04=round(t_fused, 3)),
                     files_identical_to_the_chain=bool(same),
                     note="eref + ONE generateGraph process that also writes _filtered_graph_pre / _filtered_graph / all_hit_segs / linear / cycle / "
                          "cycle_nodup / all_result (its --filtered-pre ... --all-result options)")
    except Exception as e:
        fused = dict(error=f"{type(e).__name__}: {str(e)[:200]}")
    # cross-check against the HBM-resident step (same coder header, same sample): reported refs and kept junctions
    r = rows_host
    want = {(i + 1, int(r[i, 0]), int(r[i, 1])) for i in range(len(r))
            if r[i, 1] > 0 and np.float32(r[i, 1]) / np.float32(r[i, 2]) > np.float32(0.75)}
    got = {tuple(int(x) for x in l.split("\t")[1:4]) for l in open(P["refnames"]).read().splitlines()}
    n_junc = sum(1 for l in open(P["graph"]) if l.startswith("JUNC"))
    total = sum(v for k, v in st.items() if k != "eref_first_run_builds_index")
    same_result = None if result_text_expected is None else bool(open(P["result"]).read() == result_text_expected)
    return dict(seconds=total, contigs_per_s=n_contigs / total, stage_s={k: round(v, 3) for k, v in st.items()},
                agrees_with_resident_step=bool(got == want and n_junc == n_junc_expected and same_result is not False),
                all_result_identical_to_resident_step=same_result, one_process_stage04=fused,
                refs_reported=len(got), junc_lines=n_junc, result_lines=sum(1 for _ in open(P["result"])),
                input_bytes=P["bytes"], input_generation_s=round(P["gen_s"], 1),
                note="wall clock of eref + generateGraph + filter_graph.py + uniq + matching + remove_cycle_dup.py + cat, one process "
                     "per stage, files in the page cache; eref with the index file of the DB present (built by the first run)")


# ----------------------------------------------------------------------------------------------
def reference_eref_check(b1, b2, off, rb, ro, n_ref_s, tmp):
    """When the compiled reference travels with the repo (oracle/_ref/eref_ref, built from the unmodified
    extract_ref.cpp), time IT on the same read sample at two sizes: marginal reads/s next to the port's."""
    import subprocess
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "eref_ref")
    if not os.path.exists(ref_bin):
        return None
    try:
        fa = os.path.join(tmp, "db.fa")
        with open(fa, "wb") as f:
            for i in range(n_ref_s):
                f.write(b">ref%d\n" % i + rb[ro[i]:ro[i + 1]].tobytes() + b"\n")
        n = len(off) - 1
        times = {}
        for frac in (2, 1):                          # half the sample, then all of it (first run also builds the index)
            m = n // frac
            for tag, b in (("1", b1), ("2", b2)):
                with open(os.path.join(tmp, f"s_{tag}.fq"), "wb") as f:
                    f.write(b"".join(b"@r%d\n" % i + b[off[i]:off[i + 1]].tobytes() + b"\n+\n" + b"I" * READ_LEN + b"\n"
                                     for i in range(m)))
            if frac == 2:                            # untimed run that leaves the index beside the DB
                subprocess.run([ref_bin, os.path.join(tmp, "s_1.fq"), os.path.join(tmp, "s_2.fq"), fa, os.path.join(tmp, "t.txt"),
                                "0.9", "0.85", "1"], stdout=subprocess.DEVNULL, check=True, timeout=300)
            best = None
            for _ in range(2):                       # best of two: the fixed part (4 GiB table, 16 GiB dead arrays) is noisy
                t0 = time.perf_counter()
                subprocess.run([ref_bin, os.path.join(tmp, "s_1.fq"), os.path.join(tmp, "s_2.fq"), fa, os.path.join(tmp, "t.txt"),
                                "0.9", "0.85", "1"], stdout=subprocess.DEVNULL, check=True, timeout=300)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            times[2 * m] = best
        (ra, ta), (rbn, tb) = sorted(times.items())
        marginal = (rbn - ra) / max(1e-9, tb - ta)
        return dict(binary="oracle/_ref/eref_ref (unmodified extract_ref.cpp, -O2, threads=1, cached index)",
                    runs_s={str(k): round(v, 2) for k, v in times.items()}, marginal_reads_per_s=marginal,
                    fixed_s=ta - ra / marginal)
    except Exception as e:                           # never let the cross-check break the bench line
        return dict(error=str(e)[:200])


def bam_decode_seconds(bam_path, cores):
    """BGZF inflate + BAM record decode of the WHOLE BAM of the workload on the host, through this repo's loader (hostdump
    bamtime: the same code path generateGraph loads with, no GPU): with zlib's inflate() on one thread -- what htslib's
    sam_read1 does for the reference's single-threaded loop (generate_graph.cpp:611-669) --, with zlib on `cores` threads, and
    as shipped (the loader's own DEFLATE decoder on `cores` threads).  Seconds each; the file is in the page cache."""
    import subprocess
    exe = os.path.join(ROOT, "palace_amd", "bin", "hostdump")
    if not (bam_path and os.path.exists(bam_path) and os.path.exists(exe)):
        return None
    def run(threads, zlib):
        env = dict(os.environ)
        env.pop("PALACE_BAM_ZLIB", None)
        if zlib:
            env["PALACE_BAM_ZLIB"] = "1"
        t0 = time.perf_counter()
        subprocess.run([exe, "bamtime", bam_path, str(threads)], check=True, stdout=subprocess.DEVNULL, env=env, timeout=600)
        return time.perf_counter() - t0
    try:
        return dict(zlib_1_thread=run(1, True), zlib_threads=run(cores, True), own_decoder_threads=run(cores, False), threads=cores,
                    bam_bytes=os.path.getsize(bam_path))
    except Exception as e:
        return dict(error=f"{type(e).__name__}: {str(e)[:200]}")


def cpu_baseline(torch, sample, gs, header, frac, graph_out, bam_path=None):
    """The oracle (CPU restatement of the reference algorithm) on a bounded sample of every stage, extrapolated linearly to
    the whole workload: at threads = 1 (the only configuration in which the reference's semantics are defined, SURVEY.md
    F5) -> `value`; with the read counting on min(nproc, 16) threads (race-free: saturating increment by compare-and-swap)
    -> `multi_thread`; and with the reference's dead 16.3 GiB allocation + memset (extract_ref.cpp:1296-1299) added at
    threads = 1 -> `as_shipped`.  The oracle keeps the reference's 32-step inner loop per (position, channel)."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as orc
    from palace_amd.synth import BamRecord
    cores = max(1, min(os.cpu_count() or 1, 16))
    cc = orc.header_to_cc(header)
    # ---- eref ----
    n_side = max(1000, min(sample["n_reads_side"], int(frac * sample["n_reads_side"])))
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
    table.clear()
    t0 = time.perf_counter()
    table.count_mt(b1, off, cc, cores)
    table.count_mt(b2, off, cc, cores)
    t_reads_mt = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:     # refs are independent (the reference splits them over T threads,
        list(ex.map(lambda i: orc.scan_ref(idx[i], int(ro[i + 1] - ro[i]), table, 0.9, 0.85), range(n_ref_s)))   # :1314-1329)
    t_refs_mt = time.perf_counter() - t0
    table.free()
    t0 = time.perf_counter()
    dead = orc.lib().orc_eref_reference_dead_cost()        # 16 GiB + 300 MB allocated and zeroed, never read
    t_dead = time.perf_counter() - t0 if dead else None
    total_reads = 2 * sample["n_pairs_total"]
    up_reads, up_refs = total_reads / (2 * n_side), sample["n_refs"] / n_ref_s
    t_eref = t_clear + t_reads * up_reads + t_refs * up_refs
    t_eref_mt = t_clear + t_reads_mt * up_reads + t_refs_mt * up_refs
    # ---- generateGraph: first m records of the sorted stream, rebuilt as BAM-level records (single thread, as the reference) ----
    m = max(1000, min(gs["n"], int(frac * gs["n"])))
    c = {k: v[:m].cpu().numpy() for k, v in gs["col"].items()}
    so = gs["sa_off"][: m + 1].cpu().numpy()
    sa = gs["sa"][: max(1, int(so[-1]))].cpu().numpy()
    names = gs["names"]
    recs = []
    for i in range(m):
        s_txt = None
        if so[i + 1] > so[i]:
            it = sa[so[i]]
            s_txt = f"{names[it[0]]},{it[1]},{'-' if it[7] else '+'},{it[4]}S{it[6] - it[4]}M,{it[2]},{it[3]};"
        cig = f"{c['ref_len'][i]}M{c['clip_e'][i]}S" if c["clip_e"][i] else "150M"
        recs.append(BamRecord(f"q{c['qkey'][i] & 0xffffffffffff:x}", int(c["flag"][i]) & 0xffff, int(c["tid"][i]), int(c["pos"][i]),
                              int(c["mapq"][i]), cig, int(c["mtid"][i]), int(c["mpos"][i]), nm=int(c["nm"][i]), sa=s_txt))
    tmp = tempfile.mkdtemp(prefix="palace_bench_")
    hot = sorted(set(c["tid"].tolist()) | set(gs["link"][c["tid"]].tolist()))
    with open(os.path.join(tmp, "g.fastg.fai"), "w") as f:       # reduced .fai: only contigs the sample can touch
        for a in hot:
            f.write(f"{names[a]}:{names[gs['link'][a]]};\t{gs['lens'][a]}\t0\t60\t61\n")
    targets = list(zip(names, gs["lens"].tolist()))
    gin = orc.GraphInput(recs, targets)             # marshalling is not timed
    t0 = time.perf_counter()
    gin.run(os.path.join(tmp, "g.fastg.fai"), gs["avg_depth"])
    t_graph_s = time.perf_counter() - t0
    t_graph = t_graph_s * gs["n_total"] / m
    # ... and what the reference's loop spends inside sam_read1: BGZF inflate + record decode of the whole BAM (not a sample)
    dec = bam_decode_seconds(bam_path, cores)
    t_decode = dec["zlib_1_thread"] if dec and "error" not in dec else None
    t_decode_mt = dec["zlib_threads"] if t_decode is not None else None
    # ---- matching: the whole FILTERED graph this run produced (what palace:587-590 hands to `matching`), through the oracle's
    # own text parser, with contigs.paths ----
    gpath, ppath = os.path.join(tmp, "graph.txt"), os.path.join(tmp, "contigs.paths")
    e = graph_out["edges"][(graph_out["edge_flags"] & 6) != 0]
    with open(gpath, "w") as f:
        f.write("".join(f"SEG {names[c]} 1 {graph_out['cn'][c]} 0 0.000 0\n" for c in graph_out["contig_of"].tolist()))
        f.write("".join(f"JUNC {names[l]} {'+-'[a]} {names[r_]} {'+-'[b]} {x} 0\n"
                        for l, r_, a, b, x in zip(e["left"].tolist(), e["right"].tolist(), e["oL"].tolist(), e["oR"].tolist(),
                                                  e["counts"].astype(np.int64).sum(axis=1).tolist())))
    open(ppath, "w").write(paths_text(names, gs["lens"], gs["side"]))
    cap = 128 * len(names) + (1 << 20)
    t0 = time.perf_counter()
    orc.match_run(gpath, ppath, 10, cap=cap)
    t_match = time.perf_counter() - t0
    t_full, t_full_mt = t_eref + t_graph + t_match + (t_decode or 0.0), t_eref_mt + t_graph + t_match + (t_decode_mt or 0.0)
    # the COMPILED reference, when it travels with the repo, on a small part of the same reads: a cross-check of the port's rate
    k = min(n_side, 20000)
    ref_check = reference_eref_check(b1[: k * READ_LEN], b2[: k * READ_LEN], off[: k + 1], rb, ro, n_ref_s, tmp)
    if ref_check is not None and "error" not in ref_check and sample["n_contigs"] == 1_000_000:
        ref_check["full_size_note"] = ("measured once on a GPU box, not in this run: the compiled reference on the full eref input of "
                                       "this workload (5000 refs, 6.67 M reads) took 595.6 s at threads=1 incl. its index build, stdout "
                                       "byte-identical to ours (profiles/ref_compare_eref_full.log)")
    nc = sample["n_contigs"]
    out = dict(value=nc / t_full, unit="contigs/s", cores=1, kind="port",
               sample=(f"oracle/ at threads=1. eref: {2 * n_side} of {total_reads} reads x{READ_LEN} bp ({t_reads:.1f} s) + 4 GiB table "
                       f"memset ({t_clear:.1f} s, fixed) + scan of {n_ref_s} of {sample['n_refs']} refs ({t_refs:.2f} s) -> {t_eref:.0f} s "
                       f"extrapolated; generateGraph: first {m} of {gs['n_total']} decoded records ({t_graph_s:.1f} s; full .fai parse excluded) -> {t_graph:.0f} s, "
                       + (f"plus BGZF inflate + BAM decode of the whole {dec['bam_bytes'] / 1e6:.0f} MB BAM with zlib on one thread, as htslib's sam_read1 "
                          f"does ({t_decode:.1f} s, measured, not extrapolated)" if t_decode is not None else "BGZF/BAM decode NOT included (no BAM file in this run: --no-e2e)")
                       + f"; matching: the whole filtered graph with contigs.paths ({t_match:.1f} s; own "
                       f"algorithm, reference absent); filter_graph.py itself (Python glue) is not in the sum."),
               stage_s=dict(eref=t_eref, generateGraph=t_graph, generateGraph_bam_decode=t_decode, matching=t_match), port_reads_per_s=2 * n_side / t_reads,
               extrapolated="eref and generateGraph's record loop are timed on the sample named in `sample` and scaled linearly; table memset, BAM decode and matching are whole",
               bam_decode_s=dec,
               multi_thread=dict(value=nc / t_full_mt, unit="contigs/s", cores=cores, kind="port",
                                 note=f"read counting and ref scan on {cores} threads (same sample: {t_reads_mt:.1f} s and {t_refs_mt:.2f} s); "
                                      "generateGraph's record loop and matching single-threaded, as the reference's are; BGZF inflate on the same threads (htslib can: bgzf_mt)",
                                 stage_s=dict(eref=t_eref_mt, generateGraph=t_graph, generateGraph_bam_decode=t_decode_mt, matching=t_match)),
               reference_eref=ref_check)
    if t_dead is not None:
        out["as_shipped"] = dict(value=nc / (t_full + t_dead), unit="contigs/s", cores=1, kind="port",
                                 note=f"threads=1 plus the reference's never-read Peaks arrays: 16 GiB + 300 MB allocated and zeroed "
                                      f"({t_dead:.1f} s on this host, fixed per run; extract_ref.cpp:1296-1299)")
    return out


def roofline_stages(stages, traffic):
    """{stage: (algorithmic bytes per step, live ms per step, what the bytes are)} -> the per-stage roofline objects"""
    out = {}
    for k, (alg, ms, what) in stages.items():
        ach = alg / (ms * 1e-3) / 1e9 if ms and ms > 0 else None
        out[k] = {"bound": "hbm", "algorithmic_bytes_per_step": int(alg), "ms_per_step": float(ms), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": None if ach is None else ach / HBM_PEAK_GBS, "traffic": traffic.get(k), "bytes": what}
    return out


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (torch.distributed.run, one rank
    per GPU, rendezvous on 127.0.0.1) before this process has touched a GPU, let them print the JSON line, return their
    exit status.  (Never exec from a process that initialised the GPU.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def profiled_traffic(args, world, version, fused):
    """HBM bytes per step from the committed PMC profile (profiles/phase_a_traffic.json, written by tools/prof_full.sh +
    tools/traffic_json.py): FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, per count launch and per stage.  Only quoted for
    the workload AND the library build (palace_version(): a digest of the kernel sources) it was measured on; otherwise null.
    -> (bytes per count launch, source, {stage: bytes})"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "phase_a_traffic.json")))
    except Exception:
        return None, None, {}
    if world != 1 or t.get("contigs") != args.contigs or t.get("workload", "default") != args.workload or t.get("reads", "ascii") != args.reads:
        return None, "profiles/phase_a_traffic.json is of another workload", {}
    if bool(t.get("fused_probe", False)) != bool(fused):
        return None, "profiles/phase_a_traffic.json was measured with" + ("out" if fused else "") + " the fused probe", {}
    if t.get("build") != version:
        return None, f"profiles/phase_a_traffic.json was measured on another build ({t.get('build')}); this is {version}", {}
    return t.get("bytes_per_launch"), t.get("source"), {k: v.get("bytes") for k, v in (t.get("stages") or {}).items()}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to fd 1 when a process group
    # initialises, so everything written to fd 1 before the final line is routed to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    from types import SimpleNamespace
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PALACE_BENCH_ONE_DEVICE") == "1":   # rehearsal only: every rank on GPU 0 (with PALACE_BENCH_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_exchange = os.environ.get("PALACE_FORCE_EXCHANGE") == "1"      # rehearse the N>1 code path on one GPU (reads sharded, table exchange)
    force_key_split = os.environ.get("PALACE_FORCE_KEY_SPLIT") == "1"    # ... and the key-space split with its gather (one rank: its share is everything)
    if world > 1 or force_exchange or force_key_split:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("PALACE_BENCH_BACKEND", "nccl")           # "nccl" is RCCL; gloo only for rehearsals
        if backend == "nccl":
            # the collectives of a step are small and sit between kernels of the two context streams: RCCL's own stream at high
            # priority, so that they are not queued behind the workgroups of a saturating count launch
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, pg_options=opts)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    E = SimpleNamespace(torch=torch, dev=dev, local=local, dist=dist, rank=rank, world=world,
                        force_exchange=force_exchange, force_key_split=force_key_split)
    out, failures = measure(args, E, "strong" if world > 1 else "single")
    if world > 1 and os.environ.get("PALACE_BENCH_WEAK", "1") == "1":
        # the other reading of "N GPUs": one independent sample per GPU, no collective in the data path.  Every rank runs the
        # whole 1-GPU step on the (same) full sample; aggregate = N samples per max-over-ranks time.
        weak, _ = measure(args, E, "weak")
        if rank == 0:
            out["weak"] = weak
    if rank == 0:
        if failures:
            out["failed_checks"] = failures
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()
    if failures:
        print("bench.py: FAILED CHECKS: " + "; ".join(failures), file=sys.stderr)
        sys.exit(3)


def measure(args, E, leg):
    """One measured configuration.  leg = "single" (one GPU), "strong" (one sample over E.world GPUs: `value` of the N > 1 line)
    or "weak" (after the strong steps: every rank runs the whole one-GPU step on a full sample of its own, no collective in
    the data path; only the barrier and the max-over-ranks time use the process group).  Returns (dict, failed checks)."""
    torch, dev, local = E.torch, E.dev, E.local
    solo = leg == "weak"
    rank, world, dist = (0, 1, None) if solo else (E.rank, E.world, E.dist)
    sync_dist, sync_world = E.dist, E.world                               # barrier + max over ranks: always the real group
    force_exchange, force_key_split = (E.force_exchange and not solo), (E.force_key_split and not solo)
    collectives = world > 1 or force_exchange or force_key_split
    from palace_amd import capi, coder, multigpu       # (oracle/ is imported by the cpu_baseline leg only)

    hdr = coder.header_from_picks(np.random.Generator(np.random.PCG64(SEED)).integers(0, 6, size=32))
    # Streams.  A: eref (count + scan).  B: generateGraph (classify, resolve, copy numbers) and stage 04 (selection + matching: ~150
    # small latency-bound launches), high priority.  Stage 04 beside the saturating counting kernels takes 5.4 ms instead of the
    # 1.4 ms it takes alone and costs the count launch ~1 ms.  Measured in round 4 (tools/cu_mask_ab.sh, tools/r04c-e.sh; DESIGN.md
    # section 4): confining stage 04 to a CU subset (hipExtStreamCreateWithCUMask; PALACE_BENCH_STAGE04_CUS=n puts it on a stream S
    # of its own on the first n CUs, and keeps stream A off them) does not help -- on 32 CUs of its own it still takes 7.2 ms
    # (it is slowed by the memory system the counting kernels saturate, not by the CUs they occupy), the step is 10.9 ms either
    # way; holding it back behind the partition kernels or the whole count launch (PALACE_BENCH_STAGE04_LATE=l2|1) puts it on the
    # critical path (11.3 / 11.7 ms).  Default: stage 04 on stream B.
    # With collectives (N GPUs) A and B are torch streams the contexts run on (palace_ctx_create_on_stream), so that
    # torch.distributed's collectives are stream-ordered with the library's kernels and nothing waits on the host.
    mask_of = lambda k: int(os.environ[k], 16) if os.environ.get(k) else None      # tuning runs: PALACE_BENCH_CU_MASK_A / _B (hex)
    s04_cus = int(os.environ.get("PALACE_BENCH_STAGE04_CUS", "0"))
    if collectives:
        sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
        ctx, ctx_g = capi.Ctx(local, stream=sA.cuda_stream), capi.Ctx(local, stream=sB.cuda_stream)
    else:
        mask_a = mask_of("PALACE_BENCH_CU_MASK_A")
        if mask_a is None and 0 < s04_cus < 256 and os.environ.get("PALACE_BENCH_EXCLUDE_A", "1") == "1":
            mask_a = ((1 << 256) - 1) ^ ((1 << s04_cus) - 1)         # the eref stream keeps off stage 04's compute units
        ctx = capi.Ctx(local, cu_mask=mask_a)
        ctx_g = capi.Ctx(local, high_priority=os.environ.get("PALACE_BENCH_PRIO", "1") == "1", cu_mask=mask_of("PALACE_BENCH_CU_MASK_B"))
    ctx_s = capi.Ctx(local, cu_mask=(1 << s04_cus) - 1) if 0 < s04_cus < 256 else ctx_g
    if os.environ.get("PALACE_BENCH_GRAPHS", "0") == "1":     # stage 04 as two hipGraph launches per step (measured: host enqueue 1.25 -> 0.96 ms,
        ctx_s.match_set_option("launch_graphs", 1)            # the step 10.95 -> 11.08 ms: back to back the small kernels disturb the counting kernels more)
    for opt in ("iters_per_round", "first_group_rounds"):    # tuning runs only
        if os.environ.get("PALACE_OPT_" + opt.upper()):
            ctx_s.match_set_option(opt, int(os.environ["PALACE_OPT_" + opt.upper()]))
    # one GPU: a second eref context (own stream, own count table, own scratch) so that consecutive batches overlap
    depth = args.batches_in_flight if not collectives else 1
    ectx = [ctx] + [capi.Ctx(local) for _ in range(depth - 1)]
    for e in ectx:
        e.eref_set_coder(hdr)
        for opt in ("slab_bases", "bin1_ppl", "level1_parts"):        # tuning runs only (tools/): PALACE_OPT_BIN1_PPL=5 python bench.py
            if os.environ.get("PALACE_OPT_" + opt.upper()):
                e.eref_set_option(opt, int(os.environ["PALACE_OPT_" + opt.upper()]))
        if os.environ.get("PALACE_OPT_KEY_SHARE"):    # tuning runs only: count the share rank 0 of N would (results are then partial)
            e.eref_set_key_buckets(multigpu.key_buckets_of(0, int(os.environ["PALACE_OPT_KEY_SHARE"])))
    # Phase A across ranks, three schemes (palace_amd/multigpu.py phase_a_model; none measured on more than
    # one GPU yet): "replicate" -- every rank counts ALL reads, nothing is exchanged; "key_split" -- every rank holds all reads
    # and counts ITS 1/W of the key space, the ">= 3" plane slices are all-gathered (the partition kernels shrink to the key
    # arithmetic plus 1/W of the sorting and the bytes); "shard_reads" -- the reads are sharded and the partial count tables
    # exchanged (two planes to their owners, merge, all-gather): 0.5-0.9 GB out per rank whatever W is, but the counting itself
    # shards, which wins once a sample is large (5M contigs on 8 GPUs: ~11 ms against ~17 ms for the key split).  The scheme is
    # picked per run from the model; PALACE_BENCH_SCHEME=replicate|key_split|shard_reads forces one (rehearsals, A/B runs).
    long_mode = args.workload == "long"
    n_reads_total = 2 * (int(5e8 * (1.0 if long_mode else args.contigs / 1_000_000)) // READ_LEN)
    model = multigpu.phase_a_model(n_reads_total, world)
    best = multigpu.best_step(args.contigs, n_reads_total, world)       # the whole step, serial terms included: scheme + whether rank 0 counts
    scheme, forced = best["scheme"], None
    if force_exchange:
        forced = "shard_reads"
    elif force_key_split:
        forced = "key_split"
    elif world > 1 and os.environ.get("PALACE_BENCH_SCHEME", "auto") != "auto":
        forced = os.environ["PALACE_BENCH_SCHEME"]
        if forced not in ("replicate", "key_split", "shard_reads") or (forced == "key_split" and 64 % world):
            raise SystemExit(f"PALACE_BENCH_SCHEME={forced}: not a scheme for {world} ranks")
    if forced:
        scheme = forced
    if not collectives:
        scheme = "replicate"
    shard_reads = scheme == "shard_reads"
    # Stage 04 runs on rank 0.  Beside a count launch that saturates the device it takes 4-5x what it takes alone and grows with
    # the sample (5M contigs: 27 ms), so for large samples under the read-sharded scheme rank 0 takes NO reads: ranks 1 .. W-1
    # count 1/(W-1) each, rank 0's device has stage 04 (and its share of everything else) to itself.  PALACE_BENCH_RANK0_READS=0|1 forces.
    rank0_counts = True
    if shard_reads and world > 2:
        env0 = os.environ.get("PALACE_BENCH_RANK0_READS", "auto")
        rank0_counts = (env0 == "1") if env0 in ("0", "1") else (best["rank0_counts"] if scheme == best["scheme"] else
                                                                  multigpu.step_model(args.contigs, n_reads_total, world, scheme, False)["step_ms"] >=
                                                                  multigpu.step_model(args.contigs, n_reads_total, world, scheme, True)["step_ms"])
    read_weights = None if (rank0_counts or not shard_reads) else [0.0] + [1.0] * (world - 1)
    model.update(choice_in_force=scheme, forced=bool(forced), rank0_counts=bool(rank0_counts), choice=best["scheme"],
                 step=multigpu.step_model(args.contigs, n_reads_total, world, scheme, rank0_counts),
                 step_alternatives=[multigpu.step_model(args.contigs, n_reads_total, world, sch, True) for sch in model["ms"]])
    sample = make_sample(torch, dev, args.contigs, args.refs, rank if shard_reads else 0, world if shard_reads else 1, long_mode, read_weights)
    gs = make_graph_sample(torch, dev, args.contigs, sample["n_pairs_total"], rank, world, long_mode)
    if collectives:                                 # avgDepth is a pipeline input: computed once from all shards
        tot = torch.tensor([float(gs["col"]["ref_len"].sum().item())], device=dev, dtype=torch.float64)
        dist.all_reduce(tot)
        gs["avg_depth"] = float(f"{tot.item() / gs['lens'].sum():.6g}")
    torch.cuda.synchronize()
    one_min, three_min = capi.window_minimums(0.9, 0.85)
    L = capi.lib()
    P = lambda t: t.data_ptr()
    n_side, n_refs, nt = sample["n_reads_side"], sample["n_refs"], args.contigs
    # refs shard by cumulative length across ranks (eref Phase B); every rank holds the whole (small) DB
    r_lo, r_hi = multigpu.split_by_weight(sample["ref_lens"], rank, world)
    rows = torch.zeros((n_refs, 4), dtype=torch.int32, device=dev)
    rows_host = torch.zeros((n_refs, 4), dtype=torch.int32).pin_memory()
    cn_host = torch.zeros(args.contigs, dtype=torch.int32).pin_memory()
    consumed = torch.zeros(nt, dtype=torch.int64, device=dev)
    cn_dev = torch.zeros(nt, dtype=torch.int32, device=dev)
    cand_cap = gs["n"] + gs["n_sa"] + 1
    cands = torch.zeros((cand_cap, 64), dtype=torch.uint8, device=dev)
    edges = torch.zeros((cand_cap, 32), dtype=torch.uint8, device=dev)
    cols = capi.BamCols(gs["n"], *(P(gs["col"][k]) for k in ("tid", "pos", "mtid", "mpos", "nm", "ref_len", "read_len",
                                                           "clip_s", "clip_e", "flag", "mapq", "qkey")), P(gs["sa_off"]))
    prm = capi.GraphParams.default()
    # per-contig offsets into the sorted FASTG keys (once per sample, like the keys): the classify kernel's look-ups start there
    fastg_first = torch.zeros(nt + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    capi._check(L.palace_graph_fastg_offsets(ctx_g.h, P(gs["fastg"]), gs["n_fastg"], nt, P(fastg_first)), "fastg offsets")
    ctx_g.sync()
    # stage 04 resident: the per-sample inputs of filter_graph.py and matching -l, parsed once like the BAM columns
    gs["side"] = make_side_inputs(gs)
    stage04 = None
    if rank == 0:
        stage04 = capi.Stage04(ctx_s, gs["side"]["seed"], gs["lens"].astype(np.int32), gs["trank"].cpu().numpy(), gs["lens"].astype(np.int32),
                               gs["side"]["path_off"], gs["side"]["path_tok"], 5)
    n_edges_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    exch = multigpu.Exchange(torch, dist, rank, world) if collectives else None
    # N GPUs: torch ops and collectives are issued with a CONTEXT's stream as torch's current stream (the library's streams
    # wrapped as torch.cuda.ExternalStream): the collectives of eref are stream-ordered behind the counting kernels on stream A,
    # those of generateGraph behind classify on stream B, and the host waits for nothing between them -- no synchronize at the
    # hand-overs, no count read back to size a gather (rows travel padded to a width the previous step established)
    if exch:
        on_a, on_b = (lambda: torch.cuda.stream(sA)), (lambda: torch.cuda.stream(sB))
        planes = [torch.zeros(1 << 29, dtype=torch.uint8, device=dev) for _ in range(3)]   # torch-owned so RCCL
        ctx.eref_table_attach([t.data_ptr() for t in planes])                                # can address them
        ref_ranges = [multigpu.split_by_weight(sample["ref_lens"], r, world) for r in range(world)]
        scratch_consumed = torch.zeros(nt, dtype=torch.int64, device=dev)
        low_plane = torch.zeros(1 << 29, dtype=torch.uint8, device=dev)
        n_c_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        counts_host = torch.zeros(world, dtype=torch.int64).pin_memory()
        gat = {"width": 0, "rows": None, "edges": None}          # padded gather of the candidates: rows per rank, buffers

        def pack_fn():                              # two planes per peer instead of three (include/palace_hip.h); stream A
            ctx.eref_table_pack_low(low_plane.data_ptr())
            return low_plane

        def merge_fn(parts, n_parts, slice_off, slice_bytes, packed=False):      # stream A, behind the all-to-all issued on it
            ctx.eref_table_merge_slices(parts.data_ptr(), n_parts, slice_off, slice_bytes, packed)
    last = {}
    seen = {"graph": set(), "rows": set(), "steps": 0}     # result digests of the untimed steps (warm-up, soak): one value each, or the step is not repeatable
    h_last = {}
    host_ms = {}
    ref_off_local = sample["ref_off"][r_lo:r_hi + 1].contiguous()
    # Per-DB probe index of this rank's refs, built once outside the timed region: the reference, too, scans a
    # DB through the index file it built on first use (<fasta>.k32.index.dat), and the CPU baseline below is
    # timed with its index prebuilt as well.
    probe_index = ctypes.c_void_p()
    capi._check(L.palace_eref_probe_index_build(ctx.h, P(sample["ref_bases"]), P(ref_off_local), r_hi - r_lo,
                                                sample["ref_total"], ctypes.byref(probe_index)), "probe index")
    # the count launch of a step is its final count (below): with --fused-probe 1 (or PALACE_BENCH_FUSED_PROBE=1) channel 0 of Phase B
    # rides along in the count kernel while each fine bucket's ">= 3" slice is in LDS (palace_eref_attach_probe_index)
    fused_probe = depth == 1 and os.environ.get("PALACE_BENCH_FUSED_PROBE", str(args.fused_probe)) == "1"
    if fused_probe:
        capi._check(L.palace_eref_attach_probe_index(ctx.h, probe_index), "attach probe index")

    # the reads in the form the step counts them from (resident before the timed region, like every other input)
    packed = None
    if args.reads == "packed":
        nb = int(L.palace_eref_packed_bytes(2 * n_side * READ_LEN))
        packed = [torch.zeros(nb, dtype=torch.uint8, device=dev) for _ in range(3)]
        torch.cuda.synchronize()                     # torch fills them on ITS stream; the library writes them on the context's
        capi._check(L.palace_eref_pack_reads(ctx.h, P(sample["r12"]), P(sample["read_off"]), 2 * n_side, None, 2 * n_side * READ_LEN,
                                             *(P(t) for t in packed)), "pack")
        ctx.sync()
    # one GPU (and N GPUs that each count all reads): the count of a step is the only one between its reset and its scan, so the
    # two lower planes of the table need not leave the LDS (include/palace_hip.h, option final_count)
    final_count = not shard_reads and os.environ.get("PALACE_BENCH_FINAL", "1") == "1"      # (=0: A/B runs)
    # key split: rank r counts only the keys of ITS 1/W of the key space (they are dropped where they are made: the partition
    # kernels move 1/W of the bytes, the key arithmetic stays) and the ">= 3" plane slices are all-gathered -- one collective of
    # 512 MiB / W per rank instead of the table exchange
    key_split = bool(exch) and scheme == "key_split"
    if key_split:
        ctx.eref_set_key_buckets(multigpu.key_buckets_of(rank, world))       # mirrored pairs of buckets: equal key mass per rank
    for e in ectx:
        e.eref_set_option("final_count", 1 if final_count else 0)
        if os.environ.get("PALACE_BENCH_STAGE04_LATE", args.stage04_hold) == "l2":
            e.eref_set_option("mark_before_count_kernel", 4091)
        if os.environ.get("PALACE_BENCH_STAGE04_LATE", args.stage04_hold) == "l1":
            e.eref_set_option("mark_before_level2", 4091)
    rows_l = [rows] + [torch.zeros_like(rows) for _ in range(depth - 1)]
    rows_host_l = [rows_host] + [torch.zeros((n_refs, 4), dtype=torch.int32).pin_memory() for _ in range(depth - 1)]
    seq = {"n": 0, "pending": None, "counted": None, "last": 0, "of_timed": {}}           # running batch number; the batch whose rows are still on their way

    def step(i, timed):
        m = 8 * i
        tot_b = n_side * READ_LEN
        slot = seq["n"] % depth                        # which eref context / rows buffers this batch uses
        seq["n"] += 1
        ctx, rows, rows_host = ectx[slot], rows_l[slot], rows_host_l[slot]
        if timed: seq["of_timed"][i] = slot
        # ---------------- eref: runs asynchronously on its own stream ----------------
        def eref_head():
            capi._check(L.palace_eref_table_reset(ctx.h), "reset")
            if depth > 1 and seq["counted"] is not None:
                # this batch's counting kernels start when the previous batch's are done (two count launches side by side would
                # only share the device); what then runs beside them is the previous batch's Phase B
                ctx.wait_for_mark(ectx[seq["counted"]], 4095)
            if timed: ctx.mark(m)
            # both FASTQ sides as one read set: one binning pass, the plane slices are loaded and stored once
            if packed:
                capi._check(L.palace_eref_count_reads_packed(ctx.h, *(P(t) for t in packed), 2 * tot_b, 2 * n_side), "count")
            else:
                capi._check(L.palace_eref_count_reads(ctx.h, P(sample["r12"]), P(sample["read_off"]), 2 * n_side, None, 2 * tot_b), "count")
            if timed: ctx.mark(m + 1)
            ctx.mark(4095)                             # "the counting kernels are done" (the next batch's, and a held-back stage 04, wait for it)
            seq["counted"] = slot

        skip_eref = os.environ.get("PALACE_BENCH_SKIP_EREF") == "1"      # tuning runs only: stream B alone on the device
        if not skip_eref:
            eref_head()                                # launched first: generateGraph + matching (and, on N GPUs, their small collectives at
                                                       # RCCL's high-priority stream) overlap the counting kernels

        def eref_tail():
            if exch and shard_reads:                   # count-table exchange (RCCL) on stream A, then Phase B on this rank's refs
                with on_a():
                    exch.merge_planes(planes, merge_fn, pack_fn)
            elif key_split:                            # every rank counted its range of the key space: gather the ">= 3" plane
                with on_a():
                    exch.gather_key_buckets(planes[2])
            if timed: ctx.mark(m + 2)
            capi._check(L.palace_eref_scan_refs_indexed(ctx.h, probe_index, P(sample["ref_bases"]), P(ref_off_local), r_hi - r_lo,
                                                        sample["ref_total"], one_min, three_min, P(rows) + 16 * r_lo), "scan")
            if timed: ctx.mark(m + 3)
            if exch:
                with on_a():
                    exch.gather_ranges(rows, ref_ranges)

        if not exch and not skip_eref:
            eref_tail()                                # one GPU: queue Phase B right behind the counting kernels
        # ---------------- generateGraph + filter + matching (second stream; overlaps the eref kernels) ----------------
        # One wait in the middle (the candidate count sizes the tables of what follows), one at the end; everything else is
        # enqueued: classify -> resolve (edge count stays on the device) -> copy numbers -> filter_graph.py's selection ->
        # matching on the filtered graph, all in HBM.
        g = ctx_g
        if seq.get("graph_pending"):                   # --graph-lag 1: the step before's decomposition is collected now, with this step's
            seq.pop("graph_pending")()                 # counting kernels already enqueued (stage 04's buffers are then free for this step)
        th0 = time.perf_counter()
        if timed: g.mark(m)
        capi._check(L.palace_memset(g.h, P(consumed), 0, nt * 8), "memset")
        n_c, n_b = ctypes.c_int64(), ctypes.c_int64()
        capi._check(L.palace_graph_classify_ix(g.h, ctypes.byref(cols), P(gs["sa"]), nt, P(gs["tlen"]), P(gs["trank"]),
                                               P(gs["fastg"]), gs["n_fastg"], P(fastg_first), ctypes.byref(prm), gs["ord_base"], P(consumed),
                                               P(cands), cand_cap, ctypes.byref(n_c), ctypes.byref(n_b)), "classify")
        if timed: g.mark(m + 1)
        all_c, n_cands, n_border, e_buf, cons_for_quirk = cands, n_c.value, n_b.value, edges, consumed
        n_cands_sample = n_cands
        if exch:
            # every rank resolves the same gathered candidates (only rank 0's quirk sums join the reduce).  A rank decides ITS
            # candidates of the exp-underflow zone before they travel (host libm; none in the default workload), so no count of
            # them is exchanged.  The gather itself: rows padded to `width` per rank, zero rows are candidates resolve ignores;
            # the width is what the step before saw (+ 1/8), the per-rank counts come back with the step's results and are
            # checked then.  The first step (and one whose counts outgrew the width) takes the exact gather, which reads them.
            if n_border:
                capi._check(L.palace_graph_score_border(g.h, P(cands), n_cands, n_border, ctypes.byref(prm)), "score border")
                n_border = 0
            with on_b():
                if gat["width"] <= 0:
                    all_c, n_cands = exch.gather_varlen(cands, n_cands)
                    n_cands_sample = n_cands
                    gat["learn"] = True
                else:
                    n_c_dev.fill_(n_cands)
                    gat["rows"], counts_dev = exch.gather_padded(cands, n_c_dev, gat["width"], gat["rows"])
                    counts_host.copy_(counts_dev, non_blocking=True)
                    all_c, n_cands = gat["rows"], world * gat["width"]
                    gat["learn"] = False
                if gat["edges"] is None or gat["edges"].shape[0] < max(n_cands, cand_cap):
                    gat["edges"] = torch.zeros((max(n_cands, cand_cap), 32), dtype=torch.uint8, device=dev)
                e_buf = gat["edges"]
                if rank != 0:
                    scratch_consumed.zero_()
                    cons_for_quirk = scratch_consumed
        capi._check(L.palace_graph_resolve_ex(g.h, P(all_c), n_cands, n_border, gs["n_total"], ctypes.byref(prm), P(cons_for_quirk),
                                              P(e_buf), max(1, n_cands), P(n_edges_dev), None), "resolve")
        if exch:
            with on_b():
                exch.reduce_sum(consumed)
        capi._check(L.palace_graph_copy_numbers(g.h, P(consumed), P(gs["tlen"]), nt, gs["avg_depth"], P(cn_dev)), "cn")
        if ctx_s is not g and stage04 is not None:  # stage 04 has a stream of its own: it starts when the copy numbers are there
            g.mark(4092)
            ctx_s.wait_for_mark(g, 4092)
        if timed: g.mark(m + 2); ctx_s.mark(m + 2)
        if stage04 is not None:                     # rank 0 owns the (small) stage; its result is what the sample's all_result holds
            # Stage 04 is ~150 small latency-bound launches beside the bandwidth-bound counting kernels; each costs those kernels a
            # few microseconds (kernel boundaries write back the L2 lines the partition kernels combine their stores in): about
            # 1 ms per step, measured.  Holding the rounds back until the counting kernels are done (PALACE_BENCH_STAGE04_LATE=1:
            # palace_stage04_match_after) leaves those undisturbed but puts the rounds on the critical path -- 14.8 against 12.8 ms.
            late = os.environ.get("PALACE_BENCH_STAGE04_LATE", args.stage04_hold) if not exch and not skip_eref else "0"
            # (diagnosis only, timed steps only -- the line then fails its own checks on purpose: PALACE_BENCH_DIAG_SKIP=stage04|match
            # leaves stage 04 / its matching rounds out, to see what they cost the counting kernels beside them)
            diag_skip = os.environ.get("PALACE_BENCH_DIAG_SKIP") if timed else None
            if diag_skip != "stage04":
                stage04.filter(P(e_buf), P(n_edges_dev), max(1, n_cands))
            if diag_skip and os.environ.get("PALACE_BENCH_DISTURB"):                  # "mode:ops:launches:slots:blocks"
                dm, dops, dl, dslots, dblk = (int(x) for x in os.environ["PALACE_BENCH_DISTURB"].split(":"))
                if "disturb_buf" not in seq:
                    seq["disturb_buf"] = torch.full((dslots,), -1, dtype=torch.int64, device=dev)
                    torch.cuda.synchronize()
                capi._check(L.palace_diag_disturb(ctx_s.h, P(seq["disturb_buf"]), dslots, dops, dm, dl, dblk), "disturb")
            # ("1": the rounds wait for the whole count launch; "l2": for its partition kernels -- they then run beside the count
            # kernel and Phase B only)
            if diag_skip is None:
                stage04.match(P(e_buf), P(cn_dev), 10, False, True, after=(ctx, 4095) if late == "1" else (ctx, 4091) if late in ("l1", "l2") else None)
        if timed: ctx_s.mark(m + 3)
        th1 = time.perf_counter()
        if timed:
            host_ms["graph_enqueue_incl_classify_wait"] = host_ms.get("graph_enqueue_incl_classify_wait", 0.0) + 1e3 * (th1 - th0) / args.steps
        last.update(n_cands=int(n_cands_sample))

        def finish_graph():
            """the end of stream B: wait for the decomposition, take the result views; on untimed steps also the bookkeeping
            (counts, result digest) that the JSON line reports"""
            if stage04 is None:
                return
            if timed and os.environ.get("PALACE_BENCH_DIAG_SKIP"):
                ctx_s.sync()
                return
            t0_ = time.perf_counter()
            res, contig_of = stage04.result()
            if timed:
                host_ms["stage04_result_wait_and_copy"] = host_ms.get("stage04_result_wait_and_copy", 0.0) + 1e3 * (time.perf_counter() - t0_) / args.steps
            if not timed or "n_comp" not in last:
                cnt = stage04.counts()
                n_e = int(n_edges_dev.item())
                capi._check(L.palace_d2h(g.h, cn_host.data_ptr(), P(cn_dev), nt * 4), "d2h")
                h_edges = e_buf[:n_e].cpu().numpy().view(capi.EDGE_DTYPE).reshape(-1)
                h_last.update(edges=h_edges, cn=cn_host.numpy().copy())
                last.update(n_edges=n_e, n_junc=cnt["juncs"], n_kept_junc=cnt["kept_pass2"] + cnt["kept_pass3_more"], n_arcs=cnt["arcs"],
                            n_segs_filtered=cnt["segs_filtered"], n_segs_rescued=cnt["segs_rescued"],
                            n_comp=res.n + res.n_bare, n_cycles=int(np.asarray(res.kind).sum()),
                            n_multi=int(((np.asarray(res.off)[1:] - np.asarray(res.off)[:-1]) > 1).sum()))
                # digest of the step's results: the lines of an N-GPU run and of the 1-GPU run must carry the same one
                e64 = np.ascontiguousarray(h_edges).view(np.uint64).reshape(-1, 4)
                e64 = e64[np.lexsort((e64[:, 3], e64[:, 2], e64[:, 1], e64[:, 0]))]
                hsh = hashlib.sha256()
                for arr in (e64, h_last["cn"], np.asarray(res.off), np.asarray(res.verts), np.asarray(res.kind), np.asarray(res.iter),
                            np.asarray(res.bare), np.asarray(contig_of)):
                    hsh.update(np.ascontiguousarray(arr).tobytes())
                last["digest_graph"] = hsh.hexdigest()[:16]
                seen["graph"].add(last["digest_graph"])
            h_last["result"] = (res, contig_of)                   # views, valid until the next match call

        if exch and not skip_eref:
            eref_tail()                                # the plane exchange / gather, Phase B and the row gather, all enqueued on stream A
        if args.graph_lag and not exch:
            seq["graph_pending"] = finish_graph
        else:
            finish_graph()
        # ---------------- join: eref results to the host ----------------
        # the rows of THIS batch are requested; with two batches in flight the ones waited for are the previous batch's (whose
        # Phase B ran beside this batch's counting kernels), with one they are this batch's
        capi._check(L.palace_d2h_async(ctx.h, rows_host.data_ptr(), P(rows), rows.numel() * 4), "d2h")
        ctx.mark(4094)
        if depth == 1:
            ctx.mark_wait(4094)
            if not timed:
                seen["rows"].add(hashlib.sha256(rows_host.numpy().tobytes()).hexdigest()[:16])
                seen["steps"] += 1
        else:
            if seq["pending"] is not None:
                ectx[seq["pending"]].mark_wait(4094)
            seq["pending"] = slot
        seq["last"] = slot
        if exch:
            g.mark(4093)
            g.mark_wait(4093)                          # stream B has drained on every rank (only rank 0 waited for a stage-04 result)
            cnt = exch.last_counts if gat["learn"] else [int(x) for x in counts_host.tolist()]
            if not gat["learn"] and max(cnt) > gat["width"]:
                gat["width"] = 0                       # a rank had more candidates than the padded gather carried: this step again, exactly
                return step(i, timed)
            last["n_cands"] = int(sum(cnt))
            gat["width"] = max(gat["width"], (max(cnt) + max(cnt) // 8 + 256) // 256 * 256)

    def barrier():
        if seq.get("graph_pending"):
            seq.pop("graph_pending")()
        for e in ectx:
            e.sync()
        seq["pending"] = None
        ctx_g.sync()
        ctx_s.sync()
        torch.cuda.synchronize()
        if sync_dist is not None:
            sync_dist.barrier()
            torch.cuda.synchronize()

    torch.cuda.synchronize()                         # every buffer torch made above is filled before a library stream touches it
    for _ in range(args.warmup):
        step(0, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    barrier()
    dt = time.perf_counter() - t0
    if sync_world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        sync_dist.all_reduce(tmax, op=sync_dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_step = 1e3 * dt / args.steps
    # soak: the K timed steps above are what `value` is computed from; when they took less than --soak-seconds the same
    # step keeps running (untimed for `value`) so that an outside GPU-activity sampler has something to see
    soak = None
    if dt < args.soak_seconds and not exch and sync_world == 1:
        t1, n_soak = time.perf_counter(), 0
        while time.perf_counter() - t1 < args.soak_seconds - dt:
            for _ in range(10):
                step(0, False)
            barrier()
            n_soak += 10
        soak = dict(steps=n_soak, seconds=time.perf_counter() - t1, ms_per_step=1e3 * (time.perf_counter() - t1) / max(1, n_soak),
                    note="untimed steps also lexsort and sha256 the results for `result_digest` (bookkeeping): not comparable with ms_per_step")
    K = range(args.steps)
    if os.environ.get("PALACE_BENCH_SKIP_EREF") == "1":
        count_each, count_ms, merge_ms, scan_ms = [1.0], 1.0, 0.0, 0.0
    else:
        E = lambda i: ectx[seq["of_timed"][i]]                                      # the context timed step i ran on
        count_each = [E(i).mark_elapsed(8 * i, 8 * i + 1) for i in K]
        count_ms = np.mean(count_each)                                              # one launch per step (both FASTQ sides)
        merge_ms = np.mean([E(i).mark_elapsed(8 * i + 1, 8 * i + 2) for i in K])
        scan_ms = np.mean([E(i).mark_elapsed(8 * i + 2, 8 * i + 3) for i in K])
    classify_ms = np.mean([ctx_g.mark_elapsed(8 * i, 8 * i + 1) for i in K])
    resolve_ms = np.mean([ctx_g.mark_elapsed(8 * i + 1, 8 * i + 2) for i in K])
    stage04_ms = np.mean([ctx_s.mark_elapsed(8 * i + 2, 8 * i + 3) for i in K])
    r = rows_host_l[seq["last"]].numpy()
    reported = int(((r[:, 1] > 0) & (r[:, 1].astype(np.float32) / r[:, 2].astype(np.float32) > 0.75)).sum())

    failures, out = [], None
    if rank == 0:
        L.palace_version.restype = ctypes.c_char_p
        version = L.palace_version().decode()
        fused_now = fused_probe and final_count and not key_split
        traffic, traffic_src, stage_traffic = profiled_traffic(args, world, version, fused_now)
        alg_bytes = (READ_LEN + 6 * (READ_LEN - 31)) * 2 * n_side        # per launch (both FASTQ sides of this rank)
        # when Phase B's channel-0 probe rides along in the count kernel, its look-ups (1 B per ref position) are work of this launch
        probe_bytes = sum(int(l) - 31 for l in sample["ref_lens"][r_lo:r_hi]) if fused_now else 0
        achieved = (alg_bytes + probe_bytes) / (max(count_ms, 1e-6) * 1e-3) / 1e9          # (a rank 0 that takes no reads reports 0)
        out = {
            "metric": "contigs/sec eref+generate_graph+matching, 1M-contig synth, 1/2/4/8 GPU",       # BASELINE.json, verbatim
            "value": args.contigs / (ms_step * 1e-3), "unit": "contigs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{args.contigs}-contig synthetic sample: {n_refs} phage refs ({sample['ref_total']} bp), "
                                   f"{2 * sample['n_pairs_total']} reads x {READ_LEN} bp, {gs['n_total']} primary BAM records, "
                                   f"{gs['n_fastg']} FASTG links",
                       "stages": ["eref", "generateGraph", "matching"], "seed": SEED, "workload_kind": args.workload,
                       "batches_in_flight": depth, "graph_lag": args.graph_lag, "stage04_hold": args.stage04_hold,
                       "reads": ("packed in HBM: two bits per base + 32-mer start mask, 0.375 B/base (palace_eref_count_reads_packed)" if packed else
                                 "ASCII in HBM, 1 B/base (palace_eref_count_reads)") + ("; count keeps only the '>= 3' plane (final_count)" if final_count else ""),
                       "parallelism": "1 GPU" if world == 1 else (f"reads/records/refs sharded over {world} GPUs (RCCL)" + ("" if rank0_counts else f"; rank 0 takes no reads: stage 04 has its device to itself, ranks 1-{world - 1} count") if shard_reads else
                                                                   f"records/refs and the key space sharded over {world} GPUs (RCCL): every GPU counts its 1/{world} of the keys of all reads, the '>= 3' plane is all-gathered" if key_split else
                                                                   f"records/refs sharded over {world} GPUs (RCCL), reads counted on every GPU"),
                       "parallelism_model": model,
                       "ref_index": "per-DB probe index prebuilt, as the reference's cached <fasta>.k32.index.dat (8 B/position in HBM)"
                                    + ("; its channel-0 probe rides along in the count kernel" if fused_probe and final_count and not key_split else ""),
                       "refs_reported": reported, "refs_present": int(len(sample["present"])),
                       "result_digest": {"eref_rows": hashlib.sha256(np.ascontiguousarray(r).tobytes()).hexdigest()[:16],
                                         "graph_and_components": last.get("digest_graph"),
                                         "identical_over_untimed_steps": (len(seen["graph"]) <= 1 and len(seen["rows"]) <= 1) if seen["steps"] else None,
                                         "untimed_steps_compared": seen["steps"],
                                         "note": "sha256 prefixes of the last step's results; equal for every --gpus N"},
                       "graph": {k: last.get(k) for k in ("n_cands", "n_edges", "n_junc", "n_kept_junc", "n_segs_filtered", "n_segs_rescued", "n_arcs",
                                                          "n_comp", "n_cycles", "n_multi")},
                       "stage04": "filter_graph.py's selection (seeds, 1- and 2-hop junctions, contigs.paths rescue) and matching -i 10 -l contigs.paths "
                                  "on the filtered graph, both on the device (palace_stage04_*), as palace:566-591 runs them on files"},
            "roofline": {"bound": "hbm", "kernel": ("eref count_reads_packed (bin1 + bin2 + lds_count kernels of one launch" + (", Phase B's channel-0 probe fused into lds_count)" if fused_now else ")")) if packed else
                                   "eref count_reads (streams + bin1 + bin2 + lds_count kernels of one launch)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # PMC (separate rocprofv3 passes, profiles/r01q_end_state_fused_launch.md): FETCH_SIZE x2 + WRITE_SIZE of
                         # bin1 + bin2 + lds_count per launch; only valid for the default workload on one GPU
                         "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "avg_launch_ms": count_ms, "algorithmic_bytes_per_launch": alg_bytes + probe_bytes,
                         "algorithmic_bytes_note": f"864 B per 150-bp read x {2 * n_side} reads" + (f" + {probe_bytes} B: the channel-0 look-ups of Phase B (1 B per ref "
                                                   "position), which this launch's count kernel does while a bucket's slice is in LDS" if probe_bytes else "")},
            # the other stages of the step against the same roofline (SURVEY.md section 8(d) algorithmic bytes; live event times of
            # this run; PMC traffic of the committed profile when it is of this build and workload)
            "roofline_stages": roofline_stages(dict(
                phase_b=(sum(int(l) + (2 if fused_now else 3) * (int(l) - 31) for l in sample["ref_lens"][r_lo:r_hi]), scan_ms,
                         "l + 3(l - 31) B per ref: a byte per base, three 1-byte look-ups per position" +
                         (" -- minus the channel-0 look-ups, which the count launch did" if fused_now else "")),
                classify=(52 * gs["n"] + 64 * gs["n_sa"], classify_ms, "52 B per primary record + 64 B per SA item"),
                resolve=(64 * int(last.get("n_cands", 0)) + 16 * int(last.get("n_cands", 0)), resolve_ms,
                         "64 B per candidate read + 16 B per evidence written"),
                stage04=((32 * int(last.get("n_segs_filtered", 0)) + 24 * int(last.get("n_kept_junc", 0))) * 10, stage04_ms,
                         "32 B per SEG + 24 B per JUNC of the filtered graph, read once per pass, -i 10 passes")), stage_traffic),
            "library": version,
            # eref's unit is a read, generateGraph's a BAM record -- the same step in those units
            "rates": {"reads_per_s": 2 * sample["n_pairs_total"] / (ms_step * 1e-3), "bam_records_per_s": gs["n_total"] / (ms_step * 1e-3),
                      "read_bases_per_s": 2 * sample["n_pairs_total"] * READ_LEN / (ms_step * 1e-3)},
            "stage_ms": {"eref_count_each_step": [round(float(x), 3) for x in count_each], "eref_count_both_sides": count_ms, "eref_table_merge": merge_ms, "eref_scan_refs": scan_ms,
                         "graph_classify": classify_ms, "graph_resolve": resolve_ms, "graph_filter_and_matching_on_device": stage04_ms,
                         **{"host_" + k: v for k, v in host_ms.items()},
                         "note": "eref runs on one HIP stream, generateGraph + matching on another; they overlap"},
        }
        if soak:
            out["soak"] = soak
        if world == 1 and not solo and not args.no_e2e:
            import shutil
            import tempfile
            work = os.environ.get("PALACE_BENCH_WORK_DIR")          # (tools/e2e_repeat.sh: the directory it made for this run)
            if work:
                os.makedirs(work, exist_ok=True)
            else:
                work = tempfile.mkdtemp(prefix="palace_e2e_", dir=os.environ.get("PALACE_BENCH_TMP", "/tmp"))
            try:
                paths = write_e2e_inputs(torch, sample, gs, hdr, work)
                n_junc = int((h_last["edges"]["counts"].sum(axis=1) >= 5).sum())
                # what the resident step's result reads as text: linear ++ cycles without duplicates (palace:594-600)
                from palace_amd import stage04_io
                lin, cyc = stage04_io.matching_text(*h_last["result"], gs["names"], self_loops=True, break_cycles=False)
                cl = cyc.splitlines(keepends=True)
                pairs = list(dict.fromkeys(zip(cl[0::2], cl[1::2] + (["\n"] if len(cl) % 2 else []))))      # remove_cycle_dup.py:3-30
                out["e2e"] = run_e2e(paths, gs["avg_depth"], args.contigs, r, n_junc, lin + "".join(a + b for a, b in pairs))
            except Exception as e:                   # never let this leg break the headline line
                out["e2e"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                paths = None
        else:
            work = paths = None
        try:
            if world == 1 and not solo and not args.no_cpu_baseline:         # (rank 0 at N = 1 only: the contract; the other ranks would wait for it)
                res_v, contig_of = h_last["result"]
                seg_flags, edge_flags = stage04.flags(len(h_last["edges"]))
                out["cpu_baseline"] = cpu_baseline(torch, sample, gs, hdr, args.cpu_sample_frac,
                                                   dict(contig_of=np.asarray(contig_of).copy(), cn=h_last["cn"], edges=h_last["edges"], edge_flags=edge_flags),
                                                   bam_path=paths["bam"] if paths else None)
        finally:
            if work and not os.environ.get("PALACE_BENCH_KEEP"):              # (tools/eref_cli_repeat.sh re-runs the executables on these files)
                import shutil
                shutil.rmtree(work, ignore_errors=True)
        # a line whose own cross-checks failed is still printed, but the run does not pass: wrong refs, the executables on the
        # files disagreeing with the resident step (or the leg raising), results that differ from step to step
        e2e = out.get("e2e")
        if reported != len(sample["present"]) and args.contigs >= 1_000_000 and args.refs == 5000 and os.environ.get("PALACE_BENCH_SKIP_EREF") != "1" \
                and not os.environ.get("PALACE_OPT_KEY_SHARE"):     # (below 1M contigs the read depth leaves a few present refs short; a tuning run that counts one rank's key share is partial by design)
            failures.append(f"refs_reported {reported} != refs_present {len(sample['present'])}")
        if out["config"]["result_digest"]["identical_over_untimed_steps"] is False:
            failures.append("result digests differ between untimed steps")
        if e2e is not None:
            if "error" in e2e:
                failures.append("e2e leg raised: " + e2e["error"])
            else:
                for k in ("agrees_with_resident_step", "all_result_identical_to_resident_step"):
                    if e2e.get(k) is not True:
                        failures.append(f"e2e.{k} is {e2e.get(k)}")
                one = e2e.get("one_process_stage04") or {}
                if one.get("files_identical_to_the_chain") is not True:
                    failures.append("e2e.one_process_stage04: " + str(one.get("error", "files differ from the chain's")))
    capi._check(L.palace_eref_probe_index_free(ctx.h, probe_index), "probe index free")
    if stage04 is not None:
        stage04.close()
    for e in ectx[1:]:
        e.close()
    if ctx_s is not ctx_g:
        ctx_s.close()
    ctx.close()
    ctx_g.close()
    if solo:
        # the weak record: N independent samples (one per GPU, the whole one-GPU step each), aggregate rate over the slowest rank
        return dict(scaling="weak", n_gpus=sync_world, value=sync_world * args.contigs / (ms_step * 1e-3), unit="contigs/s",
                    ms_per_step=ms_step, steps=args.steps, samples_per_step=sync_world,
                    eref_count_ms=float(count_ms), result_digest=out["config"]["result_digest"],
                    note="every rank runs the one-GPU step on a full sample of its own (here: the same synthetic sample on every rank), "
                         "no collective in the data path; value = N x contigs / max-over-ranks time per step"), failures
    return out, failures


if __name__ == "__main__":
    main()
