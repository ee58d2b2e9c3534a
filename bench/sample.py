"""The synthetic sample of the bench (SURVEY.md section 8(d) shapes), generated on the device it is given: reads and phage refs
(eref), one primary BAM record per read as decoded columns (generateGraph), the side inputs of filter_graph.py / matching."""
import numpy as np

SEED = 20261003
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
READ_LEN = 150
_T0 = [None]


def progress(msg):
    """one line on stderr with the seconds since the first call: a run that takes minutes says where it is (and a watcher that
    takes silence for a hang sees it alive)"""
    import sys
    import time
    if _T0[0] is None:
        _T0[0] = time.perf_counter()
    print(f"[bench {time.perf_counter() - _T0[0]:7.1f} s] {msg}", file=sys.stderr, flush=True)


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

