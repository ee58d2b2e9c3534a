"""BASELINE.json configs at their own size on one MI355X (driver-visible: these run in `pytest -m gpu`).

  configs[1]  50k contigs / 5k-phage ref DB: the eref CLI and the C ABI (unindexed and indexed scan) against stdout,
              index file and genome.len file of the COMPILED REFERENCE run on the same bytes (tests/golden/eref_50k.npz,
              made by tests/golden/make_eref_50k_golden.py in the build container).
  configs[3]  5M contigs (33 M reads, 5 Gbase) on ONE GPU: the slab path at its default slab size -- order independence,
              equality with direct-mode counting on a slab-sized part.  (The 8-GPU leg is the driver's SCALE run.)
  configs[4]  long contigs (100k contigs, N50 ~ 50 kb, tail > 120 kb): generateGraph through the C ABI at full size, a
              60k-record part compared exactly with the oracle (exp-underflow candidates present), shard independence.
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as orc
from palace_amd import capi, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "palace_amd", "bin")


# ------------------------------------------------------------------------------------------------
# configs[1]
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg1(tmp_path_factory):
    g = np.load(os.path.join(ROOT, "tests", "golden", "eref_50k.npz"))
    seed, n_refs, n_pairs = (int(x) for x in g["params"])
    fa, fq1, fq2, refs, r1, r2 = synth.eref_config_inputs(seed, n_refs, n_pairs, arrays=True)
    for key, b in (("sha256_db_fa", fa), ("sha256_fq1", fq1), ("sha256_fq2", fq2)):
        assert hashlib.sha256(b).hexdigest() == str(g[key]), f"regenerated input differs from the one the reference ran on ({key})"
    d = tmp_path_factory.mktemp("cfg1")
    for name, b in (("db.fa", fa), ("r_1.fq", fq1), ("r_2.fq", fq2)):
        open(d / name, "wb").write(b)
    open(d / "coder.hdr", "wb").write(g["index_header"].tobytes())
    return g, d, refs, r1, r2


def lines_from_rows(rows):
    out = []
    for i, (n_int, el, ln, _) in enumerate(rows.tolist()):
        if el > 0 and np.float32(el) / np.float32(ln) > np.float32(0.75):
            out.append(orc.format_line(i + 1, n_int, el, ln))
    return b"".join(out)


def test_config1_cli_equals_reference(cfg1):
    g, d, *_ = cfg1
    args = [os.path.join(BIN, "eref"), str(d / "r_1.fq"), str(d / "r_2.fq"), str(d / "db.fa"), str(d / "tmp.txt")]
    # first run: no index beside the DB -> built with the coder the reference drew, then used
    p = subprocess.run(args + ["0.9", "0.85", "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, PALACE_CODER_HEADER=str(d / "coder.hdr")))
    assert p.returncode == 0, p.stderr
    assert p.stdout == g["stdout_090_085"].tobytes()
    h = hashlib.sha256()
    with open(str(d / "db.fa") + ".k32.index.dat", "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    assert h.hexdigest() == str(g["index_sha256"])                      # the 2.4 GB index file, byte for byte
    assert hashlib.sha256(open(str(d / "db.fa") + ".genome.len.txt", "rb").read()).hexdigest() == str(g["genome_len_sha256"])
    # second run: the index is found and only its header is read
    p = subprocess.run(args + ["0.8", "0.5", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    assert p.stdout == g["stdout_080_050"].tobytes()
    # third run: the reads counted in many small slabs (the executable's default is 2^28 positions; here 2^22 -> 12 passes)
    p = subprocess.run(args + ["0.9", "0.85", "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, PALACE_EREF_SLAB=str(1 << 22)))
    assert p.returncode == 0, p.stderr
    assert p.stdout == g["stdout_090_085"].tobytes()


def test_config1_c_abi_equals_reference(cfg1):
    g, d, refs, r1, r2 = cfg1
    ref_off = np.zeros(len(refs) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in refs], out=ref_off[1:])
    ref_bases = np.concatenate(refs)
    n, L = r1.shape
    off = np.arange(n + 1, dtype=np.int64) * L
    with capi.Ctx(0) as ctx:
        ctx.eref_set_coder(g["index_header"])
        ctx.eref_table_reset()
        for r in (r1, r2):
            db, do = ctx.upload(np.ascontiguousarray(r).reshape(-1)), ctx.upload(off)
            ctx.eref_count_reads(db, do, n)
            ctx.sync()
            db.free(); do.free()
        rb, ro = ctx.upload(ref_bases), ctx.upload(ref_off)
        ix = ctx.eref_probe_index_build(rb, ro, len(refs), len(ref_bases))
        for key, hr, pr in (("stdout_090_085", 0.9, 0.85), ("stdout_080_050", 0.8, 0.5)):
            one_min, three_min = capi.window_minimums(hr, pr)
            rows, rows_ix = ctx.empty((len(refs), 4), np.int32), ctx.empty((len(refs), 4), np.int32)
            ctx.eref_scan_refs(rb, ro, len(refs), len(ref_bases), one_min, three_min, rows)
            ctx.eref_scan_refs_indexed(ix, rb, ro, len(refs), len(ref_bases), one_min, three_min, rows_ix)
            a, b = rows.to_host(), rows_ix.to_host()
            assert np.array_equal(a, b)
            assert lines_from_rows(a) == g[key].tobytes()
        ctx.eref_probe_index_free(ix)


# ------------------------------------------------------------------------------------------------
# the headline configuration's size (1M contigs: 5 000 refs, 3 333 333 read pairs) against the COMPILED REFERENCE
# ------------------------------------------------------------------------------------------------
def test_headline_size_eref_cli_and_timed_path_equal_reference(tmp_path):
    """tests/golden/eref_1m.npz: stdout of the unmodified extract_ref.cpp (threads=1) on an eref input of the 1M-contig configuration's
    size, made in the build container by tests/golden/make_eref_1m_golden.py.  The inputs are regenerated from the seed and proven to be
    the same bytes; then (a) the executable on the files, building its index with the coder the reference drew, and (b) the path
    bench.py times -- reads packed in HBM, every look-up of Phase B inside the count launch (options final_count + probe_all_sets), the
    indexed scan on the hit bits -- must give the reference's lines byte for byte, for both ratio pairs (the second pair off the same
    hit bits).  extract_ref.cpp:504-617, 813-903, 905-1008."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "eref_1m.npz"))
    seed, n_refs, n_pairs, n_present, pool = (int(x) for x in g["params"])
    assert (n_refs, n_pairs) == (5000, 3_333_333)                       # 6.67 M reads x 150 bp: the 1M-contig configuration
    fa, fq1, fq2, refs, r1, r2 = synth.eref_config_inputs(seed, n_refs, n_pairs, pool_bases=pool, n_present=n_present, n_phage_pairs=n_pairs // 10, arrays=True)
    for key, b in (("sha256_db_fa", fa), ("sha256_fq1", fq1), ("sha256_fq2", fq2)):
        assert hashlib.sha256(b).hexdigest() == str(g[key]), f"regenerated input differs from the one the reference ran on ({key})"
    d = tmp_path
    for name, b in (("db.fa", fa), ("r_1.fq", fq1), ("r_2.fq", fq2)):
        open(d / name, "wb").write(b)
    del fa, fq1, fq2
    open(d / "coder.hdr", "wb").write(g["index_header"].tobytes())
    # (a) the executable
    args = [os.path.join(BIN, "eref"), str(d / "r_1.fq"), str(d / "r_2.fq"), str(d / "db.fa"), str(d / "tmp.txt")]
    p = subprocess.run(args + ["0.9", "0.85", "16"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, PALACE_CODER_HEADER=str(d / "coder.hdr")))
    assert p.returncode == 0, p.stderr
    assert p.stdout == g["stdout_090_085"].tobytes() and p.stdout.count(b"\n") >= 150
    p = subprocess.run(args + ["0.8", "0.5", "16"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    assert p.stdout == g["stdout_080_050"].tobytes()
    for name in ("r_1.fq", "r_2.fq", "db.fa.k32.index.dat"):
        os.remove(d / name)
    # (b) the timed path through the C ABI
    ref_off = np.zeros(len(refs) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in refs], out=ref_off[1:])
    ref_bases = np.concatenate(refs)
    n, rl = r1.shape
    r12 = np.concatenate([np.ascontiguousarray(r1).reshape(-1), np.ascontiguousarray(r2).reshape(-1)])
    del r1, r2
    off = np.arange(2 * n + 1, dtype=np.int64) * rl
    L = capi.lib()
    with capi.Ctx(0) as ctx:
        ctx.eref_set_coder(g["index_header"])
        db, do = ctx.upload(r12), ctx.upload(off)
        nb = int(L.palace_eref_packed_bytes(len(r12)))
        packed = [ctx.empty((nb,), np.uint8) for _ in range(3)]
        ctx.eref_pack_reads(db, do, 2 * n, None, len(r12), *packed)
        ctx.sync()
        db.free(); do.free()
        rb, ro = ctx.upload(ref_bases), ctx.upload(ref_off)
        ix = ctx.eref_probe_index_build(rb, ro, len(refs), len(ref_bases))
        ctx.eref_attach_probe_index(ix)
        ctx.eref_set_option("final_count", 1)
        ctx.eref_set_option("probe_all_sets", 1)
        ctx.eref_table_reset()
        ctx.eref_count_reads_packed(*packed, len(r12), 2 * n)
        rows = ctx.empty((len(refs), 4), np.int32)
        for key, hr, pr in (("stdout_090_085", 0.9, 0.85), ("stdout_080_050", 0.8, 0.5)):
            one_min, three_min = capi.window_minimums(hr, pr)
            ctx.eref_scan_refs_indexed(ix, rb, ro, len(refs), len(ref_bases), one_min, three_min, rows)
            assert lines_from_rows(rows.to_host()) == g[key].tobytes(), key
        ctx.eref_set_option("probe_all_sets", 0)
        ctx.eref_set_option("final_count", 0)
        ctx.eref_attach_probe_index(None)
        ctx.eref_probe_index_free(ix)
        ctx.eref_table_reset()


# ------------------------------------------------------------------------------------------------
# configs[3]: 5M-contig scale on one GPU
# ------------------------------------------------------------------------------------------------
def test_config3_5m_contig_scale_slab_path():
    import torch

    import bench
    from palace_amd import coder
    dev = torch.device("cuda", 0)
    sample = bench.make_sample(torch, dev, 5_000_000, 5000)             # 33.3 M reads x 150 bp = 5 Gbase
    torch.cuda.synchronize()
    n_side, P = sample["n_reads_side"], (lambda t: t.data_ptr())
    tot = n_side * bench.READ_LEN
    assert 2 * tot > (1 << 32)                                          # several slabs at either default slab size
    hdr = coder.header_from_picks(np.random.Generator(np.random.PCG64(1)).integers(0, 6, size=32))
    L = capi.lib()
    probe = np.unique(np.random.Generator(np.random.PCG64(2)).integers(0, 2**32, size=400000, dtype=np.uint64).astype(np.uint32))
    off_side = sample["read_off"][: n_side + 1].contiguous()
    with capi.Ctx(0) as ctx:
        ctx.eref_set_coder(hdr)

        def run(calls):
            ctx.eref_table_reset()
            for bases, offs, n, total in calls:
                capi._check(L.palace_eref_count_reads(ctx.h, P(bases), P(offs), n, None, total), "count")
            ctx.sync()
            return ctx.eref_table_popcounts(), ctx.eref_table_lookup(probe)

        both = run([(sample["r12"], sample["read_off"], 2 * n_side, 2 * tot)])        # one launch, default slabs
        assert both[0][0] > both[0][1] > both[0][2] > 0
        # order independence and additivity across calls: side 2 then side 1, each in its own slabs
        swapped = run([(sample["r2"], off_side, n_side, tot), (sample["r1"], off_side, n_side, tot)])
        assert both[0] == swapped[0] and np.array_equal(both[1], swapped[1])
        # a different slab size cuts the position range elsewhere: same table
        ctx.eref_set_option("slab_bases", 3 * (1 << 28))
        resliced = run([(sample["r12"], sample["read_off"], 2 * n_side, 2 * tot)])
        ctx.eref_set_option("slab_bases", 0)
        assert both[0] == resliced[0] and np.array_equal(both[1], resliced[1])
        # a slab-sized part (2^30 bases) through the partition kernels == the same part through direct global atomics
        n_part = (1 << 30) // bench.READ_LEN
        part = (sample["r12"], sample["read_off"], n_part, n_part * bench.READ_LEN)
        binned = run([part])
        ctx.eref_set_count_mode(1, 0)
        direct = run([part])
        ctx.eref_set_count_mode(0, 0)
        assert binned[0] == direct[0] and np.array_equal(binned[1], direct[1])
    del sample
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
# configs[4]: long contigs
# ------------------------------------------------------------------------------------------------
def classify_part(ctx, gs, a, b, ord_base, consumed, out, cap, prm):
    """palace_graph_classify on records [a, b) of the sample, candidates to `out` (a view of a candidate buffer); their number"""
    L, P = capi.lib(), (lambda t: t.data_ptr())
    sa_lo = int(gs["sa_off"][a].item())
    sa_local = (gs["sa_off"][a:b + 1] - sa_lo).contiguous()
    cols = capi.BamCols(b - a, *(P(gs["col"][k][a:b]) for k in ("tid", "pos", "mtid", "mpos", "nm", "ref_len", "read_len",
                                                             "clip_s", "clip_e", "flag", "mapq", "qkey")), P(sa_local))
    n_c = ctypes.c_int64()
    import torch
    torch.cuda.synchronize()                                  # sa_local was made on torch's stream
    capi._check(L.palace_graph_classify(ctx.h, ctypes.byref(cols), P(gs["sa"][sa_lo:]), len(gs["names"]), P(gs["tlen"]), P(gs["trank"]),
                                        P(gs["fastg"]), gs["n_fastg"], ctypes.byref(prm), ord_base, P(consumed),
                                        P(out), cap, ctypes.byref(n_c)), "classify")
    ctx.sync()                                                # (cols and sa_local stay alive until the call has finished)
    return n_c.value


def graph_on_gpu(ctx, gs, lo, hi, shards=1):
    """classify records [lo, hi) of the sample (in `shards` shards with their ordinal bases), resolve, copy numbers."""
    import torch
    L, P = capi.lib(), (lambda t: t.data_ptr())
    nt = len(gs["names"])
    dev = gs["tlen"].device
    consumed = torch.zeros(nt, dtype=torch.int64, device=dev)
    n = hi - lo
    cap = n + gs["n_sa"] + 1
    cands = torch.zeros((cap, 64), dtype=torch.uint8, device=dev)
    edges = torch.zeros((cap, 32), dtype=torch.uint8, device=dev)
    cn = torch.zeros(nt, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()              # torch zero-fills on ITS stream; the library's kernels run on the context's
    prm = capi.GraphParams.default()
    n_total = 0
    for s in range(shards):
        a, b = lo + n * s // shards, lo + n * (s + 1) // shards
        n_total += classify_part(ctx, gs, a, b, a - lo, consumed, cands[n_total:], cap - n_total, prm)
    h_cands = cands[:n_total].cpu().numpy().view(capi.CAND_DTYPE).reshape(-1).copy()
    n_e = ctypes.c_int64()
    capi._check(L.palace_graph_resolve(ctx.h, P(cands), n_total, n, ctypes.byref(prm), P(consumed), P(edges), cap, ctypes.byref(n_e)), "resolve")
    capi._check(L.palace_graph_copy_numbers(ctx.h, P(consumed), P(gs["tlen"]), nt, gs["avg_depth"], P(cn)), "cn")
    ctx.sync()
    e = edges[: n_e.value].cpu().numpy().view(capi.EDGE_DTYPE).reshape(-1)
    key = (e["left"].astype(np.int64) << 34) | (e["right"].astype(np.int64) << 2) | (e["oL"].astype(np.int64) << 1) | e["oR"]
    return consumed.cpu().numpy(), cn.cpu().numpy(), e[np.argsort(key)], h_cands


def graph_text(names, lens, cons, cn, edges):
    order = np.argsort(np.array(names, dtype="S"))
    rank = np.empty(len(names), dtype=np.int64)
    rank[order] = np.arange(len(names))
    out = ["SEG %s %s %d\n" % (names[i], "%g" % (cons[i] / max(1, lens[i])), cn[i]) for i in order]
    for left, right, counts, oL, oR, _ in sorted(edges.tolist(), key=lambda e: (rank[e[0]], rank[e[1]], e[3], e[4])):
        supp, supp_nf, span, span_nf = counts
        if supp + supp_nf + span + span_nf >= 5:
            out.append("JUNC %s %s %s %s %d %d\n" % (names[left], "+-"[oL], names[right], "+-"[oR], supp + span + supp_nf, span_nf))
    return "".join(out)


def test_config4_long_contigs_full_size_and_oracle_sample(tmp_path):
    import torch

    import bench
    from palace_amd.synth import BamRecord
    dev = torch.device("cuda", 0)
    n_contigs, n_pairs = 100_000, 3_333_333
    gs = bench.make_graph_sample(torch, dev, n_contigs, n_pairs, long_mode=True)
    torch.cuda.synchronize()
    names, lens = gs["names"], gs["lens"]
    assert (lens > 120_000).sum() > 1000 and np.sort(lens)[::-1].cumsum().searchsorted(lens.sum() / 2) < n_contigs // 3   # N50 far above the median
    with capi.Ctx(0) as ctx:
        # ---- full size (6.67 M records): one shot == four shards with ordinal bases ----
        cons1, cn1, e1, c1 = graph_on_gpu(ctx, gs, 0, gs["n"])
        cons4, cn4, e4, _ = graph_on_gpu(ctx, gs, 0, gs["n"], shards=4)
        assert np.array_equal(cons1, cons4) and np.array_equal(cn1, cn4)
        assert e1.tobytes() == e4.tobytes() and len(e1) > 1000
        gate = (c1["cls"] == 2).sum()                                   # decided by host libm: the exp-underflow zone
        assert gate > 1000, gate
        far = (c1["dL"].astype(np.int64) + c1["dR"]) > 120_000           # well inside the zone: exp() is 0, evidence rejected
        assert far.sum() > 1000
        assert int(cons1.sum()) >= int(gs["col"]["ref_len"].sum().item())   # depth: every primary record (+ the mate quirk)
        # ---- a 60k-record part, exactly, against the oracle ----
        m = 60_000
        lo = int(torch.nonzero(gs["col"]["tid"] == int(np.argmax(lens))).min().item())   # starts on the longest contig
        lo = max(0, min(lo, gs["n"] - m))
        cons, cn, edges, cs = graph_on_gpu(ctx, gs, lo, lo + m)
        assert (cs["cls"] == 2).sum() > 0
        # ---- candidates of ANOTHER call appended behind the last classify call's (what an in-place all-gather leaves on rank 0):
        # the legacy resolve entry must look at them instead of trusting the count it cached for that buffer ----
        L, P = capi.lib(), (lambda t: t.data_ptr())
        prm = capi.GraphParams.default()
        nt = len(names)
        cap = m + 4000 + gs["n_sa"] + 1
        buf = torch.zeros((cap, 64), dtype=torch.uint8, device=dev)
        other = torch.zeros((cap, 64), dtype=torch.uint8, device=dev)
        cons_ab = torch.zeros(nt, dtype=torch.int64, device=dev)
        edges_ab = torch.zeros((cap, 32), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        n_b = classify_part(ctx, gs, lo, lo + m, 2000, cons_ab, other, cap, prm)         # part B: holds gate candidates
        a0 = None
        for start in range(0, 40000, 2000):                                                # part A: 2000 records without any
            if start + 2000 > lo and start < lo + m:
                continue
            cons_try = torch.zeros(nt, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            n_a = classify_part(ctx, gs, start, start + 2000, 0, cons_try, buf, cap, prm)
            if n_a > 0 and (buf[:n_a].cpu().numpy().view(capi.CAND_DTYPE)["cls"] == 2).sum() == 0:
                a0 = start
                break
        assert a0 is not None
        cons_ab += cons_try
        buf[n_a:n_a + n_b] = other[:n_b]                                                   # "gathered in place" behind A's
        torch.cuda.synchronize()
        n_e = ctypes.c_int64()
        capi._check(L.palace_graph_resolve(ctx.h, P(buf), n_a + n_b, 2000 + m, ctypes.byref(prm), P(cons_ab), P(edges_ab), cap,
                                           ctypes.byref(n_e)), "resolve")
        ctx.sync()
        got_c = buf[:n_a + n_b].cpu().numpy().view(capi.CAND_DTYPE).reshape(-1)
        was_b = other[:n_b].cpu().numpy().view(capi.CAND_DTYPE).reshape(-1)
        assert ((was_b["cls"] == 2) & (was_b["found"] != 0)).sum() > 0
        assert ((got_c["cls"] == 2) & (got_c["found"] != 0)).sum() == 0, "gate candidates behind the cached count were not decided by the host"
    c = {k: v[lo:lo + m].cpu().numpy() for k, v in gs["col"].items()}
    so = gs["sa_off"][lo:lo + m + 1].cpu().numpy()
    sa = gs["sa"].cpu().numpy()
    recs = []
    for i in range(m):
        s_txt = None
        if so[i + 1] > so[i]:
            it = sa[so[i]]
            s_txt = f"{names[it[0]]},{it[1]},{'-' if it[7] else '+'},{it[4]}S{it[6] - it[4]}M,{it[2]},{it[3]};"
        cig = f"{c['ref_len'][i]}M{c['clip_e'][i]}S" if c["clip_e"][i] else "150M"
        recs.append(BamRecord(f"q{c['qkey'][i] & 0xffffffffffff:x}", int(c["flag"][i]) & 0xffff, int(c["tid"][i]), int(c["pos"][i]),
                              int(c["mapq"][i]), cig, int(c["mtid"][i]), int(c["mpos"][i]), nm=int(c["nm"][i]), sa=s_txt))
    fai = str(tmp_path / "g.fastg.fai")
    a, b, o1, o2 = gs["fastg_links"]
    touched = set(c["tid"].tolist()) | set(c["mtid"].tolist()) | {int(x) for x in sa[so[0]:so[-1], 0]}
    q = "'"
    with open(fai, "w") as f:
        for x, y, u, v in zip(a.tolist(), b.tolist(), o1.tolist(), o2.tolist()):
            if x in touched or y in touched:
                f.write(f"{names[x]}{q if u else ''}:{names[y]}{q if (u ^ v) else ''};\t{lens[x]}\t0\t60\t61\n")
    want = orc.graph_run(recs, list(zip(names, lens.tolist())), fai, gs["avg_depth"]).decode()
    assert graph_text(names, lens, cons, cn, edges) == want
    assert want.count("JUNC") > 0


def filtered_graph_files(tmp_path, gs, contig_of, cn, edges, edge_flags):
    """the filtered graph the device selected, as the text `matching` reads (SEG in filtered-graph id order, kept JUNCs), and contigs.paths"""
    import bench
    names = gs["names"]
    gpath, ppath = str(tmp_path / "filtered_graph.txt"), str(tmp_path / "contigs.paths")
    e = edges[(edge_flags & 6) != 0]
    with open(gpath, "w") as f:
        f.write("".join(f"SEG {names[c]} 1 {cn[c]} 0 0.000 0\n" for c in contig_of.tolist()))
        f.write("".join(f"JUNC {names[l]} {'+-'[a]} {names[r_]} {'+-'[b]} {x} 0\n"
                        for l, r_, a, b, x in zip(e["left"].tolist(), e["right"].tolist(), e["oL"].tolist(), e["oR"].tolist(),
                                                  e["counts"].astype(np.int64).sum(axis=1).tolist())))
    open(ppath, "w").write(bench.paths_text(names, gs["lens"], gs["side"]))
    return gpath, ppath


def test_config3_5m_contigs_graph_and_stage04_on_one_gpu(tmp_path):
    """configs[3] (5M contigs, 33.3 M primary records) on ONE GPU, generateGraph's kernels and stage 04 at full size:
      * classify + resolve + copy numbers over all records: one shot == four shards with ordinal bases; every record's reference
        span is in the depth sums;
      * a tenth of the records (3.3 M, from the middle of the sorted stream), exactly, against the oracle's `_graph.txt`;
      * the stage-04 selection and the decomposition of the WHOLE graph on the device; the decomposition against
        oracle/match_oracle.cpp on the filtered graph the device selected (2.2 M segments), with contigs.paths.
    (The selection itself is compared with the checker's chain at the 500k / 1M / long workloads' full size in
    tests/test_gpu_bench_workloads.py; the 8-GPU leg of this configuration needs the driver's node.)"""
    import torch

    import bench
    from bench import e2e
    from palace_amd import stage04_io
    dev = torch.device("cuda", 0)
    n_contigs, n_pairs = 5_000_000, 16_666_666
    gs = bench.make_graph_sample(torch, dev, n_contigs, n_pairs)
    gs["side"] = bench.make_side_inputs(gs)
    torch.cuda.synchronize()
    names, lens = gs["names"], gs["lens"]
    assert gs["n"] == 2 * n_pairs
    L, P = capi.lib(), (lambda t: t.data_ptr())
    with capi.Ctx(0) as ctx:
        cons1, cn1, e1, c1 = graph_on_gpu(ctx, gs, 0, gs["n"])
        cons4, cn4, e4, _ = graph_on_gpu(ctx, gs, 0, gs["n"], shards=4)
        assert np.array_equal(cons1, cons4) and np.array_equal(cn1, cn4) and e1.tobytes() == e4.tobytes()
        assert len(e1) > 100_000 and len(c1) > 1_000_000
        assert int(cons1.sum()) >= int(gs["col"]["ref_len"].sum().item())
        del c1, cons4, cn4, e4
        # ---- a tenth of the records against the oracle ----
        m = gs["n"] // 10
        lo = gs["n"] // 2 - m // 2
        cons, cn, edges, _ = graph_on_gpu(ctx, gs, lo, lo + m)
        col = {k: v[lo:lo + m].cpu().numpy() for k, v in gs["col"].items()}
        so = gs["sa_off"][lo:lo + m + 1].cpu().numpy().astype(np.int64)
        sa = gs["sa"][int(so[0]):max(int(so[-1]), int(so[0]) + 1)].cpu().numpy()
        gin = orc.GraphInput.from_columns(col, so - so[0], sa, names, lens)
        touched = np.zeros(n_contigs, dtype=bool)
        touched[col["tid"]] = True
        touched[col["mtid"][col["mtid"] >= 0]] = True
        if so[-1] > so[0]:
            touched[sa[: int(so[-1] - so[0]), 0]] = True
        a, b, o1, o2 = gs["fastg_links"]
        keep = touched[a] | touched[b]
        q = "'"
        fai = str(tmp_path / "g.fastg.fai")
        with open(fai, "w") as f:
            f.write("".join(f"{names[x]}{q if u else ''}:{names[y]}{q if (u ^ v) else ''};\t{lens[x]}\t0\t60\t61\n"
                            for x, y, u, v in zip(a[keep].tolist(), b[keep].tolist(), o1[keep].tolist(), o2[keep].tolist())))
        want = gin.run(fai, gs["avg_depth"])
        del gin, col
        got = e2e.graph_text(names, lens, cons, cn, edges, rank=gs["trank"].cpu().numpy())
        assert got.encode() == want
        assert want.count(b"\nJUNC ") > 1000
        del got, want
        # ---- stage 04 over the whole graph, on the device ----
        side = gs["side"]
        n_e = len(e1)
        st = capi.Stage04(ctx, side["seed"], lens.astype(np.int32), gs["trank"].cpu().numpy(), lens.astype(np.int32), side["path_off"], side["path_tok"], 5)
        d_e = ctx.upload(np.ascontiguousarray(e1).view(np.uint8).reshape(-1))
        d_n = ctx.upload(np.array([n_e], np.int64))
        d_cn = ctx.upload(cn1)
        st.filter(d_e.ptr, d_n.ptr, n_e)
        st.match(d_e.ptr, d_cn.ptr, 10, False, True)
        res, contig_of = st.result()
        got_lin, got_cyc = stage04_io.matching_text(res, contig_of, names, self_loops=True, break_cycles=False)
        contig_of = np.asarray(contig_of).copy()
        seg_flags, edge_flags = st.flags(n_e)
        counts = st.counts()
        st.close()
    assert counts["segs_filtered"] == len(contig_of) > 1_000_000 and counts["juncs"] > 100_000
    assert counts["kept_pass2"] + counts["kept_pass3_more"] == int(((edge_flags & 6) != 0).sum()) > 50_000
    gpath, ppath = filtered_graph_files(tmp_path, gs, contig_of, cn1, e1, edge_flags)
    lin, cyc = orc.match_run(gpath, ppath, 10, self_loops=True, cap=160 * len(contig_of) + (1 << 20))
    assert got_lin.encode() == lin and got_cyc.encode() == cyc
    assert lin.count(b"\t") > 100_000


# ------------------------------------------------------------------------------------------------
# E3 at its real threshold: more than 1 Gbase in fq1 switches the reference's read subsampling on
# ------------------------------------------------------------------------------------------------
def test_e3_subsampling_at_the_real_threshold_equals_reference(tmp_path):
    """8.4 M read pairs (1.26 Gbase per side): cal_sam_ratio gives 79, so one glibc rand() % 100 draw per sequence line
    (seed 1, fq1 then fq2) decides which reads are counted (extract_ref.cpp:955-960, 1124-1148, 1239-1240).  Golden = stdout
    of the compiled reference on the same bytes (tests/golden/make_eref_e3_golden.py, ~50 min of CPU in the build container);
    100 of the 200 refs are present at 3-10x, so the dropped fifth of the reads changes the answer."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "eref_e3.npz"))
    seed, n_refs, n_pairs, n_present, n_phage = (int(x) for x in g["params"])
    fa, fq1, fq2 = synth.eref_config_inputs(seed, n_refs, n_pairs, n_present=n_present, n_phage_pairs=n_phage)
    for key, b in (("sha256_db_fa", fa), ("sha256_fq1", fq1), ("sha256_fq2", fq2)):
        assert hashlib.sha256(b).hexdigest() == str(g[key]), f"regenerated input differs from the one the reference ran on ({key})"
    for name, b in (("db.fa", fa), ("r_1.fq", fq1), ("r_2.fq", fq2)):
        open(tmp_path / name, "wb").write(b)
    del fa, fq1, fq2
    open(tmp_path / "coder.hdr", "wb").write(g["index_header"].tobytes())
    args = [os.path.join(BIN, "eref"), str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq"), str(tmp_path / "db.fa"), str(tmp_path / "tmp.txt")]
    p = subprocess.run(args + ["0.9", "0.85", "16"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, PALACE_CODER_HEADER=str(tmp_path / "coder.hdr")))
    assert p.returncode == 0, p.stderr
    assert p.stdout == g["stdout_090_085"].tobytes()
    p = subprocess.run(args + ["0.8", "0.5", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    assert p.stdout == g["stdout_080_050"].tobytes()
    assert g["stdout_090_085"].tobytes().count(b"\n") < g["stdout_080_050"].tobytes().count(b"\n") < n_present
