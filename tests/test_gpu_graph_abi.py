"""generateGraph through the C ABI directly (palace_graph_classify / _resolve / _copy_numbers) on the
structure-of-arrays sample bench.py generates, against the oracle run on the same records rebuilt as
BAM-level records.  Checks the bench's own data path, not only the CLI's."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bench_shaped_sample_equals_oracle(tmp_path):
    import torch

    import bench
    from oracle import binding as orc
    from palace_amd import capi
    from palace_amd.synth import BamRecord

    dev = torch.device("cuda", 0)
    n_contigs, n_pairs = 20000, 66666
    gs = bench.make_graph_sample(torch, dev, n_contigs, n_pairs)
    torch.cuda.synchronize()
    names, lens = gs["names"], gs["lens"]
    P = lambda t: t.data_ptr()
    L = capi.lib()
    with capi.Ctx(0) as ctx:
        cols = capi.BamCols(gs["n"], *(P(gs["col"][k]) for k in ("tid", "pos", "mtid", "mpos", "nm", "ref_len", "read_len",
                                                               "clip_s", "clip_e", "flag", "mapq", "qkey")), P(gs["sa_off"]))
        prm = capi.GraphParams.default()
        consumed = torch.zeros(n_contigs, dtype=torch.int64, device=dev)
        cap = gs["n"] + gs["n_sa"] + 1
        cands = torch.zeros((cap, 64), dtype=torch.uint8, device=dev)
        edges = torch.zeros((cap, 32), dtype=torch.uint8, device=dev)
        cn = torch.zeros(n_contigs, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()          # torch zero-fills on ITS stream; the library's kernels run on the context's
        n_c, n_e = ctypes.c_int64(), ctypes.c_int64()
        capi._check(L.palace_graph_classify(ctx.h, ctypes.byref(cols), P(gs["sa"]), n_contigs, P(gs["tlen"]), P(gs["trank"]),
                                            P(gs["fastg"]), gs["n_fastg"], ctypes.byref(prm), 0, P(consumed), P(cands), cap,
                                            ctypes.byref(n_c)), "classify")
        capi._check(L.palace_graph_resolve(ctx.h, P(cands), n_c.value, gs["n_total"], ctypes.byref(prm), P(consumed), P(edges),
                                           cap, ctypes.byref(n_e)), "resolve")
        capi._check(L.palace_graph_copy_numbers(ctx.h, P(consumed), P(gs["tlen"]), n_contigs, gs["avg_depth"], P(cn)), "cn")
        ctx.sync()
        # the same classification with the FASTG look-ups narrowed by per-contig offsets (palace_graph_classify_ix): the same
        # candidates (as a set: the append order is the waves'), the same depth sums
        first = torch.zeros(n_contigs + 1, dtype=torch.int32, device=dev)
        cands_ix, cons_ix = torch.zeros_like(cands), torch.zeros_like(consumed)
        cands_plain, cons_plain = torch.zeros_like(cands), torch.zeros_like(consumed)
        torch.cuda.synchronize()
        capi._check(L.palace_graph_fastg_offsets(ctx.h, P(gs["fastg"]), gs["n_fastg"], n_contigs, P(first)), "offsets")
        n_ix, n_pl, nb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        capi._check(L.palace_graph_classify_ix(ctx.h, ctypes.byref(cols), P(gs["sa"]), n_contigs, P(gs["tlen"]), P(gs["trank"]), P(gs["fastg"]),
                                               gs["n_fastg"], P(first), ctypes.byref(prm), 0, P(cons_ix), P(cands_ix), cap, ctypes.byref(n_ix),
                                               ctypes.byref(nb)), "classify_ix")
        capi._check(L.palace_graph_classify_ex(ctx.h, ctypes.byref(cols), P(gs["sa"]), n_contigs, P(gs["tlen"]), P(gs["trank"]), P(gs["fastg"]),
                                               gs["n_fastg"], ctypes.byref(prm), 0, P(cons_plain), P(cands_plain), cap, ctypes.byref(n_pl),
                                               ctypes.byref(nb)), "classify_ex")
        ctx.sync()
        assert n_ix.value == n_pl.value == n_c.value > 1000
        def as_set(t, n):                                    # the 64-byte records in byte order
            a = t[:n].cpu().numpy().reshape(-1, 64)
            return a[np.lexsort(a.T[::-1])].tobytes()
        assert as_set(cands_ix, n_ix.value) == as_set(cands_plain, n_pl.value)
        fo = first.cpu().numpy().astype(np.int64)
        fk_hi = (gs["fastg"].cpu().numpy().view(np.uint64) >> np.uint64(33)).astype(np.int64)
        assert fo[0] == 0 and fo[-1] == gs["n_fastg"] and np.array_equal(fo, np.searchsorted(fk_hi, np.arange(n_contigs + 1), side="left"))
        assert torch.equal(cons_ix, cons_plain)
        h_cons = consumed.cpu().numpy()
        h_cn = cn.cpu().numpy()
        h_edges = edges[: n_e.value].cpu().numpy().view(capi.EDGE_DTYPE).reshape(-1)
    # ---- GPU numbers -> the text generateGraph writes (generate_graph.cpp:1048-1076) ----
    order = np.argsort(np.array(names, dtype="S"))
    rank = np.empty(n_contigs, dtype=np.int64)
    rank[order] = np.arange(n_contigs)
    got = []
    for i in order:
        got.append("SEG %s %s %d\n" % (names[i], "%g" % (h_cons[i] / max(1, lens[i])), h_cn[i]))
    es = sorted(h_edges.tolist(), key=lambda e: (rank[e[0]], rank[e[1]], e[3], e[4]))
    for left, right, counts, oL, oR, _ in es:
        supp, supp_nf, span, span_nf = counts
        if supp + supp_nf + span + span_nf >= 5:
            got.append("JUNC %s %s %s %s %d %d\n" % (names[left], "+-"[oL], names[right], "+-"[oR], supp + span + supp_nf, span_nf))
    # ---- the same records for the oracle ----
    c = {k: v.cpu().numpy() for k, v in gs["col"].items()}
    so = gs["sa_off"].cpu().numpy()
    sa = gs["sa"].cpu().numpy()
    recs = []
    for i in range(gs["n"]):
        s_txt = None
        if so[i + 1] > so[i]:
            it = sa[so[i]]
            s_txt = f"{names[it[0]]},{it[1]},{'-' if it[7] else '+'},{it[4]}S{it[6] - it[4]}M,{it[2]},{it[3]};"
        cig = f"{c['ref_len'][i]}M{c['clip_e'][i]}S" if c["clip_e"][i] else "150M"
        recs.append(BamRecord(f"q{c['qkey'][i] & 0xffffffffffff:x}", int(c["flag"][i]) & 0xffff, int(c["tid"][i]), int(c["pos"][i]),
                              int(c["mapq"][i]), cig, int(c["mtid"][i]), int(c["mpos"][i]), nm=int(c["nm"][i]), sa=s_txt))
    fk = gs["fastg"].cpu().numpy().view(np.uint64)
    fai = str(tmp_path / "g.fastg.fai")
    with open(fai, "w") as f:
        for k in fk.tolist():
            a, b, o1, o2 = k >> 33, (k >> 2) & 0x7fffffff, (k >> 1) & 1, k & 1
            f.write(f"{names[a]}{chr(39) if o1 else ''}:{names[b]}{chr(39) if (o1 ^ o2) else ''};\t{lens[a]}\t0\t60\t61\n")
    want = orc.graph_run(recs, list(zip(names, lens.tolist())), fai, gs["avg_depth"]).decode()
    assert "".join(got) == want
    assert want.count("JUNC") > 50


def _unique_edges(e):
    """canonical, unique edge keys as generateGraph would hand them over"""
    key = (e["left"].astype(np.int64) << 34) | (e["right"].astype(np.int64) << 2) | (e["oL"] << 1) | e["oR"]
    return e[np.unique(key, return_index=True)[1]]


def _decompose_as_text(ctx, names, cn, e, aggressive):
    """bench.py's glue (graph_to_arcs) + palace_match_decompose, formatted the way matching_main.cpp formats"""
    import bench
    from palace_amd import capi
    copies, src, dst, w = bench.graph_to_arcs(cn, len(names), e)
    off, verts, kind, it, open_at = capi.match_decompose(ctx, copies, src, dst, 10, aggressive)
    tok = lambda v: names[v >> 1] + "+-"[v & 1]
    lin, cyc, seen_l, seen_c = [], [], set(), set()
    for c in range(len(kind)):
        vs = verts[off[c]:off[c + 1]]
        body = "\t".join(tok(int(v)) for v in vs) + "\n"
        if not kind[c]:
            if len(vs) == 1 and it[c] > 0:
                continue
            if body not in seen_l:
                seen_l.add(body); lin.append(body)
        elif body not in seen_c:
            seen_c.add(body); cyc.append(f"iter {it[c]}\n" + body)
    return "".join(lin).encode(), "".join(cyc).encode()


def _oracle_text(tmp_path, names, cn, e, aggressive):
    from oracle import binding as orc
    g = str(tmp_path / "g.txt")
    tot = e["counts"].sum(axis=1)
    with open(g, "w") as f:
        f.write("".join(f"SEG {nm} 1 {k} 0 0.000 0\n" for nm, k in zip(names, cn.tolist())))
        f.write("".join(f"JUNC {names[l]} {'+-'[a]} {names[r]} {'+-'[b]} {t} 0\n"
                        for l, r, a, b, t in zip(e["left"].tolist(), e["right"].tolist(), e["oL"].tolist(), e["oR"].tolist(), tot.tolist())
                        if t >= 5))
    return orc.match_run(g, None, 10, aggressive=aggressive, cap=256 * 1024 * 1024)


@pytest.mark.parametrize("n,aggressive", [(30000, False), (300000, False), (300000, True)])
def test_bench_matching_stage_equals_oracle(tmp_path, n, aggressive):
    """bench.py's glue (graph_to_arcs) + palace_match_decompose equal the oracle's linear/cycle files for the same SEG/JUNC
    text.  (n = 300000: above 2^17 segments the bare segments are merged back by several threads.)"""
    from palace_amd import capi

    rng = np.random.Generator(np.random.PCG64(4))
    names = [f"EDGE_{i + 1}_length_{int(rng.integers(60, 5000))}_cov_{rng.random() * 20:.4f}" for i in range(n)]
    cn = rng.integers(0, 4, size=n).astype(np.int32)
    e = np.zeros(40000, dtype=capi.EDGE_DTYPE)
    e["left"] = rng.integers(0, n, len(e)); e["right"] = rng.integers(0, n, len(e))
    e["oL"] = rng.integers(0, 2, len(e)); e["oR"] = rng.integers(0, 2, len(e))
    e["counts"] = rng.integers(0, 6, size=(len(e), 4))
    ring = np.arange(3000)                                    # planted heavy 10-rings so that cycles are certain
    e["left"][:3000] = ring; e["right"][:3000] = (ring // 10) * 10 + (ring + 1) % 10
    e["oL"][:3000] = 0; e["oR"][:3000] = 0; e["counts"][:3000] = 50
    e = _unique_edges(e)
    with capi.Ctx(0) as ctx:
        lin, cyc = _decompose_as_text(ctx, names, cn, e, aggressive)
    want_lin, want_cyc = _oracle_text(tmp_path, names, cn, e, aggressive)
    assert lin == want_lin
    assert cyc == want_cyc
    assert want_cyc.count(b"iter") > 0


@pytest.mark.parametrize("aggressive", [False, True])
def test_decomposition_with_the_host_checking_fixed_points(tmp_path, aggressive):
    """The decomposition enqueues a fixed number of matching iterations per round and only looks at the end; a round that
    needed more is redone with the host watching every fixed point.  Forced here two ways: a zig-zag of ascending weights
    (a_i -> b_i < a_(i+1) -> b_i < a_(i+1) -> b_(i+1) ...: one arc per iteration can be taken, hundreds of iterations), and
    iters_per_round = 1 on a random graph.  Both must equal the oracle and the default setting."""
    from palace_amd import capi
    rng = np.random.Generator(np.random.PCG64(21))
    k = 400
    n = 2 * k + 2000
    names = [f"EDGE_{i + 1}_length_{int(rng.integers(60, 5000))}_cov_{rng.random() * 20:.4f}" for i in range(n)]
    cn = rng.integers(0, 4, size=n).astype(np.int32)
    zig = np.zeros(2 * k - 1, dtype=capi.EDGE_DTYPE)          # a_i = segment i, b_i = segment k + i
    for j in range(2 * k - 1):
        i = j // 2
        zig[j]["left"] = i + (j & 1); zig[j]["right"] = k + i
        zig[j]["counts"] = (5 + j, 0, 0, 0)
    rnd = np.zeros(3000, dtype=capi.EDGE_DTYPE)
    rnd["left"] = rng.integers(2 * k, n, len(rnd)); rnd["right"] = rng.integers(2 * k, n, len(rnd))
    rnd["oL"] = rng.integers(0, 2, len(rnd)); rnd["oR"] = rng.integers(0, 2, len(rnd))
    rnd["counts"] = rng.integers(0, 6, size=(len(rnd), 4))
    rnd = _unique_edges(rnd)
    e = np.concatenate([zig, rnd])
    want = _oracle_text(tmp_path, names, cn, e, aggressive)
    with capi.Ctx(0) as ctx:
        assert _decompose_as_text(ctx, names, cn, e, aggressive) == want           # default: 7 iterations do not settle the zig-zag
        ctx.match_set_option("iters_per_round", 1)
        assert _decompose_as_text(ctx, names, cn, rnd, aggressive) == _oracle_text(tmp_path, names, cn, rnd, aggressive)
        ctx.match_set_option("iters_per_round", 64)
        assert _decompose_as_text(ctx, names, cn, e, aggressive) == want


def test_compact_decomposition_equals_full():
    """palace_match_decompose_ex(compact=1): the listed components + the bare-segment bits are exactly the full listing."""
    from palace_amd import capi
    rng = np.random.Generator(np.random.PCG64(8))
    n = 50_000
    cn = rng.integers(0, 4, size=n).astype(np.int32)
    e = np.zeros(6000, dtype=capi.EDGE_DTYPE)
    e["left"] = rng.integers(0, n, len(e)); e["right"] = rng.integers(0, n, len(e))
    e["oL"] = rng.integers(0, 2, len(e)); e["oR"] = rng.integers(0, 2, len(e))
    e["counts"] = rng.integers(0, 6, size=(len(e), 4))
    e = _unique_edges(e)
    import bench
    copies, src, dst, w = bench.graph_to_arcs(cn, n, e)
    for aggressive in (False, True):
        with capi.Ctx(0) as ctx:
            off, verts, kind, it, open_at = capi.match_decompose(ctx, copies, src, dst, 10, aggressive)
            with capi.match_decompose_views(ctx, copies, src, dst, 10, aggressive, compact=True) as r:
                c_off, c_verts, c_kind, c_it, c_open, bare, n_bare = (r.off.copy(), r.verts.copy(), r.kind.copy(), r.iter.copy(),
                                                                    r.open_at.copy(), r.bare.copy(), r.n_bare)
        is_bare = np.unpackbits(bare.view(np.uint8), bitorder="little")[:n].astype(bool)
        assert is_bare.sum() == n_bare
        touched = np.zeros(n, dtype=bool); touched[src >> 1] = True; touched[dst >> 1] = True
        assert np.array_equal(is_bare, ~touched)
        # full listing = compact components + bare singletons (round 0 and, when aggressive, the extra round), first-vertex order per round
        first = verts[off[:-1]]
        single_bare = ((off[1:] - off[:-1]) == 1) & is_bare[first >> 1] & (kind == 0)
        keep = ~single_bare
        assert np.array_equal(kind[keep], c_kind) and np.array_equal(it[keep], c_it) and np.array_equal(open_at[keep], c_open)
        lens = (off[1:] - off[:-1])[keep]
        assert np.array_equal(np.cumsum(lens), c_off[1:])
        vmask = np.repeat(keep, off[1:] - off[:-1])
        assert np.array_equal(verts[vmask], c_verts)
        assert single_bare.sum() == n_bare * (2 if aggressive else 1)
