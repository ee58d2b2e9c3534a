"""GPU parity of the eref HIP path (through the C ABI) against the oracle and the golden vectors
the compiled reference produced.  Bit-exact: this is integer/index work."""
import hashlib

import numpy as np
import pytest

from oracle import binding as orc
from palace_amd import capi, synth

pytestmark = pytest.mark.gpu


def parse_fasta(buf: bytes):
    """-> list of sequences (np.uint8) in file order (test helper; '\n'-stripped concatenation)."""
    seqs, cur = [], None
    for line in buf.split(b"\n"):
        if line.startswith(b">"):
            if cur is not None:
                seqs.append(np.frombuffer(b"".join(cur), dtype=np.uint8))
            cur = []
        elif cur is not None:
            cur.append(line)
    if cur is not None:
        seqs.append(np.frombuffer(b"".join(cur), dtype=np.uint8))
    return seqs


def oracle_key_counts(bases, offsets, cc):
    """unique canonical indices of a read set with min(3, multiplicity), via the oracle."""
    keys = []
    for r in range(len(offsets) - 1):
        s = bases[offsets[r]:offsets[r + 1]]
        if len(s) >= 32:
            k = orc.index_ref(s, cc)
            keys.append(k[k != 0])
    if not keys:
        return np.zeros(0, np.uint32), np.zeros(0, np.int64)
    u, c = np.unique(np.concatenate(keys), return_counts=True)
    return u, np.minimum(c, 3)


@pytest.fixture(scope="module")
def ctx():
    c = capi.Ctx(0)
    yield c
    c.close()


def count_on_gpu(ctx, readsets, header, keep=None):
    ctx.eref_set_coder(header)
    ctx.eref_table_reset()
    for i, (b, o) in enumerate(readsets):
        db, do = ctx.upload(b), ctx.upload(np.asarray(o, dtype=np.int64))
        dk = ctx.upload(keep[i]) if keep is not None else None
        ctx.eref_count_reads(db, do, len(o) - 1, dk)
        ctx.sync()
        db.free(); do.free()
        if dk: dk.free()


def assert_table_equals(ctx, u, c):
    got = ctx.eref_table_lookup(u) if len(u) else np.zeros(0, np.uint8)
    assert np.array_equal(got, c.astype(np.uint8))
    pops = ctx.eref_table_popcounts()
    assert pops == [int((c >= 1).sum()), int((c >= 2).sum()), int((c >= 3).sum())]   # nothing else is set


def test_count_table_golden_reads(ctx, golden_eref):
    g = golden_eref
    cc = orc.header_to_cc(g["index_header"])
    sets = [(g["r1_bases"], g["r1_offsets"]), (g["r2_bases"], g["r2_offsets"])]
    count_on_gpu(ctx, sets, g["index_header"])
    u, c = oracle_key_counts(np.concatenate([g["r1_bases"], g["r2_bases"]]),
                             np.concatenate([g["r1_offsets"], g["r2_offsets"][1:] + g["r1_offsets"][-1]]), cc)
    assert_table_equals(ctx, u, c)
    # and against the reference-shaped byte table of the oracle
    t = orc.CountTable()
    for b, o in sets:
        t.count(b, o, cc)
    assert np.array_equal(t.lookup(u), c.astype(np.uint8))
    t.free()


@pytest.mark.parametrize("key,hr,pr", [("stdout_090_085", 0.9, 0.85), ("stdout_080_050", 0.8, 0.5),
                                       ("stdout_095_090", 0.95, 0.9)])
def test_stdout_equals_reference(ctx, golden_eref, key, hr, pr):
    g = golden_eref
    count_on_gpu(ctx, [(g["r1_bases"], g["r1_offsets"]), (g["r2_bases"], g["r2_offsets"])], g["index_header"])
    refs = [s for s in parse_fasta(g["db_fasta"].tobytes()) if len(s) > 32]     # extract_ref.cpp:697
    rs = synth.reads_from_list(refs)
    db, do = ctx.upload(rs.bases), ctx.upload(rs.offsets)
    rows = ctx.empty((rs.n, 4), np.int32)
    one_min, three_min = capi.window_minimums(hr, pr)
    ctx.eref_scan_refs(db, do, rs.n, len(rs.bases), one_min, three_min, rows)
    r = rows.to_host()
    ix = ctx.eref_probe_index_build(db, do, rs.n, len(rs.bases))              # same rows through the per-DB probe index
    rows2 = ctx.empty((rs.n, 4), np.int32)
    ctx.eref_scan_refs_indexed(ix, db, do, rs.n, len(rs.bases), one_min, three_min, rows2)
    assert np.array_equal(rows2.to_host(), r)
    other = orc.header_from_picks((np.arange(32) * 5 + 1) % 6)                 # an index is tied to the coder it was built with
    if not np.array_equal(other, g["index_header"]):
        ctx.eref_set_coder(other)
        with pytest.raises(capi.PalaceError, match="another coder"):
            ctx.eref_scan_refs_indexed(ix, db, do, rs.n, len(rs.bases), one_min, three_min, rows2)
        ctx.eref_set_coder(g["index_header"])
    ctx.eref_probe_index_free(ix)
    rows2.free()
    out = b""
    for i in range(rs.n):
        n_int, el, L, _ = (int(v) for v in r[i])
        assert L == len(refs[i])
        if el > 0 and np.float32(el) / np.float32(L) > 0.75:
            out += orc.format_line(i + 1, n_int, el, L)
    assert out == g[key].tobytes()
    for b in (db, do, rows):
        b.free()


def test_index_build_equals_reference(ctx, golden_eref):
    g = golden_eref
    ctx.eref_set_coder(g["index_header"])
    refs = [s for s in parse_fasta(g["db_fasta"].tobytes()) if len(s) > 32]
    rs = synth.reads_from_list(refs)
    npos = np.array([len(s) - 31 for s in refs], dtype=np.int64)
    out_off = np.zeros(len(refs) + 1, dtype=np.int64)
    np.cumsum(3 * npos, out=out_off[1:])
    db, do, doo = ctx.upload(rs.bases), ctx.upload(rs.offsets), ctx.upload(out_off)
    out = ctx.empty(int(out_off[-1]), np.uint32)
    ctx.eref_index_refs(db, do, rs.n, out, doo)
    idx = out.to_host()
    body = b"".join(np.uint32(len(s)).tobytes() + idx[out_off[i]:out_off[i + 1]].tobytes()
                    for i, s in enumerate(refs))
    assert hashlib.sha256(body).digest() == g["index_body_sha256"].tobytes()
    for b in (db, do, doo, out):
        b.free()


def test_edge_cases(ctx):
    rng = synth.rng_for(7)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    base = synth.random_dna(rng, 5000)
    reads = [base[:0], base[:1], base[:31], base[:32], base[10:43], base[100:1100].copy(), base[200:264],
             base[300:395], base[300:396], base[300:397], np.frombuffer(b"A" * 200, dtype=np.uint8),
             np.frombuffer(base[500:700].tobytes().lower(), dtype=np.uint8)]
    withn = base[1000:1300].copy(); withn[[0, 31, 32, 150, 299]] = ord("N"); reads.append(withn)
    weird = base[2000:2100].copy(); weird[50] = 0xC3; weird[51] = ord("\r"); reads.append(weird)
    rs = synth.reads_from_list(reads)
    count_on_gpu(ctx, [(rs.bases, rs.offsets)], hdr)
    u, c = oracle_key_counts(rs.bases, rs.offsets, cc)
    assert_table_equals(ctx, u, c)
    # empty read set is a no-op
    ctx.eref_table_reset()
    ctx.eref_count_reads(ctx.upload(np.zeros(1, np.uint8)), ctx.upload(np.zeros(1, np.int64)), 0)
    assert ctx.eref_table_popcounts() == [0, 0, 0]


def test_keep_mask(ctx):
    rng = synth.rng_for(11)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    rs = synth.vector_reads(rng, synth.random_dna(rng, 50000), 3000, 100)
    keep = (rng.random(rs.n) < 0.5).astype(np.uint8)
    count_on_gpu(ctx, [(rs.bases, rs.offsets)], hdr, keep=[keep])
    kept = [rs.read(i) for i in range(rs.n) if keep[i]]
    ks = synth.reads_from_list(kept)
    u, c = oracle_key_counts(ks.bases, ks.offsets, cc)
    assert_table_equals(ctx, u, c)


def test_random_reads_medium(ctx):
    """60k x 150 bp with heavy duplication (small pool) so counts 1, 2 and >=3 all occur."""
    rng = synth.rng_for(3)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    rs = synth.vector_reads(rng, synth.random_dna(rng, 3_000_000), 60000, 150)
    count_on_gpu(ctx, [(rs.bases, rs.offsets)], hdr)
    t = orc.CountTable()
    t.count(rs.bases, rs.offsets, cc)
    u, c = oracle_key_counts(rs.bases, rs.offsets, cc)
    assert np.array_equal(t.lookup(u), c.astype(np.uint8))
    assert_table_equals(ctx, u, c)
    assert (c == 1).any() and (c == 2).any() and (c == 3).any()
    t.free()


def test_scan_matches_oracle_on_random_coverage(ctx):
    """Refs with patchy coverage, several ratio pairs: rows equal the oracle's (n_intervals, el)."""
    rng = synth.rng_for(5)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    refs, reads = [], []
    for L in (500, 501, 999, 1000, 2048, 4096, 4097, 12345, 33, 64, 65, 40000):
        s = synth.random_dna(rng, L)
        refs.append(s)
        n_seg = int(rng.integers(0, 4))
        for _ in range(n_seg):
            a = int(rng.integers(0, max(1, L - 200)))
            b = int(min(L, a + rng.integers(200, 9000)))
            if b - a < 120:
                continue
            for _ in range(int(6 * (b - a) / 100)):
                st = int(rng.integers(a, b - 100))
                reads.append(synth.mutate(rng, s[st:st + 100], 0.01))
    rr = synth.reads_from_list(reads)
    count_on_gpu(ctx, [(rr.bases, rr.offsets)], hdr)
    t = orc.CountTable()
    t.count(rr.bases, rr.offsets, cc)
    rs = synth.reads_from_list(refs)
    db, do = ctx.upload(rs.bases), ctx.upload(rs.offsets)
    rows = ctx.empty((rs.n, 4), np.int32)
    rows_ix = ctx.empty((rs.n, 4), np.int32)
    ix = ctx.eref_probe_index_build(db, do, rs.n, len(rs.bases))
    for hr, pr in ((0.9, 0.85), (0.5, 0.2), (0.99, 0.97), (0.7, 0.7)):
        one_min, three_min = capi.window_minimums(hr, pr)
        ctx.eref_scan_refs(db, do, rs.n, len(rs.bases), one_min, three_min, rows)
        got = rows.to_host()
        ctx.eref_scan_refs_indexed(ix, db, do, rs.n, len(rs.bases), one_min, three_min, rows_ix)
        assert np.array_equal(rows_ix.to_host(), got), (hr, pr)
        for i, s in enumerate(refs):
            _, n_int, el, _ = orc.scan_ref(orc.index_ref(s, cc), len(s), t, hr, pr)
            assert (int(got[i, 0]), int(got[i, 1]), int(got[i, 2])) == (n_int, el, len(s)), (i, len(s), hr, pr)
    ctx.eref_probe_index_free(ix)
    t.free()


@pytest.mark.parametrize("cap", [0, 64])                                       # 64: most keys take the overflow path (touched buckets are seeded)
def test_probe_fused_into_the_final_count_equals_the_probe_kernel_and_the_oracle(ctx, cap):
    """palace_eref_attach_probe_index: a final count (option final_count, binned path, one slab, whole key space) leaves the
    channel-0 hits of the attached DB, and the indexed scan that follows starts from them.  Rows must equal the scan that
    probes for itself and the oracle's; any other count in between (not final, or a share of the key space) must make the
    scan probe for itself again; a second sample through the same attached index must not see the first one's hits."""
    rng = synth.rng_for(53)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    refs = [synth.random_dna(rng, L) for L in (40000, 6000, 33, 2500, 12345, 700)]
    def sample(which, depth):
        reads = []
        for i in which:
            s = refs[i]
            for st in rng.integers(0, len(s) - 120, size=int(depth * len(s) / 120)):
                reads.append(synth.mutate(rng, s[st:st + 120], 0.005))
        return synth.reads_from_list(reads)
    rs_refs = synth.reads_from_list(refs)
    db, do = ctx.upload(rs_refs.bases), ctx.upload(rs_refs.offsets)
    ix = None
    try:
        ctx.eref_set_coder(hdr)
        ix = ctx.eref_probe_index_build(db, do, rs_refs.n, len(rs_refs.bases))
        ctx.eref_set_count_mode(2, cap)
        one_min, three_min = capi.window_minimums(0.9, 0.85)
        rows_f, rows_p = ctx.empty((rs_refs.n, 4), np.int32), ctx.empty((rs_refs.n, 4), np.int32)
        positives = 0
        for which, depth in (((0, 3, 4), 14), ((1, 5), 12), ((0, 1, 2, 3, 4, 5), 2)):      # three samples through one attached index
            rr = sample(which, depth)
            rb, ro = ctx.upload(rr.bases), ctx.upload(rr.offsets)
            t = orc.CountTable()
            t.count(rr.bases, rr.offsets, cc)
            want = [orc.scan_ref(orc.index_ref(s, cc), len(s), t, 0.9, 0.85)[1:3] for s in refs]
            t.free()
            # fused
            ctx.eref_attach_probe_index(ix)
            ctx.eref_set_option("final_count", 1)
            ctx.eref_table_reset()
            ctx.eref_count_reads(rb, ro, rr.n)
            ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), one_min, three_min, rows_f)
            got_f = rows_f.to_host()
            # the same count without the index attached: the scan probes for itself
            ctx.eref_attach_probe_index(None)
            ctx.eref_table_reset()
            ctx.eref_count_reads(rb, ro, rr.n)
            ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), one_min, three_min, rows_p)
            got_p = rows_p.to_host()
            assert np.array_equal(got_f, got_p), which
            assert [(int(a), int(b)) for a, b in got_f[:, :2]] == [(int(a), int(b)) for a, b in want], which
            # every entry set rides along and no plane is written (option probe_all_sets): same rows, from the hit bits alone; other
            # thresholds scan off the same bits; whatever else would read the table is refused; after the (free) reset the planes are
            # all zero -- a plain count then gives the table a fresh context's count gives (touched buckets were zeroed again)
            ctx.eref_attach_probe_index(ix)
            ctx.eref_set_option("final_count", 1)
            ctx.eref_set_option("probe_all_sets", 1)
            ctx.eref_table_reset()
            ctx.eref_count_reads(rb, ro, rr.n)
            ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), one_min, three_min, rows_p)
            assert np.array_equal(rows_p.to_host(), got_f), (which, "all sets fused")
            o2, t2 = capi.window_minimums(0.5, 0.2)
            ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), o2, t2, rows_p)
            rows_all_02 = rows_p.to_host().copy()
            with pytest.raises(capi.PalaceError):
                ctx.eref_scan_refs(db, do, rs_refs.n, len(rs_refs.bases), one_min, three_min, rows_p)
            with pytest.raises(capi.PalaceError):
                ctx.eref_table_popcounts()
            with pytest.raises(capi.PalaceError):
                ctx.eref_count_reads(rb, ro, rr.n)
            ctx.eref_set_option("probe_all_sets", 0)
            ctx.eref_set_option("final_count", 0)
            ctx.eref_attach_probe_index(None)
            ctx.eref_table_reset()                                   # (no memset: the planes are known to be zero)
            ctx.eref_count_reads(rb, ro, rr.n)
            pops_after = ctx.eref_table_popcounts()
            ctx.eref_scan_refs(db, do, rs_refs.n, len(rs_refs.bases), o2, t2, rows_p)
            assert np.array_equal(rows_p.to_host(), rows_all_02), (which, "all sets fused, other thresholds")
            ctx.eref_table_invalidate()                              # contents declared unknown: this reset clears the planes for real
            ctx.eref_table_reset()
            ctx.eref_count_reads(rb, ro, rr.n)
            assert ctx.eref_table_popcounts() == pops_after, (which, "planes were not all zero after the plane-less count")
            # attached, but the count is not a final one (three planes): nothing rides along, same rows
            ctx.eref_attach_probe_index(ix)
            ctx.eref_set_option("final_count", 0)
            ctx.eref_table_reset()
            ctx.eref_count_reads(rb, ro, rr.n)
            ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), one_min, three_min, rows_p)
            assert np.array_equal(rows_p.to_host(), got_f), which
            # a table whose fill is not known (here: declared so; really: planes merged from other ranks, or more reads counted than
            # the table has slots) takes the scan's two-stage pruning -- sentinels, then channel 0 exactly: same rows; with other
            # thresholds too (a low perfect_ratio makes every ref active in both stages)
            ctx.eref_table_invalidate()
            ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), one_min, three_min, rows_p)
            assert np.array_equal(rows_p.to_host(), got_f), which
            for hr, pr in ((0.5, 0.2), (0.9, 0.76), (0.05, 0.0)):
                o2, t2 = capi.window_minimums(hr, pr)
                ctx.eref_scan_refs(db, do, rs_refs.n, len(rs_refs.bases), o2, t2, rows_f)             # (the unindexed scan: recomputed keys, random probes)
                ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), o2, t2, rows_p)
                assert np.array_equal(rows_p.to_host(), rows_f.to_host()), (which, hr, pr, "dense")
            ctx.eref_table_reset()
            ctx.eref_count_reads(rb, ro, rr.n)
            for hr, pr in ((0.5, 0.2), (0.9, 0.76), (0.05, 0.0)):
                o2, t2 = capi.window_minimums(hr, pr)
                ctx.eref_scan_refs(db, do, rs_refs.n, len(rs_refs.bases), o2, t2, rows_f)
                ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), o2, t2, rows_p)
                assert np.array_equal(rows_p.to_host(), rows_f.to_host()), (which, hr, pr, "sparse")
            rb.free(); ro.free()
            positives += sum(1 for a, b in want if b > 0)
        assert positives >= 4                                   # (the deep samples report their refs, the shallow third one none)
    finally:
        ctx.eref_attach_probe_index(None)
        ctx.eref_set_option("final_count", 0)
        ctx.eref_set_option("probe_all_sets", 0)
        ctx.eref_set_count_mode(0, 0)
        if ix is not None:
            ctx.eref_probe_index_free(ix)
        ctx.eref_table_reset()
        db.free(); do.free()


@pytest.mark.parametrize("cap", [0, 64])                                       # 64: most keys take the overflow path (touched buckets are seeded)
def test_partial_entry_counts_of_read_shares_sum_to_the_hits_of_the_whole_sample(ctx, cap):
    """Option probe_all_sets 2 + palace_eref_entry_*: W shares of a sample's reads counted one after another (what W ranks do side by
    side), each leaving its partial counts of the DB's entries; the parts summed by entry range; the hit bits declared whole -- the
    indexed scan must then give the rows of ONE count over all reads (and the oracle's), also restricted to a range of the refs, with
    the blocks in caller-owned buffers as the collectives of a multi-GPU host need them, and for W that does not divide the block evenly."""
    import ctypes
    rng = synth.rng_for(91)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    refs = [synth.random_dna(rng, L) for L in (30000, 4000, 33, 2500, 9000, 700)]
    reads = []
    for i, depth in ((0, 9), (3, 14), (4, 7), (1, 2)):
        s_ = refs[i]
        for st in rng.integers(0, len(s_) - 120, size=int(depth * len(s_) / 120)):
            reads.append(synth.mutate(rng, s_[st:st + 120], 0.005))
    order = rng.permutation(len(reads))
    reads = [reads[k] for k in order]
    rs_refs = synth.reads_from_list(refs)
    db, do = ctx.upload(rs_refs.bases), ctx.upload(rs_refs.offsets)
    rr = synth.reads_from_list(reads)
    t = orc.CountTable()
    t.count(rr.bases, rr.offsets, cc)
    want = [orc.scan_ref(orc.index_ref(s_, cc), len(s_), t, 0.9, 0.85)[1:3] for s_ in refs]
    t.free()
    assert sum(1 for a, b in want if b > 0) >= 2
    ix = ix2 = None
    L = capi.lib()
    try:
        ctx.eref_set_coder(hdr)
        ix = ctx.eref_probe_index_build(db, do, rs_refs.n, len(rs_refs.bases))
        ix2 = ctx.eref_probe_index_build(db, do, rs_refs.n, len(rs_refs.bases))   # a second build of the same DB (another rank's): the same entries in the same order
        ctx.eref_set_count_mode(2, cap)
        one_min, three_min = capi.window_minimums(0.9, 0.85)
        rows = ctx.empty((rs_refs.n, 4), np.int32)
        cb, hb = ctx.eref_entry_layout(ix)
        assert ctx.eref_entry_layout(ix2) == (cb, hb)
        assert cb == 2 * hb and hb % (256 * 840) == 0
        for W, own_hits in ((1, False), (3, True), (8, False)):
            counts_buf = ctx.empty((cb,), np.uint8)                      # (caller-owned, as a host's collectives need it: read out below)
            hits_buf = ctx.empty((hb,), np.uint8) if own_hits else None
            ctx.eref_entry_buffers_attach(ix, counts_buf.ptr, hits_buf.ptr if own_hits else None)
            ctx.eref_entry_buffers_attach(ix2, counts_buf.ptr, None)
            parts = ctx.empty((W, cb), np.uint8)
            ctx.eref_set_option("final_count", 1)
            ctx.eref_set_option("probe_all_sets", 2)
            n_counting = max(1, W - 1)                                   # W > 1: the last "rank" takes no reads (rank 0 of a large sample does not)
            for r in range(W):
                share = synth.reads_from_list(reads[r * len(reads) // n_counting:(r + 1) * len(reads) // n_counting]) if r < n_counting else None
                ctx.eref_table_reset()
                assert not ctx.eref_entry_counts_valid(ix)                # nothing counted since the reset ...
                with pytest.raises(capi.PalaceError):                    # ... so the hit bits cannot be declared whole
                    ctx.eref_entry_hits_complete(ix, 0)
                if share is None:                                        # a rank without reads: the same call with n = 0 zeroes its block
                    capi._check(L.palace_memset(ctx.h, ctypes.c_void_p(counts_buf.ptr), 0x55, cb), "memset")
                    ctx.eref_attach_probe_index(ix)
                    capi._check(L.palace_eref_count_reads(ctx.h, None, None, 0, None, 0), "count of no reads")
                    assert ctx.eref_entry_counts_valid(ix)
                    ctx.d2d(parts.ptr + r * cb, counts_buf.ptr, cb)
                    ctx.sync()
                    assert not counts_buf.to_host().any()
                    continue
                sb, so = ctx.upload(share.bases), ctx.upload(share.offsets)
                ctx.eref_attach_probe_index(ix2 if r % 2 else ix)        # (odd "ranks" count through their own build of the index)
                if r == 0 and W > 1:
                    ctx.eref_set_count_mode(0, 0)                        # auto mode, a share far below 2^22 bases: still the fused, binned count
                ctx.eref_count_reads(sb, so, share.n)
                ctx.eref_set_count_mode(2, cap)
                with pytest.raises(capi.PalaceError):                    # one count call per reset: a second one could not be fused, and is refused
                    ctx.eref_count_reads(sb, so, share.n)
                ctx.eref_attach_probe_index(ix)
                assert ctx.eref_entry_counts_valid(ix)
                with pytest.raises(capi.PalaceError):                    # partial counts are nothing to scan from
                    ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), one_min, three_min, rows)
                ctx.d2d(parts.ptr + r * cb, counts_buf.ptr, cb)
                ctx.sync()
                sb.free(); so.free()
            # every "rank" sums its share of the block: cb / W bytes each (a multiple of 512)
            S = cb // W
            assert S * W == cb and S % 512 == 0
            for r in range(W):
                ctx.eref_entry_hits_from_counts(ix, parts.ptr + r * S, W, cb, r * S, S)      # (part p's counts OF THE RANGE lie at ptr + p * stride)
            ctx.eref_entry_hits_complete(ix, 3 * len(rr.bases))
            ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), one_min, three_min, rows)
            got = rows.to_host()
            assert [(int(a), int(b)) for a, b in got[:, :2]] == [(int(a), int(b)) for a, b in want], (W, own_hits)
            # a rank's range of the refs: the same rows inside, zero rows outside
            ctx.eref_set_option("scan_ref_lo", 1)
            ctx.eref_set_option("scan_ref_hi", 4)
            ctx.eref_scan_refs_indexed(ix, db, do, rs_refs.n, len(rs_refs.bases), one_min, three_min, rows)
            part = rows.to_host()
            assert np.array_equal(part[1:4], got[1:4]) and not part[[0, 4, 5], :2].any(), (W, "range")
            ctx.eref_set_option("scan_ref_lo", 0)
            ctx.eref_set_option("scan_ref_hi", 0)
            ctx.eref_table_reset()
            ctx.eref_entry_buffers_attach(ix, None, None)
            ctx.eref_entry_buffers_attach(ix2, None, None)
            parts.free(); counts_buf.free()
            if own_hits:
                hits_buf.free()
    finally:
        ctx.eref_attach_probe_index(None)
        ctx.eref_set_option("final_count", 0)
        ctx.eref_set_option("probe_all_sets", 0)
        ctx.eref_set_option("scan_ref_lo", 0)
        ctx.eref_set_option("scan_ref_hi", 0)
        ctx.eref_set_count_mode(0, 0)
        if ix is not None:
            ctx.eref_probe_index_free(ix)
        if ix2 is not None:
            ctx.eref_probe_index_free(ix2)
        ctx.eref_table_reset()
        db.free(); do.free()


def test_properties_full_size(ctx):
    """Size-independent properties at bench scale (2M reads): order independence, idempotence at
    saturation (x3 == x4), and the saturating merge of partial tables equals one-shot counting."""
    rng = synth.rng_for(9)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    pool = synth.random_dna(rng, 20_000_000)
    a = synth.vector_reads(rng, pool, 1_000_000, 150)
    b = synth.vector_reads(rng, pool, 1_000_000, 150)
    da, dao = ctx.upload(a.bases), ctx.upload(a.offsets)
    dbb, dbo = ctx.upload(b.bases), ctx.upload(b.offsets)
    ctx.eref_set_coder(hdr)
    probe = np.unique(rng.integers(0, 2**32, size=200000, dtype=np.uint64).astype(np.uint32))

    def run(seq):
        ctx.eref_table_reset()
        for (x, xo, n) in seq:
            ctx.eref_count_reads(x, xo, n)
        ctx.sync()
        return ctx.eref_table_popcounts(), ctx.eref_table_lookup(probe)

    A, B = (da, dao, a.n), (dbb, dbo, b.n)
    p_ab, l_ab = run([A, B])
    p_ba, l_ba = run([B, A])
    assert p_ab == p_ba and np.array_equal(l_ab, l_ba)
    p3, l3 = run([A, A, A])
    p4, l4 = run([A, A, A, A])
    assert p3 == p4 and np.array_equal(l3, l4) and p3[0] == p3[1] == p3[2]
    # merge: parts = [table(A), table(B)] -> planes == table(A then B)
    planes, nbytes = ctx.eref_table_planes()
    parts = ctx.empty(2 * 3 * nbytes, np.uint8)
    for k, S in enumerate((A, B)):
        run([S])
        for p in range(3):
            ctx.d2d(parts.ptr + (p * 2 + k) * nbytes, planes[p], nbytes)      # [plane][part][slice]
    ctx.eref_table_reset()
    ctx.eref_table_merge_slices(parts.ptr, 2, 0, nbytes)
    ctx.sync()
    assert ctx.eref_table_popcounts() == p_ab and np.array_equal(ctx.eref_table_lookup(probe), l_ab)
    # the same merge from two planes per part: (low bit of the count, count >= 2), layout [2][part][slice]
    packed = ctx.empty(2 * 2 * nbytes, np.uint8)
    for k, S in enumerate((A, B)):
        run([S])
        ctx.eref_table_pack_low(packed.ptr + (0 * 2 + k) * nbytes)
        ctx.d2d(packed.ptr + (1 * 2 + k) * nbytes, planes[1], nbytes)
    ctx.eref_table_reset()
    ctx.eref_table_merge_slices(packed.ptr, 2, 0, nbytes, packed=True)
    ctx.sync()
    assert ctx.eref_table_popcounts() == p_ab and np.array_equal(ctx.eref_table_lookup(probe), l_ab)
    for buf in (da, dao, dbb, dbo, parts, packed):
        buf.free()


def test_planes_written_by_the_caller_between_reset_and_count(ctx):
    """The contract of palace_eref_table_invalidate (include/palace_hip.h): a caller that writes the planes itself after a
    reset -- a collective receiving into attached planes -- says so, and the binned count then reads the slices instead of
    starting them from zero.  External bits: every key of read set A counted once (written with a device copy, as RCCL would)."""
    rng = synth.rng_for(31)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    pool = synth.random_dna(rng, 300_000)
    a = synth.vector_reads(rng, pool, 3000, 150)
    b = synth.vector_reads(rng, pool, 3000, 150)
    ctx.eref_set_coder(hdr)
    da, dao, dbb, dbo = ctx.upload(a.bases), ctx.upload(a.offsets), ctx.upload(b.bases), ctx.upload(b.offsets)
    probe = np.unique(rng.integers(0, 2**32, size=50000, dtype=np.uint64).astype(np.uint32))
    saved = ctx.empty(3 * (1 << 29), np.uint8)
    try:
        ctx.eref_set_count_mode(2, 0)                       # the binned kernels (the only ones with the clean-table fast path)
        ctx.eref_table_reset()
        ctx.eref_count_reads(da, dao, a.n)
        ctx.eref_count_reads(dbb, dbo, b.n)
        ctx.sync()
        want = (ctx.eref_table_popcounts(), ctx.eref_table_lookup(probe))
        planes, nbytes = ctx.eref_table_planes()
        ctx.eref_table_reset()
        ctx.eref_count_reads(da, dao, a.n)
        for p in range(3):
            ctx.d2d(saved.ptr + p * nbytes, planes[p], nbytes)
        ctx.eref_table_reset()
        for p in range(3):                                  # "received" planes: written behind the library's back
            ctx.d2d(planes[p], saved.ptr + p * nbytes, nbytes)
        ctx.eref_table_invalidate()
        ctx.eref_count_reads(dbb, dbo, b.n)
        ctx.sync()
        assert (ctx.eref_table_popcounts(), ) == (want[0], ) and np.array_equal(ctx.eref_table_lookup(probe), want[1])
    finally:
        ctx.eref_set_count_mode(0, 0)
        for buf in (da, dao, dbb, dbo, saved):
            buf.free()


@pytest.mark.parametrize("mode,cap", [(2, 0), (2, 64), (2, 1), (2, 140000)])
def test_binned_path_equals_oracle(ctx, golden_eref, mode, cap):
    """LDS-binned counting (forced on small inputs), incl. bucket overflow into the direct path; cap 140000 puts
    the upper regions beyond 2^31 keys / 2^33 bytes of the workspace (64-bit region addressing)."""
    g = golden_eref
    cc = orc.header_to_cc(g["index_header"])
    rng = synth.rng_for(13)
    dup = synth.reads_from_list([synth.random_dna(rng, 120)] * 300 + [np.frombuffer(b"ACGT" * 50, dtype=np.uint8)] * 50)
    sets = [(g["r1_bases"], g["r1_offsets"]), (g["r2_bases"], g["r2_offsets"]), (dup.bases, dup.offsets)]
    try:
        ctx.eref_set_count_mode(mode, cap)
        count_on_gpu(ctx, sets, g["index_header"])
    finally:
        ctx.eref_set_count_mode(0, 0)
    allb = np.concatenate([s[0] for s in sets])
    offs, base = [np.zeros(1, np.int64)], 0
    for b, o in sets:
        offs.append(np.asarray(o[1:], dtype=np.int64) + base)
        base += int(o[-1])
    u, c = oracle_key_counts(allb, np.concatenate(offs), cc)
    assert_table_equals(ctx, u, c)


def test_binned_equals_direct_at_scale(ctx):
    rng = synth.rng_for(17)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    a = synth.vector_reads(rng, synth.random_dna(rng, 30_000_000), 1_500_000, 150)
    keep = (rng.random(a.n) < 0.9).astype(np.uint8)
    da, dao, dk = ctx.upload(a.bases), ctx.upload(a.offsets), ctx.upload(keep)
    ctx.eref_set_coder(hdr)
    probe = np.unique(rng.integers(0, 2**32, size=300000, dtype=np.uint64).astype(np.uint32))
    res = []
    try:
        for mode in (1, 2):
            ctx.eref_set_count_mode(mode, 0)
            ctx.eref_table_reset()
            ctx.eref_count_reads(da, dao, a.n, dk)
            ctx.eref_count_reads(da, dao, a.n)                 # second pass over the same reads: counts reach 2..3
            ctx.sync()
            res.append((ctx.eref_table_popcounts(), ctx.eref_table_lookup(probe)))
    finally:
        ctx.eref_set_count_mode(0, 0)
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    assert res[0][0][2] > 0
    for b in (da, dao, dk):
        b.free()


def test_binned_equals_direct_on_skewed_keys(ctx):
    """Regions and staging rows are sized by the 2(1-x) density of hash-like keys.  Low-complexity input defeats
    that model: a few distinct 32-mers, repeated by the million, pile onto a handful of rows and regions, which must
    overflow into the exact direct path without changing the table."""
    rng = synth.rng_for(23)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    unit = [synth.random_dna(rng, 150) for _ in range(3)] + [np.frombuffer(b"A" * 150, dtype=np.uint8),
                                                             np.frombuffer(b"AC" * 75, dtype=np.uint8)]
    pick = rng.integers(0, len(unit), size=120_000)
    a = synth.reads_from_list([unit[i] for i in pick])                    # 18 Mbases, ~600 distinct keys
    b = synth.vector_reads(rng, synth.random_dna(rng, 2_000_000), 60_000, 150)   # plus a normal share
    bases = np.concatenate([a.bases, b.bases])
    offsets = np.concatenate([a.offsets, b.offsets[1:] + a.offsets[-1]])
    db_, do_ = ctx.upload(bases), ctx.upload(offsets)
    ctx.eref_set_coder(hdr)
    probe = np.unique(rng.integers(0, 2**32, size=200000, dtype=np.uint64).astype(np.uint32))
    res = []
    try:
        for mode in (1, 2):
            ctx.eref_set_count_mode(mode, 0)
            ctx.eref_table_reset()
            ctx.eref_count_reads(db_, do_, len(offsets) - 1)
            ctx.sync()
            res.append((ctx.eref_table_popcounts(), ctx.eref_table_lookup(probe)))
    finally:
        ctx.eref_set_count_mode(0, 0)
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    assert res[0][0][2] > 0
    for x in (db_, do_):
        x.free()


def test_binned_flat_stream_with_offset_base_and_ragged_reads(ctx):
    """flat streaming bin kernel: read set that does not start at offset 0, reads of every length
    class (0, <32, ==32, long), N inside reads, read ends adjacent to chunk boundaries"""
    rng = synth.rng_for(23)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    pool = synth.random_dna(rng, 400000)
    lens = [0, 1, 31, 32, 33, 63, 64, 65, 95, 96, 127, 128, 129, 150, 150, 1000, 0, 40] * 200
    reads, p = [], 0
    for L in lens:
        r = pool[p:p + L].copy(); p = (p + L + 7) % 300000
        if L > 50 and rng.random() < 0.1:
            r[int(rng.integers(0, L))] = ord("N")
        reads.append(r)
    rs = synth.reads_from_list(reads)
    junk = synth.random_dna(rng, 777)
    bases = np.concatenate([junk, rs.bases])
    offsets = rs.offsets + 777
    try:
        ctx.eref_set_count_mode(2, 0)
        ctx.eref_set_coder(hdr)
        ctx.eref_table_reset()
        db, do = ctx.upload(bases), ctx.upload(offsets)
        ctx.eref_count_reads(db, do, rs.n, None, int(offsets[-1] - offsets[0]))
        ctx.sync()
    finally:
        ctx.eref_set_count_mode(0, 0)
    u, c = oracle_key_counts(rs.bases, rs.offsets, cc)
    assert_table_equals(ctx, u, c)


@pytest.mark.parametrize("with_keep", [False, True])
def test_binned_slabs(ctx, with_keep):
    """large read sets are processed in slabs, a slab's level 1 in parts on a second stream beside level 2 of the part before;
    force tiny slabs (and three parts per slab) and compare with the oracle"""
    rng = synth.rng_for(29)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    rs = synth.vector_reads(rng, synth.random_dna(rng, 200000), 5000, 123)     # 615 kbases, reads straddle slab edges
    keep = (rng.random(rs.n) < 0.7).astype(np.uint8) if with_keep else None
    try:
        ctx.eref_set_count_mode(2, 0)
        ctx.eref_set_option("slab_bases", 64 * 1024)                                # 64 Ki-base slabs -> 10 slabs
        count_on_gpu(ctx, [(rs.bases, rs.offsets)], hdr, keep=[keep] if with_keep else None)
    finally:
        ctx.eref_set_option("slab_bases", 0)
        ctx.eref_set_count_mode(0, 0)
    kept = synth.reads_from_list([rs.read(i) for i in range(rs.n) if keep is None or keep[i]])
    u, c = oracle_key_counts(kept.bases, kept.offsets, cc)
    assert_table_equals(ctx, u, c)



def pack_reads(bases, offsets, keep=None, gap_every=0):
    """a read set as the three bit streams of palace_eref_count_reads_packed (include/palace_hip.h), stated with numpy per
    base; gap_every > 0 puts positions of no read between the reads now and then (what parser threads leave between their parts)"""
    bases, offsets = np.asarray(bases, np.uint8), np.asarray(offsets, np.int64)
    n = len(offsets) - 1
    lens = offsets[1:] - offsets[:-1]
    gaps = np.array([(37 if gap_every and r % gap_every == 0 else 0) for r in range(n)], np.int64)
    start = np.concatenate([[0], np.cumsum(lens + gaps)[:-1]]) + gaps          # packed position of every read's first base
    n_pos = int((lens + gaps).sum())
    pos = np.repeat(start - (offsets[:-1] - offsets[0]), lens) + np.arange(offsets[0], offsets[-1]) - offsets[0]
    up = bases[offsets[0]:offsets[-1]] & 0xDF
    ok = np.isin(up, np.frombuffer(b"ACGT", np.uint8))
    if keep is not None:
        ok &= np.repeat(np.asarray(keep, bool), lens)
    v0, v1, valid = (np.zeros(n_pos + 32, np.uint8) for _ in range(3))
    v0[pos] = np.isin(up, np.frombuffer(b"AT", np.uint8)); v1[pos] = np.isin(up, np.frombuffer(b"AC", np.uint8)); valid[pos] = ok
    cs = np.concatenate([[0], np.cumsum(1 - valid.astype(np.int64))])
    all_ok = (cs[32:n_pos + 32] - cs[:n_pos]) == 0                              # the 32 positions from p on are counted bases ...
    end = np.zeros(n_pos, np.int64); end[pos] = np.repeat(start + lens, lens)
    u = all_ok & (np.arange(n_pos) + 32 <= end)                                 # ... of one read
    n_bytes = int(capi.lib().palace_eref_packed_bytes(n_pos))
    out = []
    for bits in (v0[:n_pos], v1[:n_pos], u.astype(np.uint8)):
        b = np.zeros(n_bytes, np.uint8)
        pk = np.packbits(bits, bitorder="little")
        b[:len(pk)] = pk
        b[len(pk):] = 0xA5 if bits is not u else 0                              # look-ahead words: content is ignored
        out.append(b)
    return out, n_pos


@pytest.mark.parametrize("mode", [1, 2])                                        # direct kernel / binned
@pytest.mark.parametrize("with_keep,gap_every", [(False, 0), (True, 5)])
def test_packed_entry_equals_ascii_entry(ctx, mode, with_keep, gap_every):
    """palace_eref_count_reads_packed == palace_eref_count_reads == oracle on the same reads: ragged lengths (0, 31, 32, 33),
    invalid bases, lower case, dropped reads, positions of no read between reads."""
    rng = synth.rng_for(41)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    genome = synth.random_dna(rng, 60000)
    with_n = genome[300:400].copy(); with_n[40] = ord("N")
    reads = [np.zeros(0, np.uint8), np.full(31, ord("A"), np.uint8), genome[:32], genome[100:133], genome[200:300] | 0x20, with_n]
    for _ in range(3000):
        a, L = int(rng.integers(0, 59000)), int(rng.integers(20, 400))
        s = genome[a:a + L].copy()
        if rng.random() < 0.2:
            s[int(rng.integers(0, len(s)))] = ord("N")
        reads.append(s)
    rs = synth.reads_from_list(reads)
    keep = (rng.random(rs.n) < 0.6).astype(np.uint8) if with_keep else None
    kept = synth.reads_from_list([rs.read(i) for i in range(rs.n) if keep is None or keep[i]])
    u, c = oracle_key_counts(kept.bases, kept.offsets, cc)
    streams, n_pos = pack_reads(rs.bases, rs.offsets, keep, gap_every)
    try:
        ctx.eref_set_count_mode(mode, 0)
        if mode == 2:
            ctx.eref_set_option("slab_bases", 64 * 1024)
        ctx.eref_set_coder(hdr)
        ctx.eref_table_reset()
        d = [ctx.upload(s) for s in streams]
        ctx.eref_count_reads_packed(d[0], d[1], d[2], n_pos, rs.n)
        ctx.sync()
        for b in d:
            b.free()
        assert_table_equals(ctx, u, c)
        count_on_gpu(ctx, [(rs.bases, rs.offsets)], hdr, keep=[keep] if with_keep else None)
        assert_table_equals(ctx, u, c)
    finally:
        ctx.eref_set_option("slab_bases", 0)
        ctx.eref_set_count_mode(0, 0)


def test_packed_entry_argument_checks(ctx):
    hdr = orc.header_from_picks(np.zeros(32, np.int64))
    ctx.eref_set_coder(hdr)
    ctx.eref_table_reset()
    d = ctx.upload(np.zeros(64, np.uint8))
    ctx.eref_count_reads_packed(d, d, d, 0)                                     # nothing to count
    with pytest.raises(capi.PalaceError):
        capi._check(capi.lib().palace_eref_count_reads_packed(ctx.h, d.ptr + 4, d.ptr, d.ptr, 100, 0), "packed")   # misaligned stream
    with pytest.raises(capi.PalaceError):
        capi._check(capi.lib().palace_eref_count_reads_packed(ctx.h, None, d.ptr, d.ptr, 100, 0), "packed")
    d.free()
    assert capi.lib().palace_eref_packed_bytes(0) == 16 and capi.lib().palace_eref_packed_bytes(65) == 32


@pytest.mark.parametrize("with_keep", [False, True])
def test_pack_reads_on_the_device_equals_the_per_base_statement(ctx, with_keep):
    """palace_eref_pack_reads: the streams the ASCII entry makes for itself, handed out -- bit-identical to the numpy statement
    (U everywhere; P0 / P1 at every valid base)."""
    rng = synth.rng_for(43)
    genome = synth.random_dna(rng, 30000)
    reads = [np.zeros(0, np.uint8), genome[:31], genome[:32], genome[50:83] | 0x20]
    for _ in range(1500):
        a, L = int(rng.integers(0, 29000)), int(rng.integers(1, 300))
        s = genome[a:a + L].copy()
        if rng.random() < 0.3:
            s[int(rng.integers(0, len(s)))] = rng.choice(list(b"NnX.-"))
        reads.append(s)
    rs = synth.reads_from_list(reads)
    keep = (rng.random(rs.n) < 0.5).astype(np.uint8) if with_keep else None
    want, n_pos = pack_reads(rs.bases, rs.offsets, keep)
    assert n_pos == len(rs.bases)
    db, do = ctx.upload(rs.bases), ctx.upload(rs.offsets)
    dk = ctx.upload(keep) if with_keep else None
    out = [ctx.upload(np.full(len(want[0]), 0x5A, np.uint8)) for _ in range(3)]
    ctx.eref_pack_reads(db, do, rs.n, dk, n_pos, out[0], out[1], out[2])
    ctx.sync()
    got = [np.unpackbits(o.to_host(), bitorder="little")[:n_pos] for o in out]
    exp = [np.unpackbits(w, bitorder="little")[:n_pos] for w in want]
    valid = np.isin(rs.bases & 0xDF, np.frombuffer(b"ACGT", np.uint8))
    assert np.array_equal(got[2], exp[2])
    assert np.array_equal(got[0][valid], exp[0][valid]) and np.array_equal(got[1][valid], exp[1][valid])
    for b in [db, do] + out + ([dk] if dk else []):
        b.free()


@pytest.mark.parametrize("cap", [0, 64, 1])                                    # 64 / 1: most keys take the overflow path of the partition kernels
def test_final_count_keeps_only_the_top_plane_and_leaves_the_others_zero(ctx, cap):
    """option final_count (include/palace_hip.h): plane ">= 3" as without the option, the two lower planes all zero -- also
    where overflow keys had written into them --, further counts / lookups refused until the reset, and the reset (which now
    clears one plane only) gives a table that counts exactly again."""
    rng = synth.rng_for(47)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    rs = synth.vector_reads(rng, synth.random_dna(rng, 40000), 4000, 120)      # ~12x coverage: many keys reach 3
    u, c = oracle_key_counts(rs.bases, rs.offsets, cc)
    top = u[c >= 3]
    assert len(top) > 1000 and (c == 1).sum() > 100
    db, do = ctx.upload(rs.bases), ctx.upload(rs.offsets)
    try:
        ctx.eref_set_count_mode(2, cap)
        ctx.eref_set_option("final_count", 1)
        ctx.eref_set_coder(hdr)
        ctx.eref_table_reset()
        ctx.eref_count_reads(db, do, rs.n)
        assert ctx.eref_table_popcounts() == [0, 0, len(top)]
        ptrs, nbytes = ctx.eref_table_planes()
        plane3 = capi.DevBuf.__new__(capi.DevBuf)                               # a view of the context's plane, not owned
        plane3.ctx, plane3.ptr, plane3.nbytes, plane3.dtype, plane3.shape = ctx, ptrs[2], nbytes, np.dtype(np.uint32), (nbytes // 4,)
        words = plane3.to_host()
        assert np.all((words[top >> 5] >> (top & 31)) & 1)                      # popcount equal + every expected bit set = the same set
        with pytest.raises(capi.PalaceError):
            ctx.eref_count_reads(db, do, rs.n)
        with pytest.raises(capi.PalaceError):
            ctx.eref_table_lookup(u[:10])
        ctx.eref_set_option("final_count", 0)
        ctx.eref_table_reset()                                                  # clears plane 3 only
        ctx.eref_count_reads(db, do, rs.n)
        ctx.sync()
        assert_table_equals(ctx, u, c)
    finally:
        ctx.eref_set_option("final_count", 0)
        ctx.eref_set_count_mode(0, 0)
        ctx.eref_table_reset()
        db.free(); do.free()


@pytest.mark.parametrize("mode", [1, 2])                                        # direct kernels / partition kernels
def test_key_buckets_count_exactly_their_share_of_the_key_space(ctx, mode):
    """palace_eref_set_key_buckets: four calls over the same reads, each with the buckets a rank of four takes (mirrored pairs
    {r, 7 - r} of every 8: equal key mass), give four tables that are exact on their buckets and empty elsewhere -- what four
    GPUs that each hold all reads would gather --, through the ASCII and the packed entry, with final_count as the bench uses it"""
    from palace_amd import multigpu
    rng = synth.rng_for(53)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    rs = synth.vector_reads(rng, synth.random_dna(rng, 50000), 4000, 110)
    u, c = oracle_key_counts(rs.bases, rs.offsets, cc)
    streams, n_pos = pack_reads(rs.bases, rs.offsets)
    db, do = ctx.upload(rs.bases), ctx.upload(rs.offsets)
    ds = [ctx.upload(x) for x in streams]
    shares = [multigpu.key_buckets_of(r, 4) for r in range(4)]
    assert sorted(b for sh in shares for b in sh) == list(range(128))
    mass = [sum(255 - 2 * b for b in sh) for sh in shares]
    assert max(mass) == min(mass)                                               # the linear key density folds exactly
    try:
        ctx.eref_set_count_mode(mode, 0)
        ctx.eref_set_coder(hdr)
        total3 = 0
        for q in range(4):
            mine = np.isin(u >> 25, shares[q])
            assert 0.2 < mine.mean() < 0.3
            ctx.eref_set_key_buckets(shares[q])
            ctx.eref_table_reset()
            ctx.eref_count_reads(db, do, rs.n)
            ctx.sync()
            assert_table_equals(ctx, u[mine], c[mine])                         # its keys exactly, and nothing else is set
            ctx.eref_table_reset()
            ctx.eref_count_reads_packed(ds[0], ds[1], ds[2], n_pos, rs.n)
            ctx.sync()
            assert_table_equals(ctx, u[mine], c[mine])
            if mode == 2:
                ctx.eref_set_option("final_count", 1)
                ctx.eref_table_reset()
                ctx.eref_count_reads_packed(ds[0], ds[1], ds[2], n_pos, rs.n)
                pops = ctx.eref_table_popcounts()
                assert pops[:2] == [0, 0] and pops[2] == int((c[mine] >= 3).sum())
                total3 += pops[2]
                ctx.eref_set_option("final_count", 0)
                ctx.eref_table_reset()
        if mode == 2:
            assert total3 == int((c >= 3).sum())
        with pytest.raises(capi.PalaceError):
            ctx.eref_set_key_buckets([])                                        # an empty share is refused
    finally:
        ctx.eref_set_key_buckets(None)
        ctx.eref_set_option("final_count", 0)
        ctx.eref_set_count_mode(0, 0)
        ctx.eref_table_reset()
        for b in [db, do] + ds:
            b.free()


def test_plane_sparse_pack_and_unpack_round_trip(ctx):
    """palace_eref_plane_pack / _unpack (what ranks exchange instead of plane slices): counts per fine bucket = set bits of the
    '>= 3' plane there, keys = their offsets, ascending; unpacked into ANOTHER context's table bucket share by bucket share they
    give the same plane; too little room is reported by the total, and keys behind the room are not written"""
    from palace_amd import multigpu
    rng = synth.rng_for(77)
    hdr = orc.header_from_picks(rng.integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    src = synth.random_dna(rng, 60000)
    rs = synth.vector_reads(rng, src, 9000, 120)                       # ~18x: many keys reach 3
    u, c = oracle_key_counts(rs.bases, rs.offsets, cc)
    three = np.sort(u[c >= 3])
    assert len(three) > 20000
    count_on_gpu(ctx, [(rs.bases, rs.offsets)], hdr)
    other = capi.Ctx(0)
    try:
        other.eref_set_coder(hdr)
        other.eref_table_reset()
        total = 0
        for r in range(4):
            share = multigpu.key_buckets_of(r, 4)
            mine = three[np.isin(three >> 25, share)]
            n_fine = 512 * len(share)
            d_cnt, d_first = ctx.upload(np.zeros(n_fine, np.uint32)), ctx.upload(np.zeros(n_fine + 1, np.uint64))
            cap = len(mine) + 100
            d_keys = ctx.upload(np.full(cap, 0xABCD, np.uint16))
            ctx.eref_plane_pack(share, d_cnt.ptr, d_keys.ptr, cap, d_first.ptr)
            ctx.sync()
            cnt, first, keys = d_cnt.to_host(), d_first.to_host(), d_keys.to_host()
            assert int(first[-1]) == len(mine) == int(cnt.sum()) and np.array_equal(first[:-1], np.concatenate([[0], np.cumsum(cnt)[:-1]]))
            fine_of = np.concatenate([np.arange(b * 512, (b + 1) * 512) for b in sorted(share)])             # slot -> fine bucket
            got = (np.repeat(fine_of, cnt).astype(np.uint64) << np.uint64(16)) | keys[:len(mine)].astype(np.uint64)
            assert np.array_equal(got.astype(np.uint32), mine)                                               # ascending inside and across buckets
            assert (keys[len(mine):] == 0xABCD).all()
            # too little room: the total still says what is needed, nothing is written behind the room
            small = ctx.upload(np.full(len(mine), 0xABCD, np.uint16))
            ctx.eref_plane_pack(share, d_cnt.ptr, small.ptr, len(mine) // 2, d_first.ptr)
            ctx.sync()
            assert int(d_first.to_host()[-1]) == len(mine) and (small.to_host()[len(mine) // 2:] == 0xABCD).all()
            # into the other context's table -- first from a key buffer that was too small for the sender (the counts say more
            # keys than the room holds): only what is there is looked at, nothing beyond the room is read
            o_cnt, o_first = other.upload(cnt), other.upload(np.zeros(n_fine + 1, np.uint64))
            o_small = other.upload(keys[:len(mine) // 2].copy())
            other.eref_plane_unpack(share, o_cnt.ptr, o_small.ptr, len(mine) // 2, o_first.ptr)
            other.sync()
            assert other.eref_table_popcounts()[2] == total + len(mine) // 2
            o_keys = other.upload(keys)
            other.eref_plane_unpack(share, o_cnt.ptr, o_keys.ptr, cap, o_first.ptr)
            other.sync()
            total += len(mine)
            assert other.eref_table_popcounts()[2] == total
        assert total == len(three)
        probe = np.unique(np.concatenate([three, rng.integers(0, 2**32, size=50000, dtype=np.uint64).astype(np.uint32)]))
        assert np.array_equal(other.eref_table_lookup(probe) != 0, np.isin(probe, three))      # (only the ">= 3" plane travels)
    finally:
        other.close()
