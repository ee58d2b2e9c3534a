"""world_size-2 and world_size-8 gloo runs of palace_amd/multigpu.Exchange on CPU tensors: the byte movement of the
N>1 path (plane-slice all_to_all + owner merge + all_gather, padded row gathers, var-length candidate
gather, depth reduce).  The merge arithmetic is the HIP library's in production; here a torch
statement of the same bit-plane algebra stands in so the exchange can be checked end to end."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from palace_amd import multigpu


def planes_from_counts(c):
    """unary planes (count>=1, >=2, >=3) packed 8 keys per byte"""
    return [torch.from_numpy(np.packbits((c >= t).astype(np.uint8), bitorder="little")) for t in (1, 2, 3)]


def torch_merge(planes):
    def fn(parts, n_parts, slice_off, slice_bytes, packed=False):
        a1 = torch.zeros(slice_bytes, dtype=torch.uint8); a2 = a1.clone(); a3 = a1.clone()
        for p in range(n_parts):
            if packed:                                  # (low bit, high bit) of the count -> unary planes
                lo, hi = parts[0, p], parts[1, p]
                b1, b2, b3 = lo | hi, hi, lo & hi
            else:
                b1, b2, b3 = parts[0, p], parts[1, p], parts[2, p]
            a3 = a3 | b3 | (a2 & b1) | (a1 & b2)
            a2 = a2 | b2 | (a1 & b1)
            a1 = a1 | b1
        for pl, v in zip(planes, (a1, a2, a3)):
            pl[slice_off:slice_off + slice_bytes] = v
    return fn


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = multigpu.Exchange(torch, dist, rank, world)
        rng = np.random.Generator(np.random.PCG64(5))
        n_keys = 1 << 16
        counts = [rng.integers(0, 4, size=n_keys) * (rng.random(n_keys) < 0.3) for _ in range(world)]
        want = planes_from_counts(np.minimum(3, sum(counts)))
        ok_planes = True
        for packed in (False, True):                    # three unary planes per peer, or (low bit, count >= 2)
            planes = planes_from_counts(counts[rank])
            pack = (lambda pl=planes: pl[0] ^ pl[1] ^ pl[2]) if packed else None
            ex.merge_planes(planes, torch_merge(planes), pack)
            ok_planes &= bool(torch.equal(planes[2], want[2]))
            S = planes[0].numel() // world
            own = slice(rank * S, (rank + 1) * S)
            ok_planes &= bool(torch.equal(planes[0][own], want[0][own]) and torch.equal(planes[1][own], want[1][own]))
        # rows by ranges
        lens = rng.integers(100, 1000, size=37)
        ranges = [multigpu.split_by_weight(lens, r, world) for r in range(world)]
        full = torch.from_numpy(rng.integers(0, 1000, size=(37, 4)).astype(np.int32))
        table = torch.zeros_like(full)
        lo, hi = ranges[rank]
        table[lo:hi] = full[lo:hi]
        ex.gather_ranges(table, ranges)
        ok_rows = bool(torch.equal(table, full)) and ranges[0][0] == 0 and ranges[-1][1] == 37
        # var-length candidates
        n_mine = [5, 0, 11, 3][rank % 4] if world > 1 else 5
        allc = [torch.full((n, 64), r + 1, dtype=torch.uint8) for r, n in enumerate([[5, 0, 11, 3][r % 4] for r in range(world)])]
        buf = torch.zeros((16, 64), dtype=torch.uint8)
        buf[:n_mine] = allc[rank]
        got, total = ex.gather_varlen(buf, n_mine)
        ok_var = total == sum(len(x) for x in allc) and bool(torch.equal(got, torch.cat(allc)))
        # the same gather with nothing read back: rows padded to a width, zero rows beyond a rank's count, counts on the "device"
        width = 12
        buf[n_mine:] = 0xAB                                # stale rows behind the valid ones must not travel
        padded, cnts = ex.gather_padded(buf, torch.tensor([n_mine], dtype=torch.int64), width)
        want_pad = torch.zeros((world * width, 64), dtype=torch.uint8)
        for r, x in enumerate(allc):
            want_pad[r * width:r * width + len(x)] = x
        ok_var &= bool(torch.equal(padded, want_pad)) and cnts.tolist() == [len(x) for x in allc] and ex.last_counts == cnts.tolist()
        narrow, cnts = ex.gather_padded(buf, torch.tensor([n_mine], dtype=torch.int64), 4)      # too narrow: the counts say so
        ok_var &= (max(cnts.tolist()) > 4) == (max(len(x) for x in allc) > 4) and narrow.shape[0] == 4 * world
        t = torch.full((9,), rank + 1, dtype=torch.int64)
        ex.reduce_sum(t)
        ok_sum = bool((t == sum(range(1, world + 1))).all())
        # key-space split: a rank holds the plane slices of its buckets (mirrored pairs), afterwards everybody holds the plane
        full_plane = torch.from_numpy(rng.integers(0, 256, size=128 * 64).astype(np.uint8))
        plane = torch.zeros_like(full_plane)
        for b in multigpu.key_buckets_of(rank, world):
            plane.view(128, -1)[b] = full_plane.view(128, -1)[b]
        ex.gather_key_buckets(plane)
        ok_keys = bool(torch.equal(plane, full_plane))
        shares = [multigpu.key_buckets_of(r, world) for r in range(world)]
        ok_keys &= sorted(b for sh in shares for b in sh) == list(range(128)) and len({sum(255 - 2 * b for b in sh) for sh in shares}) == 1
        # ... and in sparse form: a count per fine bucket + the 16-bit offsets of its set bits (numpy stand-ins for
        # palace_eref_plane_pack / _unpack; the "plane" here has 2^16 bits per fine bucket as the real one, 8 fine buckets per
        # level-1 bucket instead of 512 -- the exchange does not care), with room that fits and room that does not
        FINE = 8
        dense = (rng.random((128, FINE, 1 << 16)) < 0.004)                   # every rank draws the same plane
        mine_b = multigpu.key_buckets_of(rank, world)

        def np_pack(buckets, cap):
            cnt = np.array([dense[b, f].sum() for b in buckets for f in range(FINE)], dtype=np.int32)
            keys = np.concatenate([np.flatnonzero(dense[b, f]) for b in buckets for f in range(FINE)]).astype(np.uint16)
            out = np.zeros(cap, dtype=np.uint16)
            out[:min(cap, len(keys))] = keys[:cap]
            return cnt, out.view(np.int16)
        got_plane = np.zeros_like(dense)
        for b in mine_b:
            got_plane[b] = dense[b]

        def pack_fn(counts, keys, first, cap):
            c, k = np_pack(mine_b, cap)
            counts.copy_(torch.from_numpy(c)); keys.copy_(torch.from_numpy(k))

        def unpack_fn(r, counts, keys, first):
            c, k = counts.numpy(), keys.numpy().view(np.uint16)
            off = np.concatenate([[0], np.cumsum(c)])
            for j, b in enumerate(multigpu.key_buckets_of(r, world)):
                for f in range(FINE):
                    row = np.zeros(1 << 16, dtype=bool)
                    row[k[off[j * FINE + f]:off[j * FINE + f + 1]]] = True
                    got_plane[b, f] = row
        orig_gather = multigpu.Exchange.gather_buckets_sparse

        def gather(cap):
            bufs = {"device": "cpu"}
            # (the method sizes its buffers for 512 fine buckets per level-1 bucket: give it lists of FINE / 512 "buckets" worth)
            bl = [multigpu.key_buckets_of(r, world) for r in range(world)]
            n_fine = FINE * len(bl[0])
            bufs.update(cap=cap, n_fine=512 * len(bl[0]))            # pre-sized below instead
            bufs.update(counts=torch.zeros(n_fine, dtype=torch.int32), keys=torch.zeros(cap, dtype=torch.int16), first=torch.zeros(n_fine + 1, dtype=torch.int64),
                        counts_all=torch.zeros((world, n_fine), dtype=torch.int32), keys_all=torch.zeros((world, cap), dtype=torch.int16))
            return ex.gather_buckets_sparse(bl, lambda c, k, f: pack_fn(c, k, f, cap), unpack_fn, cap, bufs)
        need = int(dense.sum(axis=(1, 2)).reshape(128)[mine_b].sum())
        roomy = max(int(dense.sum(axis=(1, 2))[multigpu.key_buckets_of(r, world)].sum()) for r in range(world)) + 64
        sums = gather(roomy).sum(dim=1)
        ok_sparse = bool(np.array_equal(got_plane, dense)) and int(sums[rank]) == need and int(sums.max()) <= roomy
        tight = gather(roomy // 2).sum(dim=1)                                # too little room: the counts say so (the caller redoes the step)
        ok_sparse &= int(tight.max()) > roomy // 2
        # reads sharded, partial COUNTS of the DB's probe-index entries exchanged instead of planes (merge_entry_counts): two bits per
        # entry, 16 bits per vector of eight; a numpy statement of palace_eref_entry_hits_from_counts stands in for the library's sum
        n_vec = 256 * world * 3                                             # u16 per vector: the block is 2 * n_vec bytes, a multiple of 512 * world
        per_rank = [rng.integers(0, 4, size=(n_vec, 8)) * (rng.random((n_vec, 8)) < 0.4) for _ in range(world)]    # every rank draws every rank's counts
        if world > 2:
            per_rank[0][:] = 0                                              # (a rank that takes no reads)
        def pack16(c):
            return (c.astype(np.uint16) << (2 * np.arange(8, dtype=np.uint16))).sum(axis=1).astype("<u2")
        counts_t = torch.from_numpy(pack16(per_rank[rank]).view(np.uint8).copy())
        hits_t = torch.zeros(n_vec, dtype=torch.uint8)
        def np_sum(parts, n_parts, stride, off, nbytes):
            a = parts.numpy().view("<u2").reshape(n_parts, stride // 2)[:, :nbytes // 2].astype(np.uint32)
            byte = np.zeros(nbytes // 2, dtype=np.uint8)
            for e in range(8):
                byte |= ((((a >> (2 * e)) & 3).sum(axis=0) >= 3).astype(np.uint8) << e)
            hits_t[off // 2:off // 2 + nbytes // 2] = torch.from_numpy(byte)
        ex.merge_entry_counts(counts_t, hits_t, np_sum)
        want_hits = np.packbits((np.minimum(3, sum(per_rank)) >= 3).astype(np.uint8), axis=1, bitorder="little").reshape(-1)
        ok_counts = bool(np.array_equal(hits_t.numpy(), want_hits)) and int(want_hits.sum()) > 0
        q.put((rank, ok_planes, ok_rows, ok_var, ok_sum, ok_keys and ok_sparse and ok_counts))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world", [2, 8])                 # 8 = the node the driver scales to, rehearsed over gloo
def test_exchange_world(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def test_phase_a_scheme_model_picks_by_size():
    """the cost model behind bench.py's scheme choice: 1M contigs on 4 or 8 GPUs -> key split, 5M contigs
    on 8 -> read-sharded exchange, two GPUs -> every rank counts everything (a half plane would cross ONE link)"""
    m = multigpu.phase_a_model
    assert m(6_666_666, 8)["choice"] == "key_split" and m(6_666_666, 4)["choice"] == "key_split"
    assert m(33_333_333, 8)["choice"] == "shard_reads" and m(33_333_333, 4, sparse=False)["choice"] == "shard_reads"
    assert m(6_666_666, 1)["choice"] == "replicate" and set(m(6_666_666, 1)["ms"]) == {"replicate"}
    assert m(6_666_666, 2, sparse=False)["choice"] == "replicate"                     # dense slices: half a plane would cross ONE link
    # the plane in sparse form (48 MB instead of 512 MiB at 1M contigs): the key split pays from two GPUs on
    assert m(6_666_666, 2)["choice"] == "key_split" and m(6_666_666, 8)["ms"]["key_split"] < m(6_666_666, 8, sparse=False)["ms"]["key_split"] - 1.0
    assert "key_split" not in m(6_666_666, 3)["ms"]                                   # shares are mirrored bucket pairs: W must divide 64
    slow = m(6_666_666, 8, link_gbs=5.0, sparse=False)                                 # a slow interconnect, dense slices: nothing beats counting everything
    assert slow["choice"] == "replicate" and slow["link_gbs"] == 5.0
    assert m(33_333_333, 8, link_gbs=5.0, sparse=False)["choice"] == "key_split"       # ... until the sample is large: the one gather pays
    assert m(6_666_666, 8, link_gbs=5.0)["choice"] == "key_split"                      # (in sparse form the gather is small even then)
    for w in (2, 4, 8):
        r = m(6_666_666, w)
        assert r["ms"][r["choice"]] == min(r["ms"].values()) and r["world"] == w


def test_step_model_gives_rank_0_to_stage_04_for_large_samples():
    """the whole-step model (serial terms included): on 8 GPUs stage 04 beside a count launch would be rank 0's critical path once the
    reads are sharded, so rank 0 takes no reads; on one GPU every read is counted where stage 04 runs"""
    big = multigpu.best_step(5_000_000, 33_333_333, 8)
    assert big["scheme"] == "shard_counts" and big["rank0_counts"] is False and big["stream_a_ms"] > big["stream_b_rank0_ms"]
    for scheme in ("shard_reads", "shard_counts"):
        with0, idle = (multigpu.step_model(5_000_000, 33_333_333, 8, scheme, r0) for r0 in (True, False))
        assert with0["step_ms"] > idle["step_ms"] and with0["stream_b_rank0_ms"] > with0["stream_a_ms"]
    small = multigpu.best_step(1_000_000, 6_666_666, 8)
    assert small["scheme"] == "shard_counts" and small["rank0_counts"] is True           # (1M contigs: streams A and B of rank 0 are as long as each other)
    assert multigpu.best_step(1_000_000, 6_666_666, 2)["scheme"] == "key_split"          # (two ranks: half the plane in sparse form is cheaper than the entry blocks)
    one = multigpu.best_step(1_000_000, 6_666_666, 1)
    assert one["scheme"] == "replicate" and 9 < one["step_ms"] < 12
    assert multigpu.best_step(5_000_000, 33_333_333, 8)["step_ms"] < multigpu.best_step(5_000_000, 33_333_333, 4)["step_ms"] < one["step_ms"] * 5


def test_entry_count_scheme_in_the_step_model():
    """shard_counts (reads sharded, partial counts of the DB's entries exchanged): what best_step picks on 8 GPUs for both sample sizes (round 6:
    its one-GPU and gloo rehearsals are as good as the other schemes'; `bench.py --gpus N` measures every scheme anyway), shorter than every
    scheme that moves planes, with rank 0 left to stage 04 -- and still short of 6 x"""
    for nc, nr in ((1_000_000, 6_666_666), (5_000_000, 33_333_333)):
        new = multigpu.best_step(nc, nr, 8)
        assert new["scheme"] == "shard_counts"
        planes = min((multigpu.step_model(nc, nr, 8, s, r0) for s, r0 in (("replicate", True), ("key_split", True), ("shard_reads", True), ("shard_reads", False))),
                     key=lambda c: c["step_ms"])
        assert new["step_ms"] < planes["step_ms"]
        one = multigpu.step_model(nc, nr, 1)["step_ms"]
        assert 2.5 < one / new["step_ms"] < 6.0                              # (the count launch's fixed part, measured in round 6: 1.25 ms per rank)
    assert "shard_counts" in multigpu.phase_a_model(6_666_666, 8, entry_counts=True)["ms"]
    assert "shard_counts" not in multigpu.phase_a_model(6_666_666, 8)["ms"]


def test_bench_scheme_plan_covers_every_scheme_for_the_rank_count():
    """bench.py --gpus N without a forced scheme measures each of these in a child process per rank (bench.run_all_schemes)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_main", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    labels = lambda w: [l for l, _ in mod.scheme_plan(w)]
    assert labels(2) == ["weak", "replicate", "key_split", "shard_reads", "shard_counts"]
    assert labels(8) == ["weak", "replicate", "key_split", "shard_reads", "shard_reads, rank 0 idle in Phase A", "shard_counts", "shard_counts, rank 0 idle in Phase A"]
    assert "key_split" not in labels(3) and "shard_counts, rank 0 idle in Phase A" in labels(3)
    for _, env in mod.scheme_plan(8)[1:]:
        assert env["PALACE_BENCH_LEG"] == "strong" and env["PALACE_BENCH_SCHEME"] in ("replicate", "key_split", "shard_reads", "shard_counts") and env["PALACE_BENCH_RANK0_READS"] in ("0", "1")


def test_split_by_weight_properties():
    rng = np.random.Generator(np.random.PCG64(1))
    w = rng.integers(1, 100, size=1000)
    for world in (1, 2, 3, 8):
        cuts = [multigpu.split_by_weight(w, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(w)
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        sums = [w[a:b].sum() for a, b in cuts]
        assert max(sums) - min(sums) <= 2 * w.max()
    assert multigpu.split_by_weight([], 0, 4) == (0, 0)
