"""Exchange steps of the hot path when one sample is sharded over the GPUs of a node
(one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  SURVEY.md section 8(e):

  eref Phase A   reads sharded by record range, private count table per rank; merge = every rank
                 owns 1/world of the key space: all_to_all of plane slices (each rank sends slice j of
                 its three planes straight to rank j -- all 7 xGMI links busy at once, 3 * 512/world
                 MiB per link), saturating bit-plane add of the world parts on the owner, then
                 all_gather of the merged ">= 3" plane (the only one Phase B reads).
  eref Phase B   refs sharded by cumulative length; all_gather of the per-ref rows (16 B each).
  generateGraph  records sharded by file ordinal; candidates (a few % of records) all_gathered with
                 a size pre-exchange, resolved identically on every rank; per-contig depth sums
                 all_reduced.
  matching       small after filtering: rank 0 (components are independent; no collective).

This module only moves bytes; the arithmetic on them is the HIP library's (merge_slices, resolve).
"""
from __future__ import annotations

import numpy as np


def split_by_weight(weights, rank: int, world: int):
    """Contiguous range [lo, hi) of items for `rank` with ~equal total weight per rank."""
    w = np.asarray(weights, dtype=np.float64)
    if world <= 1 or len(w) == 0:
        return 0, len(w)
    cum = np.cumsum(w)
    cuts = [int(np.searchsorted(cum, cum[-1] * r / world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, len(w)
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts[rank], cuts[rank + 1]


def key_buckets_of(rank: int, world: int):
    """The level-1 buckets (key >> 25, 0..127) rank `rank` of `world` counts when the key space is split (every rank holds all
    reads; palace_eref_set_key_buckets).  Canonical keys thin out linearly over the key space -- bucket b holds (255 - 2b) / 16384
    of them -- so every block of 2 * world buckets gives a rank its r-th bucket and its mirror: equal key mass, 128 / world
    buckets, i.e. 128 / world slices of 4 MiB of every plane.  world must divide 64."""
    assert 64 % world == 0 and 0 <= rank < world
    out = []
    for base in range(0, 128, 2 * world):
        out += [base + rank, base + 2 * world - 1 - rank]
    return sorted(out)


# ---- which Phase-A scheme for this many reads on this many GPUs ---------------------------------------------------------
# Constants measured on ONE MI355X with the current kernels, 1M-contig workload = 6.67 M reads x 150 bp (DESIGN.md section 6;
# tools/kr_diag.sh, profiles/): the count launch over all keys; the count launch of a 1/W key share of ALL reads
# (2.35 + 5.95 / W ms at the end of round 4 -- 8.28 / 5.44 / 3.80 / 3.07 ms for W = 1 / 2 / 4 / 8, tools/archive/r04z7.sh; round 3: 3.35 + 5.95 / W --:
# the key arithmetic over every read does not shard); the passes of the table exchange (pack the low
# plane, fold the parts on the owner); repacking gathered plane slices.  Everything else is interconnect arithmetic:
# xGMI is point to point, a rank reaches each peer over its own link, `link_gbs` is what one link and direction sustains.
# Round 5: the '>= 3' plane travels in SPARSE form (palace_eref_plane_pack / _unpack: a count per fine bucket + 2 B per set bit; the
# 1M-contig sample sets 24 M of the 2^32 bits = 3.6 per read, 48 MB instead of 512 MiB); pack / unpack passes: reading a rank's
# share twice, rewriting the other ranks' slices.
MODEL = dict(reads_measured=6_666_666, count_all_ms=8.2, key_fixed_ms=2.35, key_shared_ms=5.95, three_planes_factor=1.03,
             exchange_passes_ms=1.1, repack_ms=0.25, collective_latency_ms=0.05, link_gbs=50.0, plane_bytes=1 << 29,
             sparse_keys_per_read=3.6, sparse_pack_ms=0.1, sparse_unpack_ms=0.15,
             # shard_counts, MEASURED in round 6 on one GPU as rank 1 of W (tools/counts_share_diag.sh, profiles/r06b_counts_share.log: the count
             # launch of a 1/W share of the reads with every entry of the WHOLE DB's index probed inside, stage 04 beside it): 5.01 / 3.15 /
             # 2.28 / 2.13 ms for W = 2 / 4 / 7 / 8 = 1.25 + 7.3 / W -- the fixed part is what every rank does whatever its share (the DB's
             # 1.3 GB of entries read and tested against every fine bucket's slice, 65 536 workgroups, the partition kernels' tails); round
             # 5 had modelled it as 0.25 + 8.05 / W.  The sum pass over a rank's share of the W count blocks: 0.12 ms.
             count_fused_ms=7.3, entry_probe_fixed_ms=1.25, entry_hits_bytes=81.5e6, entry_sum_ms=0.12)


def phase_a_model(n_reads: int, world: int, link_gbs: float | None = None, sparse: bool = True, entry_counts: bool = False) -> dict:
    """Modelled milliseconds of eref Phase A (count launch + what it takes to have the '>= 3' plane complete on every rank) per
    step for the three schemes, and the cheapest.  n_reads: reads of the whole sample (both FASTQ sides).
      replicate    every rank counts all reads, nothing moves
      key_split    every rank holds all reads and counts its 1/W of the key space; all-gather of the plane slices
      shard_reads  reads sharded; two planes to their key-range owners, merge, all-gather of the merged '>= 3' plane"""
    m = dict(MODEL)
    if link_gbs:
        m["link_gbs"] = float(link_gbs)
    W, x = max(1, world), n_reads / m["reads_measured"]
    per_link_ms = lambda nbytes: nbytes / (m["link_gbs"] * 1e9) * 1e3 + m["collective_latency_ms"]
    out = {"replicate": m["count_all_ms"] * x}
    if W > 1:
        gather = per_link_ms(m["plane_bytes"] / W)                      # each rank pulls one slice per peer, all links at once
        gather_ks = gather + m["repack_ms"]                             # (key split: the slices of a rank's buckets are packed / put back)
        if sparse:                                                      # ... as counts + 16-bit keys, rebuilt on arrival
            sparse_bytes = 2.0 * m["sparse_keys_per_read"] * n_reads + 4 * 65536
            g_sparse = per_link_ms(sparse_bytes / W) + per_link_ms(4 * 65536 / W) + m["sparse_pack_ms"] + m["sparse_unpack_ms"] * (W - 1) / W
            gather, gather_ks = min(gather, g_sparse), min(gather_ks, g_sparse)
        if 64 % W == 0:
            out["key_split"] = (m["key_fixed_ms"] + m["key_shared_ms"] / W) * x + gather_ks
        out["shard_reads"] = (m["count_all_ms"] * m["three_planes_factor"] * x / W + per_link_ms(2 * m["plane_bytes"] / W)
                              + m["exchange_passes_ms"] + gather)
        if entry_counts:
            # reads sharded, partial counts of the DB's entries exchanged instead of planes: every rank probes the whole DB's entries
            # (fixed), sends (W - 1) / W of its count block -- one share per link --, sums its share, gathers the hit bits
            cb = 2.0 * m["entry_hits_bytes"]
            out["shard_counts"] = (m["count_fused_ms"] * x / W + m["entry_probe_fixed_ms"] + per_link_ms(cb / W) + m["entry_sum_ms"]
                                   + per_link_ms(m["entry_hits_bytes"] / W))
    choice = min(out, key=out.get)
    return dict(ms={k: round(v, 2) for k, v in out.items()}, choice=choice, link_gbs=m["link_gbs"], n_reads=int(n_reads), world=W, sparse_gather=bool(sparse),
                note="modelled from 1-GPU kernel times and per-link xGMI arithmetic; no N > 1 hardware measurement behind it")


# ---- the whole step of one rank, serial terms included ------------------------------------------------
# One-GPU stage times of the 1M-contig workload (bench.py stage_ms, round 5: Phase B 1.07 ms probing for itself) and how they scale: classify and resolve with the
# records (= reads), Phase B with the refs (a constant DB) over the ranks, stage 04 (selection + matching, on rank 0) with the
# contigs -- 1.4 ms alone on a device, 4.4 ms beside a count launch that saturates it (500k contigs: 2.7, 5M: 16-18, long: 1.5).
# Round 5: with the decomposition's phases on 2048 workgroups (the library's default; the one-GPU bench keeps 256, where the step is
# stream A's length and the shorter, denser burst costs the count launch more) stage 04 takes 2.75 ms beside a count launch at 1M
# contigs, 0.8 ms alone -- on N GPUs rank 0's stream B is the longer stream once Phase A is sharded, so it runs wide there.
STEP = dict(reset_ms=0.1, phase_b_fixed_ms=0.15, phase_b_ms=0.92, phase_b_counts_fixed_ms=0.25, phase_b_counts_ms=0.58, phase_b_counts_dense_fixed_ms=1.3, phase_b_counts_dense_ms=2.8, classify_ms=0.45, resolve_ms=0.37, small_collective_ms=0.1,
            stage04_alone_ms=(0.5, 0.3), stage04_beside_count_ms=(1.0, 3.4), stage04_beside_count_wide_ms=(0.6, 2.15))   # (fixed, per 1M contigs)


def xr_dense(n_reads: int) -> bool:
    """more key instances than 0.9 x 2^32 table slots: the scan takes its two-stage pruning (eref_scan.hip, palace_eref_scan_refs_indexed)"""
    return 3.0 * n_reads * 119 > 0.9 * 4294967296.0


def step_model(n_contigs: int, n_reads: int, world: int, scheme: str | None = None, rank0_counts: bool = True, link_gbs: float | None = None) -> dict:
    """Modelled milliseconds per step of an N-GPU run: stream A of every rank (reset, Phase A under `scheme` -- default: the
    cheapest --, Phase B on 1/W of the refs, row gather) against stream B of rank 0 (classify on 1/W of the records, candidate
    gather, resolve, depth reduce, stage 04); the step is the longer of the two.  rank0_counts=False: the reads are sharded over
    ranks 1 .. W-1 only (scheme shard_reads), so that stage 04 has rank 0's device to itself."""
    W = max(1, world)
    pa = phase_a_model(n_reads, W, link_gbs, entry_counts=scheme == "shard_counts")
    scheme = scheme or pa["choice"]
    a_phase = pa["ms"].get(scheme, pa["ms"]["replicate"])            # (a scheme forced where the model has none for it: one rank, W not dividing 64)
    if not rank0_counts and scheme == "shard_reads" and W > 2:
        m, x = MODEL, n_reads / MODEL["reads_measured"]
        a_phase += m["count_all_ms"] * m["three_planes_factor"] * x * (1.0 / (W - 1) - 1.0 / W)
    if not rank0_counts and scheme == "shard_counts" and W > 2:
        m, x = MODEL, n_reads / MODEL["reads_measured"]
        a_phase += m["count_fused_ms"] * x * (1.0 / (W - 1) - 1.0 / W)
    xc, xr, t = n_contigs / 1e6, n_reads / MODEL["reads_measured"], STEP
    coll = t["small_collective_ms"] if W > 1 else 0.0
    if scheme == "shard_counts":     # no reset (no plane is written), no probe kernel: the sentinel scatter over the whole DB + need / gather / window on 1 / W of the refs
        dense = xr_dense(n_reads)        # (a table most of whose slots are taken: half of the DB's sentinels hit and are scattered on EVERY rank, every ref is gathered)
        stream_a = a_phase + t["phase_b_counts_dense_fixed_ms" if dense else "phase_b_counts_fixed_ms"] + t["phase_b_counts_dense_ms" if dense else "phase_b_counts_ms"] / W + coll
    else:
        stream_a = t["reset_ms"] + a_phase + t["phase_b_fixed_ms"] + t["phase_b_ms"] / W + coll
    s04 = t["stage04_beside_count_ms"] if W == 1 else t["stage04_beside_count_wide_ms"] if rank0_counts else t["stage04_alone_ms"]
    stream_b = t["classify_ms"] * xr / W + coll + t["resolve_ms"] * xr + coll + s04[0] + s04[1] * xc
    return dict(scheme=scheme, rank0_counts=bool(rank0_counts), stream_a_ms=round(stream_a, 2), stream_b_rank0_ms=round(stream_b, 2),
                step_ms=round(max(stream_a, stream_b), 2))


def best_step(n_contigs: int, n_reads: int, world: int, link_gbs: float | None = None) -> dict:
    """scheme and rank-0 read share with the shortest modelled step"""
    cands = [step_model(n_contigs, n_reads, world, s, True, link_gbs) for s in phase_a_model(n_reads, world, link_gbs, entry_counts=True)["ms"]]
    if world > 2:                                  # the read-sharded schemes with rank 0 taking no reads
        cands += [step_model(n_contigs, n_reads, world, s, False, link_gbs) for s in ("shard_reads", "shard_counts")]
    return min(cands, key=lambda c: c["step_ms"])


class Exchange:
    def __init__(self, torch, dist, rank: int, world: int):
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self._recv = None
        self._recv_counts = None

    # ---- eref count table ------------------------------------------------------------------------
    def merge_planes(self, planes, merge_fn, pack_fn=None, final_gather=None):
        """planes: three 1-D uint8 tensors (this rank's partial planes, equal length B with
        B % (16 * world) == 0).  merge_fn(parts, n_parts, slice_off, slice_bytes) must fold
        parts[plane][part][slice] into the planes at slice_off (the HIP library in production).
        With pack_fn (-> 1-D uint8 tensor of length B: the low bit of every key's count, p1 ^ p2 ^ p3) only TWO
        planes travel -- (low bit, count >= 2) carry everything the three unary planes do -- and merge_fn is called
        with packed=True on parts laid out [2][part][slice].
        On return planes[2] is the global '>= 3' plane on every rank: by an all_gather of the owners' slices, or by
        final_gather() when given (the sparse form: gather_buckets_sparse with contiguous bucket ranges)."""
        torch, dist, W = self.torch, self.dist, self.world
        B = planes[0].numel()
        assert B % (16 * W) == 0 and all(p.numel() == B for p in planes)
        S = B // W
        send = planes if pack_fn is None else [pack_fn(), planes[1]]
        n_pl = len(send)
        if self._recv is None or self._recv.numel() != n_pl * B:
            self._recv = torch.empty((n_pl, W, S), dtype=torch.uint8, device=planes[0].device)
        # one all_to_all per plane: a plane already is [peer][slice], so it is its own send buffer, and a call moves
        # 512 MiB per rank.  (Round 1 saw wrong data from ONE call of 1.5 GiB in a world-size-1 rehearsal; the cause
        # was never isolated -- an element count or byte offset above 2^31 in that call is as likely as a library
        # fault -- so nothing is claimed about RCCL here: calls of this size have been checked, larger ones have not.)
        for p in range(n_pl):
            dist.all_to_all_single(self._recv[p].view(-1), send[p])
        if pack_fn is None:
            merge_fn(self._recv, W, self.rank * S, S)
        else:
            merge_fn(self._recv, W, self.rank * S, S, packed=True)
        if final_gather is not None:
            final_gather()
            return
        mine = planes[2][self.rank * S:(self.rank + 1) * S].clone()
        dist.all_gather_into_tensor(planes[2], mine)

    def merge_entry_counts(self, counts, hits, sum_fn):
        """The ranks counted shares of the READS and hold partial counts of the DB's probe-index entries (include/palace_hip.h,
        palace_eref_entry_layout): counts = this rank's block, 1-D uint8 of B bytes (16 bits per vector of eight entries, two bits
        per entry; B % (512 * world) == 0), hits = the hit-bit block, 1-D uint8 of B / 2 bytes.  One all_to_all of B / world per
        peer; sum_fn(parts, n_parts, part_stride, off, nbytes) sums the world parts of THIS rank's share of the block (bytes
        [off, off + nbytes); part p at parts + p * part_stride) into its share of `hits` (the HIP library in production); one
        all_gather of the shares.  On return `hits` is whole on every rank.  No plane crosses a link."""
        torch, dist, W = self.torch, self.dist, self.world
        B = counts.numel()
        assert B % (512 * W) == 0 and hits.numel() * 2 == B
        S = B // W
        if self._recv_counts is None or self._recv_counts.numel() != B:
            self._recv_counts = torch.empty(B, dtype=torch.uint8, device=counts.device)
        dist.all_to_all_single(self._recv_counts, counts)
        sum_fn(self._recv_counts, W, S, self.rank * S, S)
        mine = hits[self.rank * S // 2:(self.rank + 1) * S // 2].clone()
        dist.all_gather_into_tensor(hits, mine)

    def gather_key_buckets(self, plane):
        """plane: 1-D uint8 tensor of 2^29 bytes (the '>= 3' plane) of which this rank holds the 4 MiB slices of its buckets
        (key_buckets_of); afterwards every rank holds all of it.  One all_gather of 512 MiB / world per rank: the slices are
        packed into one buffer, gathered, and put back in place."""
        torch, dist, W = self.torch, self.dist, self.world
        rows = plane.view(128, -1)
        idx = [torch.tensor(key_buckets_of(r, W), device=plane.device) for r in range(W)]
        mine = rows.index_select(0, idx[self.rank]).contiguous()
        if self._recv is None or self._recv.numel() != plane.numel():
            self._recv = torch.empty(plane.numel(), dtype=torch.uint8, device=plane.device)
        allp = self._recv.view(W, 128 // W, -1)
        dist.all_gather_into_tensor(allp.view(-1), mine.view(-1))
        for r in range(W):
            if r != self.rank:
                rows.index_copy_(0, idx[r], allp[r])

    def gather_buckets_sparse(self, bucket_lists, pack_fn, unpack_fn, cap_keys: int, bufs: dict):
        """The '>= 3' plane completed on every rank without plane slices crossing the links: rank r holds the level-1 buckets
        bucket_lists[r] (equal counts); pack_fn(counts, keys, first) fills this rank's sparse form (counts: int32 [512 * n_b],
        keys: int16 [cap_keys], first: int64 [512 * n_b + 1] scratch), both are all-gathered, and unpack_fn(r, counts_r, keys_r,
        first) rebuilds rank r's buckets here.  Nothing is read back: returns the (world, 512 * n_b) counts tensor -- the caller
        checks, when the step's results are in, that no row sums to more than cap_keys (a rank whose keys were cut off: redo with
        more room, or with gather_key_buckets)."""
        torch, dist, W = self.torch, self.dist, self.world
        n_b = len(bucket_lists[0])
        assert all(len(b) == n_b for b in bucket_lists)
        n_fine = 512 * n_b
        dev = bufs["device"]
        if bufs.get("cap") != cap_keys or bufs.get("n_fine") != n_fine:
            bufs.update(cap=cap_keys, n_fine=n_fine,
                        counts=torch.zeros(n_fine, dtype=torch.int32, device=dev), keys=torch.zeros(cap_keys, dtype=torch.int16, device=dev),
                        first=torch.zeros(n_fine + 1, dtype=torch.int64, device=dev),
                        counts_all=torch.zeros((W, n_fine), dtype=torch.int32, device=dev),
                        keys_all=torch.zeros((W, cap_keys), dtype=torch.int16, device=dev))
        pack_fn(bufs["counts"], bufs["keys"], bufs["first"])
        dist.all_gather_into_tensor(bufs["counts_all"].view(-1), bufs["counts"])
        dist.all_gather_into_tensor(bufs["keys_all"].view(torch.uint8).view(-1), bufs["keys"].view(torch.uint8))      # (as bytes: gloo has no int16)
        for r in range(W):
            if r != self.rank:
                unpack_fn(r, bufs["counts_all"][r], bufs["keys_all"][r], bufs["first"])
        return bufs["counts_all"]

    # ---- small tables ------------------------------------------------------------------------------
    def gather_ranges(self, table, ranges):
        """table: (n, k) tensor, rank r filled rows ranges[r] = (lo, hi); afterwards all rows everywhere."""
        torch, dist, W = self.torch, self.dist, self.world
        width = max(hi - lo for lo, hi in ranges)
        lo, hi = ranges[self.rank]
        part = torch.zeros((width,) + tuple(table.shape[1:]), dtype=table.dtype, device=table.device)
        part[: hi - lo] = table[lo:hi]
        out = torch.empty((W, width) + tuple(table.shape[1:]), dtype=table.dtype, device=table.device)
        dist.all_gather_into_tensor(out.view(-1), part.view(-1))
        for r, (a, b) in enumerate(ranges):
            table[a:b] = out[r, : b - a]

    def gather_varlen(self, rows, n: int):
        """rows: (cap, width) tensor of which the first n rows are valid on this rank.  Returns a
        contiguous (total, width) tensor holding every rank's rows in rank order, and `total`."""
        torch, dist, W = self.torch, self.dist, self.world
        counts = torch.zeros(W, dtype=torch.int64, device=rows.device)
        mine = torch.tensor([n], dtype=torch.int64, device=rows.device)
        dist.all_gather_into_tensor(counts, mine)
        counts = [int(x) for x in counts.tolist()]
        self.last_counts = counts
        width = max(1, max(counts))
        part = torch.zeros((width, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        part[:n] = rows[:n]
        out = torch.empty((W, width, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        dist.all_gather_into_tensor(out.view(-1), part.view(-1))
        allrows = torch.cat([out[r, :c] for r, c in enumerate(counts)], dim=0).contiguous()
        return allrows, sum(counts)

    def gather_padded(self, rows, n_dev, width: int, out=None):
        """The same gather with nothing read back: rows (cap, k), of which the first *n_dev (a 1-element int64 device tensor)
        are valid, are gathered as `width` rows per rank -- rows beyond a rank's count arrive as zeros (a zero candidate is
        one that resolve ignores).  Returns ((world * width, k) tensor, (world,) int64 device tensor of the counts); the
        caller checks counts <= width afterwards (a count above it means rows were cut off: redo with a wider gather)."""
        torch, dist, W = self.torch, self.dist, self.world
        counts = torch.empty(W, dtype=torch.int64, device=rows.device)
        dist.all_gather_into_tensor(counts, n_dev)
        part = rows[:width] * (torch.arange(width, device=rows.device) < n_dev).to(rows.dtype)[:, None]
        if out is None or out.shape[0] != W * width:
            out = torch.empty((W * width, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        dist.all_gather_into_tensor(out.view(-1), part.contiguous().view(-1))
        return out, counts

    def reduce_sum(self, t):
        self.dist.all_reduce(t)
        return t
