"""One measured configuration of the bench: the HBM-resident step on two streams (eref on A; generateGraph -> stage 04 on B),
its timing, the JSON line's objects (`roofline`, `roofline_stages`, `e2e`, `cpu_baseline`) and the run's own cross-checks."""
import ctypes
import hashlib
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .cpu_baseline import cpu_baseline
from .e2e import graph_text, run_e2e, write_e2e_inputs
from .sample import HBM_PEAK_GBS, READ_LEN, SEED, make_graph_sample, make_sample, make_side_inputs, progress


def roofline_stages(stages, traffic):
    """{stage: (algorithmic bytes per step, live ms per step, what the bytes are)} -> the per-stage roofline objects"""
    out = {}
    for k, (alg, ms, what) in stages.items():
        ach = alg / (ms * 1e-3) / 1e9 if ms and ms > 0 else None
        out[k] = {"bound": "hbm", "algorithmic_bytes_per_step": int(alg), "ms_per_step": float(ms), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": None if ach is None else ach / HBM_PEAK_GBS, "traffic": traffic.get(k), "bytes": what}
    return out


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
    if int(t.get("fused_probe", 0)) != int(fused):
        return None, f"profiles/phase_a_traffic.json was measured with Phase B's look-ups {'inside' if int(t.get('fused_probe', 0)) else 'outside'} the count launch; this run has them {'inside' if int(fused) else 'outside'}", {}
    if t.get("build") != version:
        return None, f"profiles/phase_a_traffic.json was measured on another build ({t.get('build')}); this is {version}", {}
    return t.get("bytes_per_launch"), t.get("source"), {k: v.get("bytes") for k, v in (t.get("stages") or {}).items()}


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
    from palace_amd import capi, coder, multigpu, stage04_io       # (oracle/ is imported by the cpu_baseline leg only)

    hdr = coder.header_from_picks(np.random.Generator(np.random.PCG64(SEED)).integers(0, 6, size=32))
    # Streams.  A: eref (count + scan).  B: generateGraph (classify, resolve, copy numbers) and stage 04 (selection + matching: ~150
    # small latency-bound launches), high priority.  Stage 04 beside the saturating counting kernels takes ~3.5 ms instead of the ~1.1 ms
    # it takes alone and costs the count launch ~0.6 ms; confining it to CUs of its own, holding its rounds back behind the partition
    # kernels or the count launch, capturing it as hipGraphs and a two-deep pipeline of stream B were all measured in rounds 3-5 and
    # none shortened the step (profiles/HISTORY.md); the knobs went in round 6.
    # With collectives (N GPUs) A and B are torch streams the contexts run on (palace_ctx_create_on_stream), so that
    # torch.distributed's collectives are stream-ordered with the library's kernels and nothing waits on the host.
    if collectives:
        sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
        ctx, ctx_g = capi.Ctx(local, stream=sA.cuda_stream), capi.Ctx(local, stream=sB.cuda_stream)
    else:
        ctx, ctx_g = capi.Ctx(local), capi.Ctx(local, high_priority=True)
    ctx_s = ctx_g
    # the decomposition's arc- and vertex-sized phases on 256 workgroups here (the library's default is 2048: stage 04 alone on the
    # device then takes 0.8 instead of 1.4 ms, beside the counting kernels 2.75 instead of 4.4 ms -- but as a shorter, denser burst of
    # random atomics it costs the count launch 1.05 ms instead of 0.7, and the step is stream A's length: 10.1 against 9.86 ms, A/B
    # on one box, tools/ab.sh r05d)
    if world == 1 or solo:                          # (N GPUs: rank 0's stream B is the longer stream once Phase A is sharded: the library's 2048)
        ctx_s.match_set_option("decomp_grid", 256)
    for opt in ("iters_per_round", "decomp_grid", "one_word_keys"):    # tuning runs only
        if os.environ.get("PALACE_OPT_" + opt.upper()):
            ctx_s.match_set_option(opt, int(os.environ["PALACE_OPT_" + opt.upper()]))
    # (a second eref context with a count table of its own, so that a step's counting kernels run beside the step before's Phase B --
    # --batches-in-flight 2 of rounds 3-5 -- bought 6 % while Phase B was 1.5 ms and distorted the count launch's own duration; with Phase B
    # inside the count launch there is 0.6 ms left to hide.  Removed in round 6.)
    for e in (ctx,):
        e.eref_set_coder(hdr)
        if os.environ.get("PALACE_OPT_SLAB_BASES"):       # tuning runs only
            e.eref_set_option("slab_bases", int(os.environ["PALACE_OPT_SLAB_BASES"]))
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
        if forced not in ("replicate", "key_split", "shard_reads", "shard_counts") or (forced == "key_split" and 64 % world):
            raise SystemExit(f"PALACE_BENCH_SCHEME={forced}: not a scheme for {world} ranks")
    if forced:
        scheme = forced
    if not collectives:
        scheme = "replicate"
    # tuning runs only (one GPU): PALACE_OPT_COUNTS_SHARE=W -- the step of rank 1 of W under shard_counts: 1 / W of the reads counted (partial
    # entry counts), 1 / W of the refs scanned, no peer; results are partial by design (what the model's constants are measured with)
    counts_share = int(os.environ.get("PALACE_OPT_COUNTS_SHARE", "0")) if (world == 1 and not solo and not collectives) else 0
    if counts_share > 1:
        scheme = "shard_counts"
    # "shard_counts" (end of round 5, PALACE_BENCH_SCHEME=shard_counts: opt-in until an N-GPU run has checked it): the reads are sharded
    # as under shard_reads, but what the ranks exchange is their partial COUNTS of the DB's probe-index entries (two bits per entry,
    # summed by entry range, hit bits all-gathered: include/palace_hip.h, palace_eref_entry_layout) -- no plane crosses a link
    shard_counts = scheme == "shard_counts"
    shard_reads = scheme == "shard_reads"                   # (the plane exchange)
    reads_sharded = shard_reads or shard_counts
    # Stage 04 runs on rank 0.  Beside a count launch that saturates the device it takes 4-5x what it takes alone and grows with
    # the sample (5M contigs: 27 ms), so for large samples under the read-sharded scheme rank 0 takes NO reads: ranks 1 .. W-1
    # count 1/(W-1) each, rank 0's device has stage 04 (and its share of everything else) to itself.  PALACE_BENCH_RANK0_READS=0|1 forces.
    rank0_counts = True
    if reads_sharded and world > 2:
        env0 = os.environ.get("PALACE_BENCH_RANK0_READS", "auto")
        rank0_counts = (env0 == "1") if env0 in ("0", "1") else (best["rank0_counts"] if scheme == best["scheme"] else
                                                                  multigpu.step_model(args.contigs, n_reads_total, world, scheme, False)["step_ms"] >=
                                                                  multigpu.step_model(args.contigs, n_reads_total, world, scheme, True)["step_ms"])
    read_weights = None if (rank0_counts or not reads_sharded) else [0.0] + [1.0] * (world - 1)
    model.update(choice_in_force=scheme, forced=bool(forced), rank0_counts=bool(rank0_counts), choice=best["scheme"],
                 step=multigpu.step_model(args.contigs, n_reads_total, world, scheme, rank0_counts),
                 step_alternatives=[multigpu.step_model(args.contigs, n_reads_total, world, sch, True) for sch in model["ms"]])
    if counts_share > 1:
        sample = make_sample(torch, dev, args.contigs, args.refs, 1, counts_share, long_mode, None)
    else:
        sample = make_sample(torch, dev, args.contigs, args.refs, rank if reads_sharded else 0, world if reads_sharded else 1, long_mode, read_weights)
    gs = make_graph_sample(torch, dev, args.contigs, sample["n_pairs_total"], rank, world, long_mode)
    if collectives:                                 # avgDepth is a pipeline input: computed once from all shards
        tot = torch.tensor([float(gs["col"]["ref_len"].sum().item())], device=dev, dtype=torch.float64)
        dist.all_reduce(tot)
        gs["avg_depth"] = float(f"{tot.item() / gs['lens'].sum():.6g}")
    torch.cuda.synchronize()
    if rank == 0: progress(f"{leg}: sample generated on the device")
    one_min, three_min = capi.window_minimums(0.9, 0.85)
    L = capi.lib()
    P = lambda t: t.data_ptr()
    n_side, n_refs, nt = sample["n_reads_side"], sample["n_refs"], args.contigs
    # refs shard by cumulative length across ranks (eref Phase B); every rank holds the whole (small) DB
    r_lo, r_hi = multigpu.split_by_weight(sample["ref_lens"], rank, world) if counts_share <= 1 else multigpu.split_by_weight(sample["ref_lens"], 1, counts_share)
    rows = torch.zeros((n_refs, 4), dtype=torch.int32, device=dev)
    rows_host = torch.zeros((n_refs, 4), dtype=torch.int32).pin_memory()
    cn_host = torch.zeros(args.contigs, dtype=torch.int32).pin_memory()
    consumed_host = torch.zeros(args.contigs, dtype=torch.int64).pin_memory()
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

        # The '>= 3' plane completed on every rank in SPARSE form (palace_eref_plane_pack / _unpack: a count per fine bucket and 2 B per
        # set bit -- 48 MB for the 1M-contig sample instead of 512 MiB of plane slices).  The room for a rank's keys is what the
        # first step (dense gather) found, + 1/4; the counts come back with the step's results and are checked then.
        sparse = {"on": world > 1 and 128 % world == 0 and os.environ.get("PALACE_BENCH_SPARSE_GATHER", "1") == "1", "cap": 0, "bufs": {"device": dev}, "learn": False}
        if os.environ.get("PALACE_BENCH_SPARSE_GATHER", "1") == "1" and world == 1:
            sparse["on"] = True                          # (one-rank rehearsals of the exchange paths: pack and count, nothing to unpack)
        split_buckets = [multigpu.key_buckets_of(r, world) for r in range(world)] if 64 % world == 0 else None
        slice_buckets = [list(range(r * 128 // world, (r + 1) * 128 // world)) for r in range(world)] if 128 % world == 0 else None
        n_fine_rank = 65536 // world if 128 % world == 0 else 0
        if sparse["on"]:
            sparse.update(probe_counts=torch.zeros(max(1, n_fine_rank), dtype=torch.int32, device=dev), probe_keys=torch.zeros(8, dtype=torch.int16, device=dev),
                          probe_first=torch.zeros(n_fine_rank + 1, dtype=torch.int64, device=dev),
                          need_host=torch.zeros(1, dtype=torch.int64).pin_memory(), sums_host=torch.zeros(world, dtype=torch.int64).pin_memory())

        def sparse_gather(bucket_lists):
            counts_all = exch.gather_buckets_sparse(
                bucket_lists, lambda c, k, f: ctx.eref_plane_pack(bucket_lists[rank], P(c), P(k), sparse["cap"], P(f)),
                lambda r, c, k, f: ctx.eref_plane_unpack(bucket_lists[r], P(c), P(k), sparse["cap"], P(f)), sparse["cap"], sparse["bufs"])
            sparse["sums_host"].copy_(counts_all.sum(dim=1, dtype=torch.int64), non_blocking=True)
            sparse["steps"] = sparse.get("steps", 0) + 1
    last = {}
    seen = {"graph": set(), "rows": set(), "steps": 0}     # result digests of the untimed steps (warm-up, soak): one value each, or the step is not repeatable
    h_last = {}
    host_ms = {}
    ref_off_local = sample["ref_off"][r_lo:r_hi + 1].contiguous()
    # Per-DB probe index of this rank's refs, built once outside the timed region: the reference, too, scans a
    # DB through the index file it built on first use (<fasta>.k32.index.dat), and the CPU baseline below is
    # timed with its index prebuilt as well.
    probe_index = ctypes.c_void_p()
    if shard_counts:                                # every rank probes the WHOLE DB's entries (the ranks' partial counts line up entry by entry)
        ref_off_local = sample["ref_off"].contiguous()
        capi._check(L.palace_eref_probe_index_build(ctx.h, P(sample["ref_bases"]), P(ref_off_local), n_refs,
                                                    sample["ref_total"], ctypes.byref(probe_index)), "probe index")
    else:
        capi._check(L.palace_eref_probe_index_build(ctx.h, P(sample["ref_bases"]), P(ref_off_local), r_hi - r_lo,
                                                    sample["ref_total"], ctypes.byref(probe_index)), "probe index")
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
    final_count = not shard_reads
    # key split: rank r counts only the keys of ITS 1/W of the key space (they are dropped where they are made: the partition
    # kernels move 1/W of the bytes, the key arithmetic stays) and the ">= 3" plane slices are all-gathered -- one collective of
    # 512 MiB / W per rank instead of the table exchange
    key_split = bool(exch) and scheme == "key_split"
    # The count launch of a step is its final count, so ALL of Phase B's look-ups ride along in the count kernel (the index's four entry
    # sets tested against each fine bucket's ">= 3" slice while it is in LDS: palace_eref_attach_probe_index + option probe_all_sets) and
    # the plane is never written -- no slice write-back, no probe kernel, no reset before the next step.  Not for the ranks of a key
    # split or a plane exchange: whoever exchanges plane slices afterwards needs the plane, and probes it in the scan.
    fused_all = final_count and not key_split and not shard_reads and not shard_counts
    fused_probe = fused_all or shard_counts
    fused_mode = 2 if fused_all else 0                     # (the field profiles/phase_a_traffic.json is keyed by)
    if fused_probe:
        capi._check(L.palace_eref_attach_probe_index(ctx.h, probe_index), "attach probe index")
    if fused_all:
        capi._check(L.palace_eref_set_option(ctx.h, b"probe_all_sets", 1), "probe_all_sets")
    ec = None
    if shard_counts:                                # the two blocks in torch-owned memory (what the collectives address); an idle rank's counts stay zero
        cb, hb = ctx.eref_entry_layout(probe_index)
        assert cb % (512 * world) == 0, "the count block does not split evenly over this many ranks"
        ec = {"counts": torch.zeros(cb, dtype=torch.uint8, device=dev), "hits": torch.zeros(hb, dtype=torch.uint8, device=dev),
              "keys_total": 3 * 2 * sample["n_pairs_total"] * READ_LEN}
        torch.cuda.synchronize()
        ctx.eref_entry_buffers_attach(probe_index, P(ec["counts"]), P(ec["hits"]))
        capi._check(L.palace_eref_set_option(ctx.h, b"probe_all_sets", 2), "probe_all_sets")
        capi._check(L.palace_eref_set_option(ctx.h, b"scan_ref_lo", r_lo), "scan_ref_lo")
        capi._check(L.palace_eref_set_option(ctx.h, b"scan_ref_hi", r_hi), "scan_ref_hi")
    if key_split:
        ctx.eref_set_key_buckets(multigpu.key_buckets_of(rank, world))       # mirrored pairs of buckets: equal key mass per rank
    ctx.eref_set_option("final_count", 1 if final_count else 0)
    seq = {}

    # Timing marks.  One GPU, one batch in flight: the marks of a step are read at the step's end -- every stream has drained by then, the
    # reads do not wait -- and the next step but one reuses them: 16 marks per context in all.  (With a mark of its own per step and
    # stream the enqueueing thread's turn-around between two steps grew with the number of events alive: 0.36 ms at 30 steps, 0.88 at
    # 100 -- time the instrumentation cost the steps it measures.)  Everything else keeps a mark per step, read after the timed region.
    harvest = not collectives and not os.environ.get("PALACE_BENCH_DIAG_SKIP") and not os.environ.get("PALACE_BENCH_SKIP_EREF")
    acc = {k: [] for k in ("count", "merge", "scan", "between", "classify", "resolve", "stage04")}

    def step(i, timed):
        m = 8 * (i & 1) if harvest else 8 * i
        tot_b = n_side * READ_LEN
        # ---------------- eref: runs asynchronously on its own stream ----------------
        def eref_head():
            capi._check(L.palace_eref_table_reset(ctx.h), "reset")
            if timed: ctx.mark(m)
            # both FASTQ sides as one read set: one binning pass, the plane slices are loaded and stored once
            if packed:
                capi._check(L.palace_eref_count_reads_packed(ctx.h, *(P(t) for t in packed), 2 * tot_b, 2 * n_side), "count")
            else:
                capi._check(L.palace_eref_count_reads(ctx.h, P(sample["r12"]), P(sample["read_off"]), 2 * n_side, None, 2 * tot_b), "count")
            if timed: ctx.mark(m + 1)

        skip_eref = os.environ.get("PALACE_BENCH_SKIP_EREF") == "1"      # tuning runs only: stream B alone on the device
        if not skip_eref:
            eref_head()                                # launched first: generateGraph + matching (and, on N GPUs, their small collectives at
                                                       # RCCL's high-priority stream) overlap the counting kernels

        def eref_tail():
            if exch and shard_counts:                  # the ranks' partial entry counts: all-to-all by entry range, sum, all-gather of the hit bits
                with on_a():
                    exch.merge_entry_counts(ec["counts"], ec["hits"],
                                            lambda parts, n, stride, off, nb: ctx.eref_entry_hits_from_counts(probe_index, P(parts), n, stride, off, nb))
                ctx.eref_entry_hits_complete(probe_index, ec["keys_total"])
            elif shard_counts:                         # (one rank, rehearsal: its counts are the sample's)
                ctx.eref_entry_hits_from_counts(probe_index, P(ec["counts"]), 1, ec["counts"].numel(), 0, ec["counts"].numel())
                ctx.eref_entry_hits_complete(probe_index, ec["keys_total"])
            if exch and shard_reads:                   # count-table exchange (RCCL) on stream A, then Phase B on this rank's refs
                with on_a():
                    exch.merge_planes(planes, merge_fn, pack_fn, final_gather=(lambda: sparse_gather(slice_buckets)) if sparse["cap"] else None)
            elif key_split:                            # every rank counted its range of the key space: gather the ">= 3" plane
                with on_a():
                    if sparse["cap"]:
                        sparse_gather(split_buckets)
                    else:
                        exch.gather_key_buckets(planes[2])
            if exch and (shard_reads or key_split) and sparse["on"] and not sparse["cap"]:
                # the first step took the dense gather: how many keys the sparse form of a rank's share holds sizes the later steps'
                ctx.eref_plane_pack((slice_buckets if shard_reads else split_buckets)[rank], P(sparse["probe_counts"]), P(sparse["probe_keys"]), 0, P(sparse["probe_first"]))
                sparse["learn"] = True
            if timed: ctx.mark(m + 2)
            if shard_counts:                           # the whole DB's index, this rank's range of the refs (options scan_ref_lo / _hi)
                # (the rows of the other ranks' refs come out as zero rows here and are overwritten by the row gather below)
                capi._check(L.palace_eref_scan_refs_indexed(ctx.h, probe_index, P(sample["ref_bases"]), P(ref_off_local), n_refs,
                                                            sample["ref_total"], one_min, three_min, P(rows)), "scan")
            else:
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
        if timed: g.mark(m + 2); ctx_s.mark(m + 2)
        if stage04 is not None:                     # rank 0 owns the (small) stage; its result is what the sample's all_result holds
            # (diagnosis only, timed steps only -- the line then fails its own checks on purpose: PALACE_BENCH_DIAG_SKIP=stage04|match
            # leaves stage 04 / its matching rounds out, to see what they cost the counting kernels beside them)
            diag_skip = os.environ.get("PALACE_BENCH_DIAG_SKIP") if timed else None
            if diag_skip != "stage04":
                stage04.filter(P(e_buf), P(n_edges_dev), max(1, n_cands))
            if diag_skip is None:
                stage04.match(P(e_buf), P(cn_dev), 10, False, True)
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
                capi._check(L.palace_d2h(g.h, consumed_host.data_ptr(), P(consumed), nt * 8), "d2h")
                h_last.update(edges=h_edges, cn=cn_host.numpy().copy(), consumed=consumed_host.numpy().copy())
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
        finish_graph()
        # ---------------- join: eref results to the host ----------------
        capi._check(L.palace_d2h_async(ctx.h, rows_host.data_ptr(), P(rows), rows.numel() * 4), "d2h")
        ctx.mark(4094)
        ctx.mark_wait(4094)
        if timed and harvest:
            acc["count"].append(ctx.mark_elapsed(m, m + 1)); acc["merge"].append(ctx.mark_elapsed(m + 1, m + 2)); acc["scan"].append(ctx.mark_elapsed(m + 2, m + 3))
            if i > 0: acc["between"].append(ctx.mark_elapsed(8 * ((i - 1) & 1) + 3, m))
            acc["classify"].append(ctx_g.mark_elapsed(m, m + 1)); acc["resolve"].append(ctx_g.mark_elapsed(m + 1, m + 2))
            acc["stage04"].append(ctx_s.mark_elapsed(m + 2, m + 3))
        if not timed:
            seen["rows"].add(hashlib.sha256(rows_host.numpy().tobytes()).hexdigest()[:16])
            seen["steps"] += 1
        if exch:
            g.mark(4093)
            g.mark_wait(4093)                          # stream B has drained on every rank (only rank 0 waited for a stage-04 result)
            if sparse["on"] and (shard_reads or key_split):
                ctx.mark(4090); ctx.mark_wait(4090)                 # stream A has drained (its results were waited for above; the sums travel behind them)
                if sparse["learn"]:                                # the dense step: the largest share in sparse form, over the ranks, sizes the room
                    need = sparse["probe_first"][-1:].clone()
                    dist.all_reduce(need, op=dist.ReduceOp.MAX)
                    sparse["cap"] = (int(need.item()) * 5 // 4 + 65536) // 65536 * 65536
                    sparse["learn"] = False
                elif sparse["cap"] and int(sparse["sums_host"].max().item()) > sparse["cap"]:
                    sparse["cap"] = 0                              # a rank's keys were cut off: this step again with the dense gather (which sizes the room anew)
                    return step(i, timed)
            cnt = exch.last_counts if gat["learn"] else [int(x) for x in counts_host.tolist()]
            if not gat["learn"] and max(cnt) > gat["width"]:
                gat["width"] = 0                       # a rank had more candidates than the padded gather carried: this step again, exactly
                return step(i, timed)
            last["n_cands"] = int(sum(cnt))
            gat["width"] = max(gat["width"], (max(cnt) + max(cnt) // 8 + 256) // 256 * 256)

    def barrier():
        ctx.sync()
        ctx_g.sync()
        ctx_s.sync()
        torch.cuda.synchronize()
        if sync_dist is not None:
            sync_dist.barrier()
            torch.cuda.synchronize()

    torch.cuda.synchronize()                         # every buffer torch made above is filled before a library stream touches it
    if rank == 0: progress(f"{leg}: resident inputs ready; warm-up")
    for _ in range(args.warmup):
        step(0, False)
    barrier()
    # The sample's host side is millions of Python objects (a name and a score text per contig): a full collection of the cyclic
    # garbage collector walks them all -- one 60 ms pause somewhere in a 100-step run, 0.6 ms on the mean step.  They are long-lived:
    # frozen out of the collector's sight, and no collection inside the timed region.
    import gc
    gc.collect(); gc.freeze(); gc.disable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if sync_world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        sync_dist.all_reduce(tmax, op=sync_dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_step = 1e3 * dt / args.steps
    if rank == 0: progress(f"{leg}: {args.steps} timed steps, {ms_step:.2f} ms each")
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
    if harvest:
        count_each, between_ms = acc["count"], (float(np.mean(acc["between"])) if acc["between"] else None)
        count_ms, merge_ms, scan_ms = np.mean(acc["count"]), np.mean(acc["merge"]), np.mean(acc["scan"])
        classify_ms, resolve_ms, stage04_ms = np.mean(acc["classify"]), np.mean(acc["resolve"]), np.mean(acc["stage04"])
    else:
        if os.environ.get("PALACE_BENCH_SKIP_EREF") == "1":
            count_each, count_ms, merge_ms, scan_ms, between_ms = [1.0], 1.0, 0.0, 0.0, None
        else:
            E = lambda i: ctx
            count_each = [E(i).mark_elapsed(8 * i, 8 * i + 1) for i in K]
            count_ms = np.mean(count_each)                                              # one launch per step (both FASTQ sides)
            merge_ms = np.mean([E(i).mark_elapsed(8 * i + 1, 8 * i + 2) for i in K])
            scan_ms = np.mean([E(i).mark_elapsed(8 * i + 2, 8 * i + 3) for i in K])
            between_ms = float(np.mean([E(i).mark_elapsed(8 * i + 3, 8 * (i + 1)) for i in range(args.steps - 1)])) if args.steps > 1 else None
        classify_ms = np.mean([ctx_g.mark_elapsed(8 * i, 8 * i + 1) for i in K])
        resolve_ms = np.mean([ctx_g.mark_elapsed(8 * i + 1, 8 * i + 2) for i in K])
        stage04_ms = np.mean([ctx_s.mark_elapsed(8 * i + 2, 8 * i + 3) for i in K])
    r = rows_host.numpy()
    reported = int(((r[:, 1] > 0) & (r[:, 1].astype(np.float32) / r[:, 2].astype(np.float32) > 0.75)).sum())

    failures, out = [], None
    if exch:
        model["sparse_gather"] = dict(steps=sparse.get("steps", 0), cap_keys_per_rank=sparse["cap"],
                                      note="steps whose '>= 3' plane was completed from counts + 16-bit keys (palace_eref_plane_pack / _unpack) "
                                           "instead of plane slices; the first step of a run takes the dense gather and sizes the room")
    if rank == 0:
        L.palace_version.restype = ctypes.c_char_p
        version = L.palace_version().decode()
        fused_now = fused_all
        traffic, traffic_src, stage_traffic = profiled_traffic(args, world, version, fused_mode)
        alg_bytes = (READ_LEN + 6 * (READ_LEN - 31)) * 2 * n_side        # per launch (both FASTQ sides of this rank)
        # when Phase B's channel-0 probe rides along in the count kernel, its look-ups (1 B per ref position) are work of this launch
        fused_sets = 0 if not fused_now else (3 if fused_all else 1)         # channels of Phase B's look-ups the count launch does
        probe_bytes = fused_sets * sum(int(l) - 31 for l in sample["ref_lens"][r_lo:r_hi])
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
                       "reads": ("packed in HBM: two bits per base + 32-mer start mask, 0.375 B/base (palace_eref_count_reads_packed)" if packed else
                                 "ASCII in HBM, 1 B/base (palace_eref_count_reads)") + ("; count keeps only the '>= 3' plane (final_count)" if final_count else ""),
                       "parallelism": "1 GPU" if world == 1 else (f"reads/records/refs sharded over {world} GPUs (RCCL)" + ("; the ranks exchange partial counts of the DB's probe-index entries, no plane crosses a link" if shard_counts else "")
                                                                   + ("" if rank0_counts else f"; rank 0 takes no reads: stage 04 has its device to itself, ranks 1-{world - 1} count") if reads_sharded else
                                                                   f"records/refs and the key space sharded over {world} GPUs (RCCL): every GPU counts its 1/{world} of the keys of all reads, the '>= 3' plane is all-gathered" if key_split else
                                                                   f"records/refs sharded over {world} GPUs (RCCL), reads counted on every GPU"),
                       "parallelism_model": model,
                       "ref_index": "per-DB probe index prebuilt, as the reference's cached <fasta>.k32.index.dat (19.5 B/position in HBM: four entry sets of 2-byte entries, three position -> entry maps, sentinel positions)"
                                    + (("; all of its look-ups ride along in the count kernel, the '>= 3' plane is never written" if fused_all else
                                        "; its channel-0 probe rides along in the count kernel") if fused_probe and final_count and not key_split else ""),
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
            "roofline": {"bound": "hbm", "kernel": ("eref count_reads_packed (bin1 + bin2 + lds_count kernels of one launch" + ((", Phase B's look-ups of all three channels fused into lds_count, no plane written)" if fused_all else ", Phase B's channel-0 probe fused into lds_count)") if fused_now else ")")) if packed else
                                   "eref count_reads (streams + bin1 + bin2 + lds_count kernels of one launch)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # PMC (separate rocprofv3 passes, profiles/r01q_end_state_fused_launch.md): FETCH_SIZE x2 + WRITE_SIZE of
                         # bin1 + bin2 + lds_count per launch; only valid for the default workload on one GPU
                         "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "avg_launch_ms": count_ms, "algorithmic_bytes_per_launch": alg_bytes + probe_bytes,
                         # the same launch priced on Phase A's bytes alone (the look-ups it also does counted as nothing), and the whole of
                         # eref -- Phase A + Phase B's SURVEY bytes -- over the launch + the scan behind it: the two figures that do not
                         # depend on which kernel Phase B's look-ups are done in
                         "frac_phase_a_bytes_only": alg_bytes / (max(count_ms, 1e-6) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_eref_count_plus_scan": (alg_bytes + sum(int(l) + 3 * (int(l) - 31) for l in sample["ref_lens"][r_lo:r_hi]))
                                                      / (max(count_ms + scan_ms, 1e-6) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_note": f"864 B per 150-bp read x {2 * n_side} reads" + (f" + {probe_bytes} B: the look-ups of Phase B ({fused_sets} B per ref "
                                                   "position: " + ("all three channels" if fused_sets == 3 else "channel 0") + "), which this launch's count kernel does while a bucket's slice is in LDS" if probe_bytes else "")},
            # the other stages of the step against the same roofline (SURVEY.md section 8(d) algorithmic bytes; live event times of
            # this run; PMC traffic of the committed profile when it is of this build and workload)
            "roofline_stages": roofline_stages(dict(
                phase_b=(sum(int(l) + (3 - fused_sets) * (int(l) - 31) for l in sample["ref_lens"][r_lo:r_hi]), scan_ms,
                         "l + 3(l - 31) B per ref: a byte per base, three 1-byte look-ups per position" +
                         (" -- minus the look-ups the count launch did (" + ("all three channels" if fused_sets == 3 else "channel 0") + ")" if fused_now else "")),
                classify=(52 * gs["n"] + 64 * gs["n_sa"], classify_ms, "52 B per primary record + 64 B per SA item"),
                resolve=(64 * int(last.get("n_cands", 0)) + 16 * int(last.get("n_cands", 0)), resolve_ms,
                         "64 B per candidate read + 16 B per evidence written"),
                stage04=((32 * int(last.get("n_segs_filtered", 0)) + 24 * int(last.get("n_kept_junc", 0))) * 10, stage04_ms,
                         "32 B per SEG + 24 B per JUNC of the filtered graph, read once per pass, -i 10 passes")), stage_traffic),
            "library": version,
            # eref's unit is a read, generateGraph's a BAM record -- the same step in those units
            "rates": {"reads_per_s": 2 * sample["n_pairs_total"] / (ms_step * 1e-3), "bam_records_per_s": gs["n_total"] / (ms_step * 1e-3),
                      "read_bases_per_s": 2 * sample["n_pairs_total"] * READ_LEN / (ms_step * 1e-3)},
            "stage_ms": {"eref_count_each_step": [round(float(x), 3) for x in count_each], "eref_count_both_sides": count_ms, "eref_table_merge": merge_ms, "eref_scan_refs": scan_ms, "eref_stream_between_steps": between_ms, "eref_stream_between_steps_median": (float(np.median(acc["between"])) if harvest and acc["between"] else None),
                         "eref_stream_between_steps_max": (float(np.max(acc["between"])) if harvest and acc["between"] else None),
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
                # the eref rows (ordinal, n_intervals, el, ref_len per ref) of the resident step as it was timed -- tests/test_gpu_bench_workloads.py
                # holds them against the oracle's scan of every ref on the oracle's table of every read
                np.save(os.path.join(work, "eref_rows_resident_step.npy"), r.copy())
                with open(os.path.join(work, "eref_rows_resident_step.json"), "w") as fh:
                    json.dump({"fused_probe_mode": int(fused_mode), "probe_all_sets": bool(fused_all), "reads": args.reads, "final_count": bool(final_count)}, fh)
                progress("e2e: writing the sample's files")
                paths = write_e2e_inputs(torch, sample, gs, hdr, work)
                progress("e2e: running the executables on them")
                n_junc = int((h_last["edges"]["counts"].sum(axis=1) >= 5).sum())
                # what the resident step's result reads as text: linear ++ cycles without duplicates (palace:594-600)
                lin, cyc = stage04_io.matching_text(*h_last["result"], gs["names"], self_loops=True, break_cycles=False)
                cl = cyc.splitlines(keepends=True)
                pairs = list(dict.fromkeys(zip(cl[0::2], cl[1::2] + (["\n"] if len(cl) % 2 else []))))      # remove_cycle_dup.py:3-30
                # ... and what its depth sums, copy numbers and edges read as `_graph.txt` (generate_graph.cpp:1019-1076)
                want_graph = graph_text(gs["names"], gs["lens"], h_last["consumed"], h_last["cn"], h_last["edges"])
                out["e2e"] = run_e2e(paths, gs["avg_depth"], args.contigs, r, n_junc, lin + "".join(a + b for a, b in pairs), want_graph)
            except Exception as e:                   # never let this leg break the headline line
                out["e2e"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                paths = None
        else:
            work = paths = None
        try:
            if world == 1 and not solo and not args.no_cpu_baseline:         # (rank 0 at N = 1 only: the contract; the other ranks would wait for it)
                progress("cpu_baseline")
                res_v, contig_of = h_last["result"]
                seg_flags, edge_flags = stage04.flags(len(h_last["edges"]))
                lin, cyc = stage04_io.matching_text(res_v, contig_of, gs["names"], self_loops=True, break_cycles=False)
                cl = cyc.splitlines(keepends=True)
                pairs = list(dict.fromkeys(zip(cl[0::2], cl[1::2] + (["\n"] if len(cl) % 2 else []))))
                out["cpu_baseline"] = cpu_baseline(torch, sample, gs, hdr, args.cpu_sample_frac,
                                                   dict(contig_of=np.asarray(contig_of).copy(), cn=h_last["cn"], edges=h_last["edges"], edge_flags=edge_flags,
                                                        result_text=lin + "".join(a + b for a, b in pairs)), paths=paths)
                for k, v in out["cpu_baseline"].get("parity", {}).items():     # the checker beside the timing: a free parity point per stage
                    if v is False:
                        failures.append(f"cpu_baseline.parity.{k} is False")
                # files -> files against the CPU path timed the same way (files in, process exit): the like-for-like ratio, stated
                e2e, cb = out.get("e2e") or {}, out["cpu_baseline"]
                if cb.get("value") and "error" not in e2e and e2e.get("contigs_per_s"):
                    one, nf = e2e.get("one_process_stage04") or {}, e2e.get("no_fork") or {}
                    ratio = lambda x: None if not x else round(x / cb["value"], 1)
                    e2e["vs_cpu_baseline"] = dict(chain=ratio(e2e["contigs_per_s"]), fused=ratio(one.get("contigs_per_s")),
                                                  no_fork_chain=ratio(nf.get("contigs_per_s")), no_fork_fused=ratio(nf.get("one_process_stage04_contigs_per_s")),
                                                  cores=cb.get("cores"), multi_thread_chain=ratio(e2e["contigs_per_s"]) and cb.get("multi_thread", {}).get("value") and
                                                  round(e2e["contigs_per_s"] / cb["multi_thread"]["value"], 1),
                                                  note="executables on files (wall clock to the caller's return) / cpu_baseline.value (one thread, files in, to process exit); "
                                                       "`value` / cpu_baseline.value is NOT like for like: the headline step starts from inputs resident in HBM")
        finally:
            if work and not os.environ.get("PALACE_BENCH_KEEP"):              # (tools/eref_cli_repeat.sh re-runs the executables on these files)
                import shutil
                shutil.rmtree(work, ignore_errors=True)
        # a line whose own cross-checks failed is still printed, but the run does not pass: wrong refs, the executables on the
        # files disagreeing with the resident step (or the leg raising), results that differ from step to step
        e2e = out.get("e2e")
        if reported != len(sample["present"]) and args.contigs >= 1_000_000 and args.refs == 5000 and os.environ.get("PALACE_BENCH_SKIP_EREF") != "1" \
                and not os.environ.get("PALACE_OPT_KEY_SHARE") and not os.environ.get("PALACE_OPT_COUNTS_SHARE"):     # (below 1M contigs the read depth leaves a few present refs short; a tuning run that counts one rank's key share is partial by design)
            failures.append(f"refs_reported {reported} != refs_present {len(sample['present'])}")
        if out["config"]["result_digest"]["identical_over_untimed_steps"] is False:
            failures.append("result digests differ between untimed steps")
        if e2e is not None:
            if "error" in e2e:
                failures.append("e2e leg raised: " + e2e["error"])
            else:
                for k in ("agrees_with_resident_step", "all_result_identical_to_resident_step", "graph_txt_identical_to_resident_step"):
                    if e2e.get(k) is not True:
                        failures.append(f"e2e.{k} is {e2e.get(k)}")
                if (e2e.get("no_fork") or {}).get("same_outputs") is False:
                    failures.append("e2e.no_fork: the one-process runs wrote other outputs than the forked ones")
                one = e2e.get("one_process_stage04") or {}
                if one.get("files_identical_to_the_chain") is not True:
                    failures.append("e2e.one_process_stage04: " + str(one.get("error", "files differ from the chain's")))
    capi._check(L.palace_eref_probe_index_free(ctx.h, probe_index), "probe index free")
    if stage04 is not None:
        stage04.close()
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

