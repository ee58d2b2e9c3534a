"""`cpu_baseline`: the CPU path timed on this host, on a bounded sample of the workload -- the ONLY part of the bench that may
touch oracle/ (the CPU restatement of the reference, and oracle/_ref: the unmodified reference compiled by oracle/Makefile)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .sample import READ_LEN, paths_text, progress


def reference_eref(b1, b2, off, refs, tmp, small_div=10):
    """The COMPILED reference (oracle/_ref/eref_ref = the unmodified extract_ref.cpp, built by oracle/Makefile; it travels with the
    repo) on the read sample the port is timed on: threads=1, index cached by an untimed first run, at two sizes (the whole sample
    and 1/small_div of it) -> marginal reads/s and the fixed seconds of a run (4 GiB table + the never-read 16.3 GiB Peaks arrays
    allocated and zeroed, extract_ref.cpp:1257, 1296-1299, + the scan of this small DB).  Each timed run's stdout is compared
    with `palace_amd/bin/eref` on the same files and the same index (one more parity point, on this box): the big sample with the
    thresholds 0.01 / 0.0 -- at a twentieth of the read depth nothing passes the pipeline's 0.9 / 0.85, and equal empty outputs
    say little; the thresholds do not enter the reference's run time --, the small one with the pipeline's 0.9 / 0.85."""
    import subprocess
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "eref_ref")
    eref_bin = os.path.join(ROOT, "palace_amd", "bin", "eref")
    if not os.path.exists(ref_bin):
        return None
    try:
        fa = os.path.join(tmp, "db.fa")
        with open(fa, "wb") as f:
            for i, r in enumerate(refs):
                f.write(b">ref%d\n" % i + r.tobytes() + b"\n")
        n = len(off) - 1
        fq = lambda tag, m: os.path.join(tmp, f"s{m}_{tag}.fq")
        sizes = sorted({max(500, n // small_div), n})
        for m in sizes:
            for tag, b in (("1", b1), ("2", b2)):
                rec = np.empty((m, 13 + READ_LEN + 3 + READ_LEN + 1), dtype=np.uint8)          # "@r%010d\n" seq "\n+\n" qual "\n"
                rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
                idx = np.arange(m)
                for k in range(10):
                    rec[:, 2 + k] = (idx // 10 ** (9 - k)) % 10 + 48
                rec[:, 12] = 10
                rec[:, 13:13 + READ_LEN] = b[: m * READ_LEN].reshape(m, READ_LEN)
                rec[:, 13 + READ_LEN:16 + READ_LEN] = np.frombuffer(b"\n+\n", dtype=np.uint8)
                rec[:, 16 + READ_LEN:16 + 2 * READ_LEN] = ord("I")
                rec[:, 16 + 2 * READ_LEN] = 10
                rec.tofile(fq(tag, m))
        cmd = lambda exe, m, hit, perfect, threads: [exe, fq("1", m), fq("2", m), fa, os.path.join(tmp, "t.txt"), hit, perfect, threads]
        # untimed: leaves <db>.k32.index.dat beside the DB (the reference builds it on first use, extract_ref.cpp:1245-1251)
        subprocess.run(cmd(ref_bin, sizes[0], "0.9", "0.85", "1"), stdout=subprocess.DEVNULL, check=True, timeout=600)
        times, parity = {}, {}
        for m in sizes:
            progress(f"cpu_baseline: the compiled reference on {2 * m} reads")
            hit, perfect = ("0.01", "0.0") if m == sizes[-1] and len(sizes) > 1 else ("0.9", "0.85")
            t0 = time.perf_counter()
            r = subprocess.run(cmd(ref_bin, m, hit, perfect, "1"), stdout=subprocess.PIPE, check=True, timeout=900)
            times[2 * m] = time.perf_counter() - t0
            ours = subprocess.run(cmd(eref_bin, m, hit, perfect, str(min(16, os.cpu_count() or 1))), stdout=subprocess.PIPE, check=True, timeout=600)
            parity[f"{2 * m} reads, thresholds {hit} {perfect}"] = dict(identical=bool(ours.stdout == r.stdout), lines=r.stdout.count(b"\n"))
        # one more parity point, untimed, at the PIPELINE's thresholds with refs that pass them: the sampled reads above are a twentieth (and
        # a two-hundredth) of the depth, so nothing of them passes 0.9 / 0.85 and equal empty outputs say little; here two refs of the DB are
        # tiled at 12x (both strands, a few substitutions) on top of a slice of the sample
        try:
            rng = np.random.Generator(np.random.PCG64(7))
            comp = np.zeros(256, dtype=np.uint8)
            comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
            tiles = []
            for r in refs[:2]:
                st = np.arange(0, max(1, len(r) - READ_LEN), max(1, READ_LEN // 12))
                tiles.append(r[st[:, None] + np.arange(READ_LEN)[None, :]])
            t1 = np.concatenate(tiles)
            sub = rng.random(t1.shape) < 0.003
            t1 = np.where(sub, np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=t1.shape)], t1)
            k = min(2000, n)
            both = (np.concatenate([t1, b1[: k * READ_LEN].reshape(k, READ_LEN)]), np.concatenate([comp[t1[:, ::-1]], b2[: k * READ_LEN].reshape(k, READ_LEN)]))
            for tag, arr in zip(("1", "2"), both):
                with open(fq(tag, "tiled"), "wb") as f:
                    q = b"I" * READ_LEN
                    f.write(b"".join(b"@t%d\n" % i + arr[i].tobytes() + b"\n+\n" + q + b"\n" for i in range(len(arr))))
            r = subprocess.run(cmd(ref_bin, "tiled", "0.9", "0.85", "1"), stdout=subprocess.PIPE, check=True, timeout=600)
            ours = subprocess.run(cmd(eref_bin, "tiled", "0.9", "0.85", str(min(16, os.cpu_count() or 1))), stdout=subprocess.PIPE, check=True, timeout=600)
            parity[f"two refs tiled at 12x + {2 * k} sampled reads, thresholds 0.9 0.85"] = dict(identical=bool(ours.stdout == r.stdout and r.stdout.count(b"\n") >= 1),
                                                                                               lines=r.stdout.count(b"\n"))
        except Exception as e:
            parity["two refs tiled at 12x"] = dict(identical=False, lines=0, error=f"{type(e).__name__}: {str(e)[:200]}")
        (ra, ta), (rbn, tb) = min(times.items()), max(times.items())
        marginal = (rbn - ra) / max(1e-9, tb - ta) if rbn > ra else None
        return dict(binary="oracle/_ref/eref_ref (unmodified extract_ref.cpp, g++ -O2, threads=1, index cached by an untimed first run)",
                    runs_s={str(k): round(v, 2) for k, v in times.items()}, marginal_reads_per_s=marginal,
                    fixed_s=None if marginal is None else ta - ra / marginal, refs_in_db=len(refs),
                    stdout_vs_eref_cli=parity, stdout_identical=all(v["identical"] for v in parity.values()))
    except Exception as e:                           # never let the cross-check break the bench line
        return dict(error=f"{type(e).__name__}: {str(e)[:300]}")


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


def cpu_baseline(torch, sample, gs, header, frac, graph_out, paths=None):
    """The reference's CPU path on a bounded sample of the workload, timed on this host at threads = 1 (the only configuration in
    which the reference's semantics are defined):
      eref           the COMPILED reference (oracle/_ref/eref_ref, the unmodified extract_ref.cpp) on `frac` of the reads against a
                     1 % DB: its measured marginal reads/s scales the read term; its stdout is compared with bin/eref's.  The port
                     (oracle/eref_oracle.c, which keeps the reference's 32-step inner loop per position and channel) is timed on
                     the same reads beside it (`port`): it is the slower of the two and only stands in when the binary is absent.
      generateGraph  oracle/graph_oracle.cpp over ALL records (up to 8 M; above that `frac` of them, scaled) + BGZF inflate and BAM
                     decode of the whole file with zlib on one thread (what sam_read1 does); its text is compared with `_graph.txt`.
      matching       oracle/match_oracle.cpp on the whole filtered graph with contigs.paths; compared with the step's result.
    `value` = contigs / (eref + generateGraph + matching), the table memset counted once; `as_shipped` adds the reference's dead
    16.3 GiB allocation + memset (extract_ref.cpp:1296-1299); `multi_thread` = read counting / ref scan / inflate on min(nproc, 16)
    threads (the port's race-free compare-and-swap increments)."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as orc
    cores = max(1, min(os.cpu_count() or 1, 16))
    cc = orc.header_to_cc(header)
    tmp = tempfile.mkdtemp(prefix="palace_bench_", dir=os.environ.get("PALACE_BENCH_TMP", "/tmp"))
    # ---- eref ----
    n_side = max(1000, min(sample["n_reads_side"], int(frac * sample["n_reads_side"])))
    b1 = sample["r1"][: n_side * READ_LEN].cpu().numpy()
    b2 = sample["r2"][: n_side * READ_LEN].cpu().numpy()
    off = np.arange(n_side + 1, dtype=np.int64) * READ_LEN
    # the DB of the sample: 1 % of the refs, half of them refs the reads come from (so that the scan has windows to find)
    n_ref_s = max(1, sample["n_refs"] // 100)
    present = [int(x) for x in sample["present"][: n_ref_s // 2]]
    absent = [i for i in range(sample["n_refs"]) if i not in set(int(x) for x in sample["present"])][: n_ref_s - len(present)]
    ro_all = sample["ref_off"].cpu().numpy()
    refs = [sample["ref_bases"][int(ro_all[i]):int(ro_all[i + 1])].cpu().numpy() for i in sorted(present + absent)]
    idx = [orc.index_ref(r, cc) for r in refs]      # cached index: not timed
    progress(f"cpu_baseline: the port on {2 * n_side} reads, one thread")
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
        orc.scan_ref(idx[i], len(refs[i]), table, 0.9, 0.85)
    t_refs = time.perf_counter() - t0
    table.clear()
    progress(f"cpu_baseline: the port on {cores} threads")
    t0 = time.perf_counter()
    table.count_mt(b1, off, cc, cores)
    table.count_mt(b2, off, cc, cores)
    t_reads_mt = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:     # refs are independent (the reference splits them over T threads,
        list(ex.map(lambda i: orc.scan_ref(idx[i], len(refs[i]), table, 0.9, 0.85), range(n_ref_s)))   # :1314-1329)
    t_refs_mt = time.perf_counter() - t0
    table.free()
    progress("cpu_baseline: the reference's dead 16.3 GiB allocation")
    t0 = time.perf_counter()
    dead = orc.lib().orc_eref_reference_dead_cost()        # 16 GiB + 300 MB allocated and zeroed, never read
    t_dead = time.perf_counter() - t0 if dead else None
    total_reads = 2 * sample["n_pairs_total"]
    up_reads, up_refs = total_reads / (2 * n_side), sample["n_refs"] / n_ref_s
    t_eref_port = t_clear + t_reads * up_reads + t_refs * up_refs
    t_eref_mt = t_clear + t_reads_mt * up_reads + t_refs_mt * up_refs
    # the compiled reference on the same reads (and the same 1 % DB): its marginal rate is the read term of `value`
    progress("cpu_baseline: the compiled reference (three runs)")
    ref = reference_eref(b1, b2, off, refs, tmp)
    progress("cpu_baseline: generateGraph through the port")
    ref_ok = ref is not None and "error" not in ref and ref.get("marginal_reads_per_s")
    if ref_ok:
        t_eref = t_clear + total_reads / ref["marginal_reads_per_s"] + t_refs * up_refs
        ref["eref_seconds_extrapolated"] = t_eref
        ref["eref_seconds_as_shipped"] = max(ref["fixed_s"], t_clear) + total_reads / ref["marginal_reads_per_s"] + t_refs * (up_refs - 1)
    else:
        t_eref = t_eref_port
    if ref is not None and "error" not in ref and sample["n_contigs"] == 1_000_000:
        ref["full_size_note"] = ("measured once on a GPU box, not in this run: the compiled reference on the full eref input of "
                                 "this workload (5000 refs, 6.67 M reads) took 595.6 s at threads=1 incl. its index build, stdout "
                                 "byte-identical to ours (profiles/ref_compare_eref_full.log)")
    del b1, b2
    # ---- generateGraph: the records of the sorted stream as BAM-level records (single thread, as the reference) ----
    m = gs["n"] if gs["n"] <= 8_000_000 else max(1000, int(frac * gs["n"]))
    c = {k: v[:m].cpu().numpy() for k, v in gs["col"].items()}
    so = gs["sa_off"][: m + 1].cpu().numpy().astype(np.int64)
    sa = gs["sa"][: max(1, int(so[-1]))].cpu().numpy()
    names = gs["names"]
    gin = orc.GraphInput.from_columns(c, so, sa, names, gs["lens"])             # marshalling is not timed
    if paths and m == gs["n"]:
        fai = paths["fastg_fai"]                   # the sample's own file: the .fai parse (generate_graph.cpp:119-169) is in the time
    else:
        fai = os.path.join(tmp, "g.fastg.fai")
        hot = sorted(set(c["tid"].tolist()) | set(gs["link"][c["tid"]].tolist()))
        with open(fai, "w") as f:                  # reduced .fai: only contigs the sample can touch
            for a in hot:
                f.write(f"{names[a]}:{names[gs['link'][a]]};\t{gs['lens'][a]}\t0\t60\t61\n")
    t0 = time.perf_counter()
    graph_txt = gin.run(fai, gs["avg_depth"])
    t_graph_s = time.perf_counter() - t0
    t_graph = t_graph_s * gs["n_total"] / m
    parity = {}
    if paths and m == gs["n"] and os.path.exists(paths["graph"]):
        parity["graph_txt_identical_to_generateGraph"] = bool(open(paths["graph"], "rb").read() == graph_txt)
    del gin, graph_txt, c
    progress("cpu_baseline: BAM decode, matching")
    # ... and what the reference's loop spends inside sam_read1: BGZF inflate + record decode of the whole BAM (not a sample)
    dec = bam_decode_seconds(paths["bam"] if paths else None, cores)
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
    cap = 160 * len(graph_out["contig_of"]) + (1 << 20)
    t0 = time.perf_counter()
    lin, cyc = orc.match_run(gpath, ppath, 10, self_loops=True, cap=cap)
    t_match = time.perf_counter() - t0
    if graph_out.get("result_text") is not None:
        cl = cyc.decode().splitlines(keepends=True)
        pairs = list(dict.fromkeys(zip(cl[0::2], cl[1::2] + (["\n"] if len(cl) % 2 else []))))          # remove_cycle_dup.py:3-30
        parity["all_result_identical_to_resident_step"] = bool(lin.decode() + "".join(a + b for a, b in pairs) == graph_out["result_text"])
    if ref is not None and "error" not in ref:
        parity["eref_stdout_identical_to_compiled_reference"] = ref.get("stdout_identical")
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    t_rest = t_graph + t_match + (t_decode or 0.0)
    t_full, t_full_port, t_full_mt = t_eref + t_rest, t_eref_port + t_rest, t_eref_mt + t_graph + t_match + (t_decode_mt or 0.0)
    nc = sample["n_contigs"]
    stage_kind = dict(eref=("reference: oracle/_ref/eref_ref, the unmodified extract_ref.cpp compiled with g++ -O2" if ref_ok else "port: oracle/eref_oracle.c"),
                      generateGraph="port: oracle/graph_oracle.cpp (generate_graph.cpp needs htslib, absent: unbuildable here) + this repo's BAM loader with zlib on one thread",
                      matching="port: oracle/match_oracle.cpp (this repo's own algorithm; bin/matching is absent from the reference tree)")
    out = dict(value=nc / t_full, unit="contigs/s", cores=1, kind="reference" if ref_ok else "port", kind_by_stage=stage_kind,
               sample=("threads=1. eref: " + (f"the compiled reference on {2 * n_side} of {total_reads} reads x{READ_LEN} bp against {n_ref_s} of {sample['n_refs']} refs "
                                              f"(runs {ref['runs_s']} s -> {ref['marginal_reads_per_s']:.0f} reads/s marginal, {ref['fixed_s']:.1f} s fixed per run as shipped); "
                                              if ref_ok else f"the port on {2 * n_side} of {total_reads} reads x{READ_LEN} bp ({t_reads:.1f} s); ")
                       + f"+ 4 GiB table memset ({t_clear:.1f} s, fixed) + the port's scan of {n_ref_s} of {sample['n_refs']} refs ({t_refs:.2f} s, scaled) -> {t_eref:.0f} s; "
                       f"generateGraph: {m} of {gs['n_total']} decoded records through the port ({t_graph_s:.1f} s" + (", the .fai parse included" if fai == (paths or {}).get("fastg_fai") else "; full .fai parse excluded") + f") -> {t_graph:.0f} s, "
                       + (f"plus BGZF inflate + BAM decode of the whole {dec['bam_bytes'] / 1e6:.0f} MB BAM with zlib on one thread, as htslib's sam_read1 "
                          f"does ({t_decode:.1f} s, measured, not extrapolated)" if t_decode is not None else "BGZF/BAM decode NOT included (no BAM file in this run: --no-e2e)")
                       + f"; matching: the whole filtered graph with contigs.paths ({t_match:.1f} s; own "
                       f"algorithm, reference absent); filter_graph.py itself (Python glue) is not in the sum."),
               stage_s=dict(eref=t_eref, generateGraph=t_graph, generateGraph_bam_decode=t_decode, matching=t_match),
               extrapolated=("eref's read term = all reads / the measured marginal rate, its ref scan = the sample's refs scaled linearly; "
                             + ("generateGraph's record loop is whole" if m == gs["n"] else "generateGraph's record loop is timed on the first records named in `sample` and scaled linearly")
                             + "; table memset, BAM decode and matching are whole"),
               parity=parity, bam_decode_s=dec,
               port=dict(value=nc / t_full_port, unit="contigs/s", cores=1, kind="port", reads_per_s=2 * n_side / t_reads, stage_s=dict(eref=t_eref_port),
                         note=f"the same sum with eref's read term from oracle/eref_oracle.c ({2 * n_side} reads in {t_reads:.1f} s)"),
               multi_thread=dict(value=nc / t_full_mt, unit="contigs/s", cores=cores, kind="port",
                                 note=f"read counting and ref scan on {cores} threads (same sample: {t_reads_mt:.1f} s and {t_refs_mt:.2f} s); "
                                      "generateGraph's record loop and matching single-threaded, as the reference's are; BGZF inflate on the same threads (htslib can: bgzf_mt)",
                                 stage_s=dict(eref=t_eref_mt, generateGraph=t_graph, generateGraph_bam_decode=t_decode_mt, matching=t_match)),
               reference_eref=ref)
    if ref_ok:
        out["as_shipped"] = dict(value=nc / (ref["eref_seconds_as_shipped"] + t_rest), unit="contigs/s", cores=1, kind="reference",
                                 note=f"threads=1 with the reference's measured fixed cost per run ({ref['fixed_s']:.1f} s on this host: the 4 GiB table and its "
                                      "never-read Peaks arrays, 16 GiB + 300 MB allocated and zeroed, extract_ref.cpp:1257, 1296-1299) in place of the table memset alone")
    elif t_dead is not None:
        out["as_shipped"] = dict(value=nc / (t_full + t_dead), unit="contigs/s", cores=1, kind="port",
                                 note=f"threads=1 plus the reference's never-read Peaks arrays: 16 GiB + 300 MB allocated and zeroed "
                                      f"({t_dead:.1f} s on this host, fixed per run; extract_ref.cpp:1296-1299)")
    return out
