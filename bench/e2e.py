"""files -> files: the bench sample as the FILES the pipeline hands to the three executables (palace:473-480, 555-600), and the
chain of executables on them, one process per stage as the driver runs them."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .sample import READ_LEN, paths_text, progress


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


def e2e_paths(work):
    """the file names of one sample's work directory"""
    return {k: os.path.join(work, v) for k, v in dict(
        fq1="reads_1.fq", fq2="reads_2.fq", fa="phagedb.fa", hdr="coder.hdr", bam="reads_pe_primary.sort.bam", cols="bam_cols",
        fastg_fai="assembly_graph.fastg.fai", fasta_fai="assembly_graph.fasta.fai", blast="assembly_graph.fasta.blast",
        hit="hit_seqs.out", score="node_scores.out", paths="contigs.paths", graph="s_graph.txt", pre="s_filtered_graph_pre.txt",
        filt="s_filtered_graph.txt", allhit="all_hit_segs.txt", lin="s_linear.txt", cyc="s_cycle.txt", nodup="s_cycle_nodup.txt",
        result="s_all_result.txt", refnames="s_ref_names.txt", tmp="s_tmp.txt").items()}


def write_e2e_inputs(torch, sample, gs, hdr, work):
    """Every file of palace:473-480 and 555-600 for this sample.  Generation is not timed."""
    t0 = time.perf_counter()
    P = e2e_paths(work)
    write_eref_inputs(torch, sample, hdr, P)
    write_graph_inputs(gs, P)
    P["gen_s"] = time.perf_counter() - t0
    P["bytes"] = {k: os.path.getsize(P[k]) for k in ("fq1", "fq2", "fa", "bam", "fastg_fai")}
    return P


def write_eref_inputs(torch, sample, hdr, P):
    """eref's side: the two FASTQ files, the phage DB, the coder header its first run builds the index with"""
    n_side = sample["n_reads_side"]
    fastq_to_file(torch, sample["r1"], n_side, "1", P["fq1"])
    fastq_to_file(torch, sample["r2"], n_side, "2", P["fq2"])
    rb, ro = sample["ref_bases"].cpu().numpy(), sample["ref_off"].cpu().numpy()
    with open(P["fa"], "wb") as f:
        for i in range(sample["n_refs"]):
            b = rb[ro[i]:ro[i + 1]].tobytes()
            f.write(b">phage_%d synthetic\n" % (i + 1) + b"\n".join(b[k:k + 80] for k in range(0, len(b), 80)) + b"\n")
    open(P["hdr"], "wb").write(np.asarray(hdr, dtype=np.uint8).tobytes())


def write_graph_inputs(gs, P, bam=True):
    """generateGraph's and stage 04's side: the BAM (and the decoded columns it is made from, which the tests' oracle chain reads
    back), the FASTG .fai, the side files of filter_graph.py, contigs.paths"""
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
    if bam:
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


def graph_text(names, lens, consumed, cn, edges, min_count=5, rank=None):
    """`_graph.txt` as generate_graph.cpp:1019-1076 writes it, from the numbers the device holds: SEG lines in name-byte order with
    depth = consumed / max(1, L) at stream precision 6 (%g), then JUNC lines in (left, right, oL, oR) order of the names for edges
    whose four counters add up to min_count; edges: records of capi.EDGE_DTYPE"""
    if rank is None:                                                    # (rank: the names' byte order when the caller has it already)
        order = np.argsort(np.array(names, dtype="S"), kind="stable")
        rank = np.empty(len(names), dtype=np.int64)
        rank[order] = np.arange(len(names))
    else:
        rank = np.asarray(rank, dtype=np.int64)
        order = np.empty(len(names), dtype=np.int64)
        order[rank] = np.arange(len(names))
    depth = np.asarray(consumed, dtype=np.float64) / np.maximum(1, np.asarray(lens, dtype=np.int64))
    out = ["SEG %s %s %d\n" % (names[i], "%g" % depth[i], cn[i]) for i in order.tolist()]
    tot = edges["counts"].astype(np.int64).sum(axis=1)
    e = edges[tot >= min_count]
    k = np.lexsort((e["oR"], e["oL"], rank[e["right"]], rank[e["left"]]))
    for left, right, counts, oL, oR in zip(e["left"][k].tolist(), e["right"][k].tolist(), e["counts"][k].tolist(), e["oL"][k].tolist(), e["oR"][k].tolist()):
        supp, supp_nf, span, span_nf = counts
        out.append("JUNC %s %s %s %s %d %d\n" % (names[left], "+-"[oL], names[right], "+-"[oR], supp + span + supp_nf, span_nf))
    return "".join(out)


def run_e2e(P, avg_depth, n_contigs, rows_host, n_junc_expected, result_text_expected=None, graph_text_expected=None):
    """The chain of palace:473-480 and 555-600 on the files, one process per stage as the driver runs them.  Returns wall
    seconds per stage.  eref is run twice: the first run builds <db>.k32.index.dat (once per DB, extract_ref.cpp:1245-1251),
    the second finds it -- the steady state of a DB shared by many samples and the one that enters `seconds`."""
    import subprocess
    B = os.path.join(ROOT, "palace_amd", "bin")
    S = os.path.join(ROOT, "palace_amd", "scripts")
    fused_cmd = None
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
        fused_cmd = [os.path.join(B, "generateGraph"), "--hit-seqs", P["hit"], "--node-scores", P["score"], "--blast", P["blast"], "--fasta-fai", P["fasta_fai"],
                     "--paths", P["paths"], "--filtered-pre", fp["pre"], "--filtered", fp["filt"], "--all-hit-segs", fp["allhit"], "--linear", fp["lin"],
                     "--cycle", fp["cyc"], "--cycle-nodup", fp["nodup"], "--all-result", fp["result"], "-s", "-i", "10",
                     P["bam"], P["fastg_fai"], fp["graph"], f"{avg_depth:.6g}"]
        subprocess.run(fused_cmd, check=True)
        t_fused = time.perf_counter() - t0
        same = all(open(fp[k], "rb").read() == open(P[k], "rb").read() for k in ("graph", "pre", "filt", "allhit", "lin", "cyc", "nodup", "result"))
        fused = dict(seconds=st["eref"] + t_fused, contigs_per_s=n_contigs / (st["eref"] + t_fused),
                     stage_s=dict(eref=round(st["eref"], 3), generateGraph_with_stage04=round(t_fused, 3)),
                     files_identical_to_the_chain=bool(same),
                     note="eref + ONE generateGraph process that also writes _filtered_graph_pre / _filtered_graph / all_hit_segs / linear / cycle / "
                          "cycle_nodup / all_result (its --filtered-pre ... --all-result options)")
    except Exception as e:
        fused = dict(error=f"{type(e).__name__}: {str(e)[:200]}")
    # The executables fork first thing and the process the caller started leaves when every output is closed, while the worker's
    # address space is torn down behind its back (host/fast_exit.hpp).  The same three programs as ONE process each
    # (PALACE_NO_FORK=1: the wait includes the teardown, as it does for any CPU comparator), outputs to scratch names:
    nf = {}
    progress("e2e: chain and one-process run done; the same as single processes (PALACE_NO_FORK=1)")
    try:
        env1 = dict(os.environ, PALACE_NO_FORK="1")
        time.sleep(0.5)                              # (the last worker above may still be going away)
        with open(P["refnames"] + ".nofork", "wb") as f:
            t0 = time.perf_counter(); subprocess.run(eref, check=True, stdout=f, env=env1); nf["eref"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        subprocess.run([os.path.join(B, "generateGraph"), P["bam"], P["fastg_fai"], P["graph"] + ".nofork", f"{avg_depth:.6g}"], check=True, env=env1)
        nf["generateGraph"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        subprocess.run([os.path.join(B, "matching"), "-g", P["filt"], "-r", P["lin"] + ".nofork", "-c", P["cyc"] + ".nofork", "-s", "-i", "10", "-l", P["paths"]],
                       check=True, env=env1)
        nf["matching"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        subprocess.run([a + ".nofork" if a.endswith(".fused") else a for a in fused_cmd], check=True, env=env1)
        nf["generateGraph_with_stage04"] = time.perf_counter() - t0
        nf["same_outputs"] = bool(all(open(P[k] + ".nofork", "rb").read() == open(P[k], "rb").read() for k in ("refnames", "graph", "lin", "cyc"))
                                  and open(P["result"] + ".fused.nofork", "rb").read() == open(P["result"], "rb").read())
    except Exception as e:
        nf = dict(error=f"{type(e).__name__}: {str(e)[:200]}")
    # cross-check against the HBM-resident step (same coder header, same sample): reported refs and kept junctions
    r = rows_host
    want = {(i + 1, int(r[i, 0]), int(r[i, 1])) for i in range(len(r))
            if r[i, 1] > 0 and np.float32(r[i, 1]) / np.float32(r[i, 2]) > np.float32(0.75)}
    got = {tuple(int(x) for x in l.split("\t")[1:4]) for l in open(P["refnames"]).read().splitlines()}
    n_junc = sum(1 for l in open(P["graph"]) if l.startswith("JUNC"))
    total = sum(v for k, v in st.items() if k != "eref_first_run_builds_index")
    same_result = None if result_text_expected is None else bool(open(P["result"]).read() == result_text_expected)
    same_graph = None if graph_text_expected is None else bool(open(P["graph"]).read() == graph_text_expected)
    no_fork = nf
    if "error" not in nf:
        chain_nf = total - st["eref"] - st["generateGraph"] - st["matching"] + nf["eref"] + nf["generateGraph"] + nf["matching"]
        no_fork = dict(seconds=chain_nf, contigs_per_s=n_contigs / chain_nf, stage_s={k: round(v, 3) for k, v in nf.items() if k != "same_outputs"},
                       one_process_stage04_seconds=nf["eref"] + nf["generateGraph_with_stage04"],
                       one_process_stage04_contigs_per_s=n_contigs / (nf["eref"] + nf["generateGraph_with_stage04"]), same_outputs=nf["same_outputs"],
                       note="PALACE_NO_FORK=1: eref, generateGraph and matching as one process each, so that the wait for a stage includes the "
                            "teardown of its address space (mapped inputs, the inflated BAM stream, the HIP runtime) -- the like-for-like figure "
                            "beside a CPU comparator timed to process exit; the other stages of the chain as timed above")
    return dict(seconds=total, contigs_per_s=n_contigs / total, no_fork=no_fork, stage_s={k: round(v, 3) for k, v in st.items()},
                agrees_with_resident_step=bool(got == want and n_junc == n_junc_expected and same_result is not False and same_graph is not False),
                all_result_identical_to_resident_step=same_result, graph_txt_identical_to_resident_step=same_graph, one_process_stage04=fused,
                refs_reported=len(got), junc_lines=n_junc, result_lines=sum(1 for _ in open(P["result"])),
                input_bytes=P["bytes"], input_generation_s=round(P["gen_s"], 1),
                note="wall clock of eref + generateGraph + filter_graph.py + uniq + matching + remove_cycle_dup.py + cat, one process "
                     "per stage, files in the page cache; eref with the index file of the DB present (built by the first run)")

