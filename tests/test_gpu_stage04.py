"""The resident stage 04 (palace_stage04_*: filter_graph.py's selection and `matching`, both on the device) against
  * the outputs of the REFERENCE's filter_graph.py (tests/golden/filter_cases.npz): kept SEG set as a sorted multiset, JUNC list in order;
  * this repository's own file chain (scripts/filter_graph.py -> uniq -> bin/matching -l contigs.paths): byte-identical linear / cycle files.
matching itself stays "parity unpinned" (the reference binary is absent): the second check pins the in-memory
path to the file path, not to the reference."""
import os
import subprocess
import sys

import numpy as np
import pytest

from palace_amd import capi, stage04_io, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = os.path.join(ROOT, "palace_amd", "scripts")
BIN = os.path.join(ROOT, "palace_amd", "bin")
G = np.load(os.path.join(ROOT, "tests", "golden", "filter_cases.npz"))


def _script_filter():
    sys.path.insert(0, SCRIPTS)
    import filter_graph
    return filter_graph


def _load_case(files, tmp_path, blast_ratio=0.7, score_threshold=0.7):
    """the text inputs of filter_graph.py -> the arrays of palace_stage04_inputs + the edges of the graph text"""
    P = lambda n: str(tmp_path / n)
    for k in ("fasta_fai", "blast", "hit_seqs", "node_scores"):
        open(P(k), "w").write(files[k])
    flt = _script_filter().GraphFilter(blast_ratio, score_threshold)      # (the script's own readers of the side files)
    flt.load_fasta_index(P("fasta_fai")); flt.load_blast(P("blast")); flt.load_gene_hits(P("hit_seqs")); flt.load_scores(P("node_scores"))
    names = [l.split("\t")[0] for l in files["fasta_fai"].splitlines()]
    index = {n: i for i, n in enumerate(names)}
    seed = np.array([(n in flt.blast_hit) | ((n in flt.gene_hit) << 1) | ((n in flt.score_hit) << 2) for n in names], np.uint8)
    raw_seg, cn, tlen = {}, np.zeros(len(names), np.int32), np.zeros(len(names), np.int32)
    junc_lines, edges, seen = [], [], set()
    for line in files["graph"].splitlines(keepends=True):
        c = line.split()
        if c[0] == "SEG":
            raw_seg[c[1]] = line
            cn[index[c[1]]] = int(float(c[3]))
            tlen[index[c[1]]] = 1
        else:
            key = (index[c[1]], c[2] == "-", index[c[3]], c[4] == "-")
            if key in seen:
                continue                                     # the aggregated edge exists once; the file's duplicate line is dropped on output anyway
            seen.add(key)
            junc_lines.append(line)
            edges.append((key[0], key[2], (int(c[5]), 0, 0, int(c[6])), key[1], key[3], (0,) * 6))
    e = np.array(edges, dtype=capi.EDGE_DTYPE) if edges else np.zeros(0, capi.EDGE_DTYPE)
    off, tok = stage04_io.paths_csr(files["contigs_paths"], names)
    return dict(names=names, seed=seed, tlen=tlen, rank=stage04_io.name_ranks(names), name_len=stage04_io.name_lengths(names),
                path_off=off, path_tok=tok, edges=e, junc_lines=junc_lines, raw_seg=raw_seg, cn=cn, flt=flt)


def _run_filter(ctx, case, min_count):
    st = capi.Stage04(ctx, case["seed"], case["tlen"], case["rank"], case["name_len"], case["path_off"], case["path_tok"], min_count)
    e = case["edges"]
    d_e = ctx.upload(e.view(np.uint8).reshape(-1) if len(e) else np.zeros(32, np.uint8))
    d_n = ctx.upload(np.array([len(e)], np.int64))
    st.filter(d_e.ptr, d_n.ptr, max(1, len(e)))
    return st, d_e, d_n


@pytest.mark.parametrize("case_no", [0, 1, 2])
def test_in_memory_filter_equals_reference_outputs(tmp_path, case_no):
    files = {k: G[f"case{case_no}_{k}"].tobytes().decode() for k in ("graph", "fasta_fai", "blast", "hit_seqs", "node_scores", "contigs_paths", "pre")}
    case = _load_case(files, tmp_path)
    with capi.Ctx(0) as ctx:
        st, d_e, d_n = _run_filter(ctx, case, min_count=0)           # every JUNC line of the text is an edge here
        seg_flags, edge_flags = st.flags(len(case["edges"]))
        counts = st.counts()
        st.close()
    names, flt = case["names"], case["flt"]
    got_seg = sorted([flt.seg_line(names[i], case["raw_seg"][names[i]]) for i in np.flatnonzero(seg_flags & 1)] +
                     [case["raw_seg"][names[i]].strip() + " 0 1.0 0\n" for i in np.flatnonzero((seg_flags & 3) == 2)])
    got_junc = [l for l, f in zip(case["junc_lines"], edge_flags) if f & 2] + [l for l, f in zip(case["junc_lines"], edge_flags) if (f & 6) == 4]
    want = files["pre"].splitlines(keepends=True)
    assert got_seg == sorted(l for l in want if l.startswith("SEG"))
    assert got_junc == [l for l in want if not l.startswith("SEG")]
    assert counts["segs_selected"] + counts["segs_rescued"] == len(got_seg) and counts["kept_pass2"] + counts["kept_pass3_more"] == len(got_junc)
    assert len(got_junc) > 3 and counts["segs_rescued"] >= 0


def _file_chain(tmp_path, files, avg, flags):
    """scripts/filter_graph.py (native core) -> uniq -> bin/matching, as palace:566-591 runs them"""
    P = lambda n: str(tmp_path / n)
    for k, v in files.items():
        open(P(k), "w").write(v)
    subprocess.run([sys.executable, os.path.join(SCRIPTS, "filter_graph.py"), P("fastg_fai"), P("graph"), P("pre.txt"), f"{avg:.6g}", "0",
                    P("hit_seqs"), P("node_scores"), P("blast"), "0.7", P("fasta_fai"), P("hits.txt"), P("contigs_paths"), "0.7"], check=True)
    with open(P("filt.txt"), "wb") as f:
        subprocess.run(["uniq", P("pre.txt")], check=True, stdout=f)
    subprocess.run([os.path.join(BIN, "matching"), "-g", P("filt.txt"), "-r", P("lin.txt"), "-c", P("cyc.txt"), "-i", "10", "-l", P("contigs_paths")] + flags,
                   check=True)
    return open(P("lin.txt")).read(), open(P("cyc.txt")).read(), open(P("filt.txt")).read()


@pytest.mark.parametrize("seed,n_contigs,n_events,flags,iters", [(5, 60, 4000, ["-s"], 0), (6, 400, 30000, ["-s"], 0), (6, 400, 30000, ["-s"], 1),
                                                                   (7, 400, 30000, ["-b", "--aggressive"], 0), (7, 400, 30000, ["-b", "--aggressive"], 2),
                                                                   (8, 2500, 120000, ["-s"], 0)])
def test_in_memory_stage04_equals_the_file_chain(tmp_path, seed, n_contigs, n_events, flags, iters):
    from oracle import binding as orc
    rng = synth.rng_for(seed)
    targets, fai_text, recs, avg = synth.random_graph_case(rng, n_contigs, n_events)
    names, lens = [t[0] for t in targets], [t[1] for t in targets]
    open(tmp_path / "g.fastg.fai", "w").write(fai_text)
    o = orc.graph_default_opts()
    o.min_count = 2
    graph = orc.graph_run(recs, targets, str(tmp_path / "g.fastg.fai"), avg, o).decode()      # a `_graph.txt` (the checker's; any would do)
    files = dict(graph=graph, fastg_fai=fai_text, **synth.filter_side_files(rng, names, lens))
    lin, cyc, filt = _file_chain(tmp_path, files, avg, flags)
    case = _load_case(files, tmp_path)
    with capi.Ctx(0) as ctx:
        ctx.match_set_option("iters_per_round", iters)         # 0: the defaults; 1 or 2: rounds do not settle, the host-checked run takes over
        st, d_e, d_n = _run_filter(ctx, case, min_count=0)
        d_cn = ctx.upload(case["cn"])
        st.match(d_e.ptr, d_cn.ptr, 10, "--aggressive" in flags, True)
        res, contig_of = st.result()
        got_lin, got_cyc = stage04_io.matching_text(res, contig_of, names, self_loops="-s" in flags, break_cycles="-b" in flags)
        seg_order = [names[c] for c in contig_of]              # (views into the object's pinned memory: read before close)
        counts = st.counts()
        st.close()
    assert seg_order == [l.split(" ")[1] for l in filt.splitlines() if l.startswith("SEG")]     # the file's SEG order
    assert got_lin == lin
    assert got_cyc == cyc
    assert lin.count("\t") > 5 and counts["arcs"] > 10


def test_stage04_reports_what_the_reference_script_dies_on(tmp_path):
    """an id in contigs.paths that names no contig: filter_graph.py:138 raises KeyError; the resident filter reports it"""
    names = [f"EDGE_{i + 1}_length_{500 + i}_cov_3.0" for i in range(6)]
    off, tok = stage04_io.paths_csr("NODE_1\n1+,99-,2+\n", names)
    assert tok.tolist() == [0, -1, 2]
    with capi.Ctx(0) as ctx:
        st = capi.Stage04(ctx, np.ones(6, np.uint8), np.full(6, 500, np.int32), stage04_io.name_ranks(names), stage04_io.name_lengths(names), off, tok)
        d_e, d_n = ctx.upload(np.zeros(32, np.uint8)), ctx.upload(np.zeros(1, np.int64))
        st.filter(d_e.ptr, d_n.ptr, 1)
        with pytest.raises(capi.PalaceError, match="unknown contig id"):
            st.counts()
        st.close()


@pytest.mark.parametrize("seed,n_contigs,n_events,flags", [(15, 80, 6000, ["-s"]), (16, 500, 40000, ["-s", "-b"]), (17, 500, 40000, ["--aggressive"])])
def test_stage04_inside_generateGraph_writes_the_files_of_the_chain(tmp_path, seed, n_contigs, n_events, flags):
    """generateGraph with the stage-04 options (one process, the graph never leaves HBM) writes byte for byte the files the
    five-process chain of palace:555-600 writes: `_graph.txt`, `_filtered_graph_pre.txt`, `_filtered_graph.txt`,
    all_hit_segs.txt, linear, cycle, cycle_nodup, all_result."""
    rng = synth.rng_for(seed)
    targets, fai_text, recs, avg = synth.random_graph_case(rng, n_contigs, n_events)
    names, lens = [t[0] for t in targets], [t[1] for t in targets]
    P = lambda n: str(tmp_path / n)
    synth.write_bam(P("t.bam"), targets, recs, block=30000)
    side = synth.filter_side_files(rng, names, lens)
    for k, v in dict(fastg_fai=fai_text, **side).items():
        open(P(k), "w").write(v)
    common = ["--min-count", "2"]
    # the chain
    subprocess.run([os.path.join(BIN, "generateGraph"), *common, P("t.bam"), P("fastg_fai"), P("c_graph.txt"), f"{avg:.6g}"], check=True)
    subprocess.run([sys.executable, os.path.join(SCRIPTS, "filter_graph.py"), P("fastg_fai"), P("c_graph.txt"), P("c_pre.txt"), f"{avg:.6g}", "0",
                    P("hit_seqs"), P("node_scores"), P("blast"), "0.7", P("fasta_fai"), P("c_hits.txt"), P("contigs_paths"), "0.7"], check=True)
    with open(P("c_filt.txt"), "wb") as f:
        subprocess.run(["uniq", P("c_pre.txt")], check=True, stdout=f)
    subprocess.run([os.path.join(BIN, "matching"), "-g", P("c_filt.txt"), "-r", P("c_lin.txt"), "-c", P("c_cyc.txt"), "-i", "10", "-l", P("contigs_paths")] + flags,
                   check=True)
    subprocess.run([sys.executable, os.path.join(SCRIPTS, "remove_cycle_dup.py"), P("c_cyc.txt"), P("c_nodup.txt")], check=True, stdout=subprocess.DEVNULL)
    open(P("c_all.txt"), "wb").write(open(P("c_lin.txt"), "rb").read() + open(P("c_nodup.txt"), "rb").read())
    # one process
    p = subprocess.run([os.path.join(BIN, "generateGraph"), *common, "--hit-seqs", P("hit_seqs"), "--node-scores", P("node_scores"), "--blast", P("blast"),
                        "--fasta-fai", P("fasta_fai"), "--paths", P("contigs_paths"), "--filtered-pre", P("f_pre.txt"), "--filtered", P("f_filt.txt"),
                        "--all-hit-segs", P("f_hits.txt"), "--linear", P("f_lin.txt"), "--cycle", P("f_cyc.txt"), "--cycle-nodup", P("f_nodup.txt"),
                        "--all-result", P("f_all.txt"), "-i", "10", *flags, P("t.bam"), P("fastg_fai"), P("f_graph.txt"), f"{avg:.6g}"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    for name in ("graph", "pre", "filt", "hits", "lin", "cyc", "nodup", "all"):
        assert open(P(f"f_{name}.txt"), "rb").read() == open(P(f"c_{name}.txt"), "rb").read(), name
    assert open(P("c_all.txt")).read().count("\t") > 5 and open(P("c_pre.txt")).read().count("JUNC") > 5


# ---- hypothesis: the device selection against the script on adversarial graphs ---------------------------------------------------
from hypothesis import HealthCheck, given, settings, strategies as hst


@hst.composite
def filter_cases(draw):
    """small graphs chosen to break the selection: self loops, junctions listed twice and in both directions, seeds that touch
    nothing, chains seed - a - b - c (pass 3 must stop after two hops), paths whose supported share sits at the 0.5 / 2000-base
    edges, paths with ';' and with NODE header lines, a contig on several paths, scores in e-notation and just at the threshold"""
    n = draw(hst.integers(3, 14))
    lens = [draw(hst.sampled_from([56, 100, 999, 1000, 1001, 2000, 2001, 4000])) for _ in range(n)]
    names = ["EDGE_%d_length_%d_cov_%d.5" % (i + 1, lens[i], i + 1) for i in range(n)]
    blast = draw(hst.sets(hst.integers(0, n - 1), max_size=3))
    gene = draw(hst.sets(hst.integers(0, n - 1), max_size=2))
    score = [draw(hst.sampled_from(["0.100", "0.700", "0.701", "0.999", "1e-05", "7.01e-01", "0.5"])) for _ in range(n)]
    order = draw(hst.permutations(range(n)))                              # SEG order in the graph file != fai order
    juncs, keys = [], set()
    for _ in range(draw(hst.integers(0, 18))):
        a, b = draw(hst.integers(0, n - 1)), draw(hst.integers(0, n - 1))
        j = (a, draw(hst.sampled_from("+-")), b, draw(hst.sampled_from("+-")), draw(hst.integers(5, 40)), draw(hst.integers(0, 3)))
        if j[:4] not in keys:                                             # (generateGraph aggregates: one line per oriented pair)
            keys.add(j[:4])
            juncs.append(j)
    if juncs and draw(hst.booleans()):
        juncs.append(juncs[0])                                            # the same JUNC line twice
    paths = []
    for _ in range(draw(hst.integers(0, 5))):
        members = [draw(hst.integers(0, n - 1)) for _ in range(draw(hst.integers(1, 5)))]
        paths.append(",".join("%d%s" % (m + 1, draw(hst.sampled_from("+-"))) for m in members) + draw(hst.sampled_from(["", ";"])))
    return n, lens, names, blast, gene, score, order, juncs, paths


@settings(max_examples=120, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(filter_cases())
def test_device_selection_equals_the_script_on_adversarial_graphs(case):
    import tempfile
    from pathlib import Path
    n, lens, names, blast, gene, score, order, juncs, paths = case
    files = dict(
        fasta_fai="".join("%s\t%d\t0\t60\t61\n" % (names[i], lens[i]) for i in range(n)),
        # one blast hit row per seed: full-length, 99 % identity (cumulative aligned length / length > 0.7)
        blast="".join("%s\tref1\t99.0\t%d\t0\t0\t1\t%d\t1\t%d\t0.0\t100\n" % (names[i], lens[i], lens[i], lens[i]) for i in sorted(blast)),
        hit_seqs="".join(">%s\n" % names[i] for i in sorted(gene)),
        node_scores="".join("%s\t%s\n" % (names[i], score[i]) for i in range(n)),
        contigs_paths="".join("NODE_%d_length_9_cov_1\n%s\n" % (k + 1, p) for k, p in enumerate(paths)),
        graph="".join("SEG %s %g %d\n" % (names[i], 3.5 + i, 1 + i % 3) for i in order) +
              "".join("JUNC %s %s %s %s %d %d\n" % (names[a], oa, names[b], ob, c1, c2) for a, oa, b, ob, c1, c2 in juncs),
    )
    with tempfile.TemporaryDirectory(prefix="palace_s4fuzz_") as d:
        tmp = Path(d)
        for k, v in files.items():
            (tmp / k).write_text(v)
        (tmp / "fastg_fai").write_text("x\t1\t0\t60\t61\n")
        rc = _script_filter().run([str(tmp / "fastg_fai"), str(tmp / "graph"), str(tmp / "pre"), "5.0", "0", str(tmp / "hit_seqs"), str(tmp / "node_scores"),
                                   str(tmp / "blast"), "0.7", str(tmp / "fasta_fai"), str(tmp / "hits"), str(tmp / "contigs_paths"), "0.7"])
        assert rc == 0
        want = (tmp / "pre").read_text().splitlines(keepends=True)
        case_arrays = _load_case(files, tmp)
    # the graph file lists SEG lines in `order`: that is the rank the device orders the selected segments by
    rank = np.empty(n, np.int32)
    rank[list(order)] = np.arange(n, dtype=np.int32)
    case_arrays["rank"] = rank
    with capi.Ctx(0) as ctx:
        st, d_e, d_n = _run_filter(ctx, case_arrays, min_count=0)
        seg_flags, edge_flags = st.flags(len(case_arrays["edges"]))
        st.close()
    flt, nm = case_arrays["flt"], case_arrays["names"]
    got_seg = sorted([flt.seg_line(nm[i], case_arrays["raw_seg"][nm[i]]) for i in np.flatnonzero(seg_flags & 1)] +
                     [case_arrays["raw_seg"][nm[i]].strip() + " 0 1.0 0\n" for i in np.flatnonzero((seg_flags & 3) == 2)])
    jl = case_arrays["junc_lines"]
    got_junc = [l for l, f in zip(jl, edge_flags) if f & 2] + [l for l, f in zip(jl, edge_flags) if (f & 6) == 4]
    assert got_seg == sorted(l for l in want if l.startswith("SEG"))
    assert got_junc == [l for l in want if not l.startswith("SEG")]


@settings(max_examples=50, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(filter_cases(), hst.sampled_from([["-s"], ["-s", "-b"], ["--aggressive"], []]))
def test_device_stage04_equals_the_file_chain_on_adversarial_graphs(case, flags):
    """the same adversarial graphs through selection + decomposition on the device against scripts/filter_graph.py -> uniq ->
    bin/matching on files: SEG order of the filtered graph, linear and cycle files byte for byte (copy numbers 1..3, self
    loops, junctions in both directions, path arcs between kept and dropped contigs)"""
    import tempfile
    from pathlib import Path
    n, lens, names, blast, gene, score, order, juncs, paths = case
    files = dict(
        fasta_fai="".join("%s\t%d\t0\t60\t61\n" % (names[i], lens[i]) for i in range(n)),
        blast="".join("%s\tref1\t99.0\t%d\t0\t0\t1\t%d\t1\t%d\t0.0\t100\n" % (names[i], lens[i], lens[i], lens[i]) for i in sorted(blast)),
        hit_seqs="".join(">%s\n" % names[i] for i in sorted(gene)),
        node_scores="".join("%s\t%s\n" % (names[i], score[i]) for i in range(n)),
        contigs_paths="".join("NODE_%d_length_9_cov_1\n%s\n" % (k + 1, p) for k, p in enumerate(paths)),
        fastg_fai="x\t1\t0\t60\t61\n",
        # (SEG lines in name order, as generateGraph writes them: the device's segment order is the name rank)
        graph="".join("SEG %s %g %d\n" % (names[i], 3.5 + i, 1 + i % 3) for i in sorted(range(n), key=lambda i: names[i].encode())) +
              "".join("JUNC %s %s %s %s %d %d\n" % (names[a], oa, names[b], ob, c1, c2) for a, oa, b, ob, c1, c2 in juncs),
    )
    with tempfile.TemporaryDirectory(prefix="palace_s4fuzz_") as d:
        tmp = Path(d)
        lin, cyc, filt = _file_chain(tmp, files, 5.0, flags)
        case_arrays = _load_case(files, tmp)
    with capi.Ctx(0) as ctx:
        st, d_e, d_n = _run_filter(ctx, case_arrays, min_count=0)
        d_cn = ctx.upload(case_arrays["cn"])
        st.match(d_e.ptr, d_cn.ptr, 10, "--aggressive" in flags, True)
        res, contig_of = st.result()
        got_lin, got_cyc = stage04_io.matching_text(res, contig_of, case_arrays["names"], self_loops="-s" in flags, break_cycles="-b" in flags)
        seg_order = [case_arrays["names"][c] for c in contig_of]
        st.close()
    assert seg_order == [l.split(" ")[1] for l in filt.splitlines() if l.startswith("SEG")]
    assert got_lin == lin
    assert got_cyc == cyc


def test_stage04_paths_with_empty_lines_bad_offsets_and_the_call_s_own_edge_bound():
    """(review of round 3) many empty path lines beside one long line: the path arc table is sized by the token PAIRS, not by
    tokens minus lines; offsets that do not ascend are refused; a later filter call with a SMALLER edge array is held to that
    call's bound, not to the largest bound the object has seen"""
    names = [f"EDGE_{i + 1}_length_{500 + i}_cov_3.0" for i in range(80)]
    n = len(names)
    long_line = np.arange(0, 2 * 60, 2, dtype=np.int32)                       # 60 tokens, 59 consecutive pairs (+ conjugates = 118 arcs)
    off = np.concatenate([np.zeros(101, np.int64), [len(long_line)]])          # 100 empty lines in front of it
    with capi.Ctx(0) as ctx:
        st = capi.Stage04(ctx, np.ones(n, np.uint8), np.full(n, 500, np.int32), stage04_io.name_ranks(names), stage04_io.name_lengths(names), off, long_line)
        e = np.zeros(64, capi.EDGE_DTYPE)
        e["left"] = np.arange(64) % n; e["right"] = (np.arange(64) + 1) % n; e["counts"][:, 0] = 9
        d_e, d_n, d_cn = ctx.upload(e.view(np.uint8).reshape(-1)), ctx.upload(np.array([64], np.int64)), ctx.upload(np.ones(n, np.int32))
        st.filter(d_e.ptr, d_n.ptr, 64)
        st.match(d_e.ptr, d_cn.ptr, 10, False, True)
        res, contig_of = st.result()
        c = st.counts()
        assert c["juncs"] == 64 and c["arcs"] >= 118
        # the same object, a smaller edge array whose device-side count says more than the array holds
        small = ctx.upload(e[:8].view(np.uint8).reshape(-1))
        st.filter(small.ptr, d_n.ptr, 8)                                        # *d_n_edges is still 64 > 8
        with pytest.raises(capi.PalaceError, match="more edges on the device than the bound"):
            st.counts()
        st.close()
        bad = off.copy(); bad[50] = 7                                           # 0 ... 7 0 ...: descends
        with pytest.raises(capi.PalaceError, match="ascend"):
            capi.Stage04(ctx, np.ones(n, np.uint8), np.full(n, 500, np.int32), stage04_io.name_ranks(names), stage04_io.name_lengths(names), bad, long_line)
