"""The CHECKER's chain over one sample's work directory (bench/e2e.py writes it; the executables ran on the same files):

    bam_cols/ (the decoded columns the BAM was written from)  --oracle/graph_oracle.cpp-->  o_graph.txt
    o_graph.txt  --scripts/filter_graph.py, PALACE_FILTER_PY=1 (the implementation pinned to the reference script's outputs)-->  o_pre.txt
    o_pre.txt  --uniq-->  o_filt.txt  --oracle/match_oracle.cpp, -s -i 10 -l contigs.paths-->  linear, cycle
    cycle  --scripts/remove_cycle_dup.py (pinned to the reference script's output)-->  cycle_nodup;  all_result = linear ++ cycle_nodup

Nothing of the product's GPU path or native host code runs here: what comes out is what the files of palace:555-600 must hold.
Test infrastructure (imports oracle/)."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = os.path.join(ROOT, "palace_amd", "scripts")


def load_columns(cols_dir, lo=0, hi=None):
    """the column files of bench/e2e.py write_graph_inputs -> (col dict, sa_off, sa rows, names, lens); [lo, hi) = a record range"""
    f = lambda name, dt: np.fromfile(os.path.join(cols_dir, name), dtype=dt)
    sa_off = f("sa_off.i32", np.int32).astype(np.int64)
    n = len(sa_off) - 1
    hi = n if hi is None else hi
    col = {k: f(k + ".i32", np.int32)[lo:hi] for k in ("tid", "pos", "mtid", "mpos", "nm", "ref_len", "clip_e")}
    col["flag"] = f("flag.u16", np.uint16)[lo:hi]
    col["mapq"] = f("mapq.u8", np.uint8)[lo:hi]
    col["qkey"] = f("qkey.u64", np.uint64)[lo:hi]
    sa = f("sa.i32", np.int32).reshape(-1, 8)
    so = sa_off[lo:hi + 1]
    sa, so = sa[so[0]:max(so[-1], so[0])], so - so[0]
    names, lens = [], []
    with open(os.path.join(cols_dir, "targets.tsv")) as t:
        for line in t:
            a, b = line.rstrip("\n").split("\t")
            names.append(a)
            lens.append(int(b))
    return col, so, sa, names, np.array(lens, dtype=np.int64)


def avg_depth_of(cols_dir, lens):
    """the `<avgDepth>` argument the bench hands to generateGraph: sum of reference-consumed lengths / sum of contig lengths, as %.6g text"""
    total = float(np.fromfile(os.path.join(cols_dir, "ref_len.i32"), dtype=np.int32).astype(np.int64).sum())
    return float(f"{total / lens.sum():.6g}")


def oracle_graph(P, lo=0, hi=None, fastg_fai=None):
    """-> (graph text bytes, names, lens, avg_depth, seconds of the oracle run itself)"""
    from oracle import binding as orc
    col, so, sa, names, lens = load_columns(P["cols"], lo, hi)
    avg = avg_depth_of(P["cols"], lens)
    gin = orc.GraphInput.from_columns(col, so, sa, names, lens)
    t0 = time.perf_counter()
    text = gin.run(fastg_fai or P["fastg_fai"], avg)
    return text, names, lens, avg, time.perf_counter() - t0


def filter_py(P, graph_path, pre_path, hits_path, avg):
    """scripts/filter_graph.py's Python implementation (never the native core) with the arguments palace:568-579 passes"""
    subprocess.run([sys.executable, os.path.join(SCRIPTS, "filter_graph.py"), P["fastg_fai"], graph_path, pre_path, f"{avg:.6g}", "0",
                    P["hit"], P["score"], P["blast"], "0.7", P["fasta_fai"], hits_path, P["paths"], "0.7"],
                   check=True, env=dict(os.environ, PALACE_FILTER_PY="1", PYTHONHASHSEED="0"))


def uniq(src, dst):
    with open(dst, "wb") as f:
        subprocess.run(["uniq", src], check=True, stdout=f)


def oracle_stage04(P, graph_path, out_prefix, avg, self_loops=True, break_cycles=False, aggressive=False, iterations=10):
    """filter -> uniq -> matching -> remove_cycle_dup -> cat on `graph_path`; -> dict of the output texts (bytes)"""
    from oracle import binding as orc
    sys.path.insert(0, SCRIPTS)
    import remove_cycle_dup
    pre, filt, hits = out_prefix + "_pre.txt", out_prefix + "_filt.txt", out_prefix + "_hits.txt"
    filter_py(P, graph_path, pre, hits, avg)
    uniq(pre, filt)
    n_seg = sum(1 for line in open(filt, "rb") if line.startswith(b"SEG"))
    lin, cyc = orc.match_run(filt, P["paths"], iterations, self_loops, break_cycles, aggressive, cap=256 * max(1, n_seg) + (1 << 20))
    nodup = "".join(remove_cycle_dup.dedup_records(cyc.decode().splitlines(keepends=True))).encode()
    return dict(pre=open(pre, "rb").read(), filt=open(filt, "rb").read(), allhit=open(hits, "rb").read(), lin=lin, cyc=cyc, nodup=nodup,
                result=lin + nodup)


def split_graph(text):
    """graph text -> (SEG lines sorted, JUNC lines in order): the SEG block of the reference script is a set iteration"""
    lines = text.splitlines(keepends=True)
    return sorted(l for l in lines if l.startswith(b"SEG")), [l for l in lines if not l.startswith(b"SEG")]
