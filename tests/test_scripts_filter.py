"""palace_amd/scripts/{filter_graph,remove_cycle_dup}.py against outputs of the reference's own
scripts (tests/golden/filter_cases.npz, made by tests/golden/make_filter_golden.py).
SEG block: compared as a sorted multiset (the reference's order depends on PYTHONHASHSEED, SURVEY
F4); JUNC block: compared in order; all_hit_segs.txt: byte-identical."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = os.path.join(ROOT, "palace_amd", "scripts")
G = np.load(os.path.join(ROOT, "tests", "golden", "filter_cases.npz"))


def text(key):
    return G[key].tobytes().decode()


def split_blocks(txt):
    lines = txt.splitlines(keepends=True)
    return sorted(l for l in lines if l.startswith("SEG")), [l for l in lines if not l.startswith("SEG")]


NATIVE = os.path.join(ROOT, "palace_amd", "bin", "filter_graph")


@pytest.mark.parametrize("impl", ["python", "native"])
@pytest.mark.parametrize("case", [0, 1, 2])
def test_filter_graph_equals_reference(tmp_path, case, impl):
    if impl == "native":
        assert os.access(NATIVE, os.X_OK), "palace_amd/bin/filter_graph is not built (python -c 'import __graft_entry__ as g; g.build()')"
    P = lambda n: str(tmp_path / n)
    for k in ("graph", "fastg_fai", "fasta_fai", "blast", "hit_seqs", "node_scores", "contigs_paths"):
        open(P(k), "w").write(text(f"case{case}_{k}"))
    args = [P("fastg_fai"), P("graph"), P("pre.txt"), text(f"case{case}_argv_depth"), "0", P("hit_seqs"),
            P("node_scores"), P("blast"), "0.7", P("fasta_fai"), P("all_hit_segs.txt"), P("contigs_paths"), "0.7"]
    for hashseed in ("0", "12345"):                      # output must not depend on the hash seed
        subprocess.run([sys.executable, os.path.join(SCRIPTS, "filter_graph.py")] + args, check=True,
                       env=dict(os.environ, PYTHONHASHSEED=hashseed, PALACE_FILTER_PY="1" if impl == "python" else "0"))
        got_seg, got_rest = split_blocks(open(P("pre.txt")).read())
        want_seg, want_rest = split_blocks(text(f"case{case}_pre"))
        assert got_seg == want_seg
        assert got_rest == want_rest
        assert open(P("all_hit_segs.txt")).read() == text(f"case{case}_all_hit_segs")
        if hashseed == "0":
            first = open(P("pre.txt")).read()
        else:
            assert open(P("pre.txt")).read() == first
    # SEG block comes first, JUNC block second, as consumers expect
    lines = first.splitlines()
    n_seg = sum(l.startswith("SEG") for l in lines)
    assert all(l.startswith("SEG") for l in lines[:n_seg]) and all(l.startswith("JUNC") for l in lines[n_seg:])


def test_remove_cycle_dup_equals_reference(tmp_path):
    src, dst = str(tmp_path / "c.txt"), str(tmp_path / "o.txt")
    open(src, "w").write(text("cycle_in"))
    subprocess.run([sys.executable, os.path.join(SCRIPTS, "remove_cycle_dup.py"), src, dst], check=True,
                   stdout=subprocess.DEVNULL)
    assert open(dst).read() == text("cycle_out")
    open(src, "w").write("")
    subprocess.run([sys.executable, os.path.join(SCRIPTS, "remove_cycle_dup.py"), src, dst], check=True,
                   stdout=subprocess.DEVNULL)
    assert open(dst).read() == ""


def _random_filter_case(rng, n):
    """graph text + side files with the awkward inputs: scientific-notation SEG fields, duplicate SEG lines with other
    values, self loops, duplicate junctions, junctions in front of their SEG lines, rescue through contigs.paths"""
    sys.path.insert(0, ROOT)
    from palace_amd import synth
    lens = rng.integers(200, 9000, size=n)
    names = [f"EDGE_{i + 1}_length_{int(lens[i])}_cov_{rng.random() * 40:.6f}" for i in range(n)]
    side = synth.filter_side_files(rng, names, lens)
    seg = []
    for i, nm in enumerate(names):
        depth = rng.choice([f"{rng.random() * 50:.4f}", f"{rng.random() * 9:.3f}e+01", "1e2", "2.5E-4", "7", "1.23456e+3"])
        seg.append(f"SEG {nm} {depth} {int(rng.integers(1, 5))}\n")
        if rng.random() < 0.03:
            seg.append(f"SEG {nm} {rng.random() * 50:.4f} 9\n")                 # the same name again, other values
    junc = []
    for _ in range(n):
        a, b = (int(x) for x in rng.integers(0, n, size=2))
        if rng.random() < 0.05:
            b = a
        line = f"JUNC {names[a]} {'+-'[int(rng.integers(2))]} {names[b]} {'+-'[int(rng.integers(2))]} {int(rng.integers(1, 30))} {int(rng.integers(0, 9))}\n"
        junc.append(line)
        if rng.random() < 0.05:
            junc.append(line)
    lines = seg + junc
    k = len(lines) // 3
    lines = junc[:3] + lines[:k] + lines[k:]                                     # a few junctions before every SEG line
    return "".join(lines), side


@pytest.mark.parametrize("seed,n", [(1, 40), (2, 400), (3, 4000)])
def test_filter_graph_native_equals_python(tmp_path, seed, n):
    assert os.access(NATIVE, os.X_OK)
    rng = np.random.default_rng(seed)
    graph, side = _random_filter_case(rng, n)
    P = lambda nm: str(tmp_path / nm)
    open(P("graph"), "w").write(graph)
    open(P("fastg_fai"), "w").write("x\t1\n")
    for k, v in side.items():
        open(P(k), "w").write(v)
    outs = {}
    for impl, threads in (("1", "1"), ("0", "1"), ("0", "5")):       # the native core parses in parts on threads
        tag = impl + threads
        args = [P("fastg_fai"), P("graph"), P(f"pre{tag}"), "12.5", "0", P("hit_seqs"), P("node_scores"), P("blast"), "0.7",
                P("fasta_fai"), P(f"hits{tag}"), P("contigs_paths"), "0.7"]
        subprocess.run([sys.executable, os.path.join(SCRIPTS, "filter_graph.py")] + args, check=True,
                       env=dict(os.environ, PALACE_FILTER_PY=impl, PALACE_HOST_THREADS=threads))
        outs[tag] = (open(P(f"pre{tag}"), "rb").read(), open(P(f"hits{tag}"), "rb").read())
    assert outs["11"][0].count(b"SEG") > 3 and outs["11"][0].count(b"JUNC") > 3 and outs["11"][1].count(b"SAMPLE") > 1
    assert outs["01"] == outs["11"] and outs["05"] == outs["11"]


def test_filter_graph_native_rejects_unknown_contig(tmp_path):
    """a junction naming a contig that has no SEG line: the script dies with KeyError, the native core with exit code 1"""
    P = lambda nm: str(tmp_path / nm)
    open(P("graph"), "w").write("SEG EDGE_1_length_500_cov_3.0 10 1\nJUNC EDGE_1_length_500_cov_3.0 + EDGE_2_length_600_cov_3.0 + 5 0\n")
    open(P("fai"), "w").write("EDGE_1_length_500_cov_3.0\t500\t0\t60\t61\nEDGE_2_length_600_cov_3.0\t600\t0\t60\t61\n")
    open(P("hit"), "w").write("EDGE_1_length_500_cov_3.0\t1\n")
    for nm in ("scores", "blast", "paths", "fastg"):
        open(P(nm), "w").write("")
    args = [P("fastg"), P("graph"), P("pre"), "1", "0", P("hit"), P("scores"), P("blast"), "0.7", P("fai"), P("hits"), P("paths"), "0.7"]
    for impl in ("1", "0"):
        r = subprocess.run([sys.executable, os.path.join(SCRIPTS, "filter_graph.py")] + args, capture_output=True,
                           env=dict(os.environ, PALACE_FILTER_PY=impl))
        assert r.returncode == 1
