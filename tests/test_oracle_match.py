"""Structural checks of the matching oracle (own algorithm; the reference's is absent):
output grammar, conjugate-closure of every emitted adjacency, copy-number budget, determinism."""
import os
import re

import numpy as np

from oracle import binding as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "filter_cases.npz"))
TOK = re.compile(r"^EDGE_\d+_length_\d+_cov_[0-9.]+[+-]$")


def write_case(tmp_path, case):
    g, p = str(tmp_path / "g.txt"), str(tmp_path / "contigs.paths")
    open(g, "w").write(G[f"case{case}_pre"].tobytes().decode())
    open(p, "w").write(G[f"case{case}_contigs_paths"].tobytes().decode())
    return g, p


def arcs_of(graph_path, paths_path):
    ids, arcs = {}, set()
    flip = {"+": "-", "-": "+"}
    for ln in open(graph_path):
        t = ln.split()
        if t and t[0] == "SEG":
            ids[t[1].split("_")[1]] = t[1]
    for ln in open(graph_path):
        t = ln.split()
        if t and t[0] == "JUNC":
            arcs.add((t[1] + t[2], t[3] + t[4]))
            arcs.add((t[3] + flip[t[4]], t[1] + flip[t[2]]))          # make_final_fa.py:20-34
            for n in (t[1], t[3]):
                ids.setdefault(n.split("_")[1], n)
    for ln in open(paths_path):
        if ln.startswith("NODE"):
            continue
        toks = [x.strip().rstrip(";") for x in ln.strip().split(",")]
        for a, b in zip(toks, toks[1:]):
            if a[:-1] in ids and b[:-1] in ids:
                arcs.add((ids[a[:-1]] + a[-1], ids[b[:-1]] + b[-1]))
                arcs.add((ids[b[:-1]] + flip[b[-1]], ids[a[:-1]] + flip[a[-1]]))
    return arcs


def test_grammar_and_adjacency(tmp_path):
    for case in (0, 1, 2):
        g, p = write_case(tmp_path, case)
        lin, cyc = orc.match_run(g, p, 10, self_loops=True)
        arcs = arcs_of(g, p)
        segs = {ln.split()[1] for ln in open(g) if ln.startswith("SEG")}
        assert lin and not any(l.startswith((b"iter", b"self")) for l in lin.splitlines())
        used = set()
        for ln in lin.decode().splitlines():
            toks = ln.split("\t")
            assert all(TOK.match(t) for t in toks)
            used.update(t[:-1] for t in toks)
            for a, b in zip(toks, toks[1:]):
                assert (a, b) in arcs
        cl = cyc.decode().splitlines()
        assert len(cl) % 2 == 0                                   # two-line records (remove_cycle_dup.py:9-13)
        for head, body in zip(cl[0::2], cl[1::2]):
            assert head.startswith("iter ") or head == "self"
            toks = body.split("\t")
            for a, b in zip(toks, toks[1:] + toks[:1]):
                assert (a, b) in arcs
            used.update(t[:-1] for t in toks)
        assert segs <= used                                        # every segment is reported at least once
        assert (lin, cyc) == orc.match_run(g, p, 10, self_loops=True)


def test_options_change_output_consistently(tmp_path):
    g, p = write_case(tmp_path, 1)
    base = orc.match_run(g, p, 10)
    brk = orc.match_run(g, p, 10, break_cycles=True)
    assert brk[1] == base[1] and len(brk[0]) >= len(base[0])
    one = orc.match_run(g, p, 1)
    assert len(one[0]) <= len(base[0])
    nop = orc.match_run(g, None, 10)
    assert nop[0]
