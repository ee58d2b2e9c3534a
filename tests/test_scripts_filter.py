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


@pytest.mark.parametrize("case", [0, 1, 2])
def test_filter_graph_equals_reference(tmp_path, case):
    P = lambda n: str(tmp_path / n)
    for k in ("graph", "fastg_fai", "fasta_fai", "blast", "hit_seqs", "node_scores", "contigs_paths"):
        open(P(k), "w").write(text(f"case{case}_{k}"))
    args = [P("fastg_fai"), P("graph"), P("pre.txt"), text(f"case{case}_argv_depth"), "0", P("hit_seqs"),
            P("node_scores"), P("blast"), "0.7", P("fasta_fai"), P("all_hit_segs.txt"), P("contigs_paths"), "0.7"]
    for hashseed in ("0", "12345"):                      # output must not depend on the hash seed
        subprocess.run([sys.executable, os.path.join(SCRIPTS, "filter_graph.py")] + args, check=True,
                       env=dict(os.environ, PYTHONHASHSEED=hashseed))
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
