#!/usr/bin/env python3
"""Generate tests/golden/filter_cases.npz by RUNNING the reference's Python glue
(share/palace/scripts/filter_graph.py and remove_cycle_dup.py, pure stdlib) on seeded synthetic
inputs, with PYTHONHASHSEED=0.  Build-container only; the GPU box reads the committed .npz.
Stored: the input file texts and the output bytes -- no reference source text."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import binding as orc  # noqa: E402
from palace_amd import synth  # noqa: E402

REF = "/root/reference/share/palace/scripts"


def one_case(seed, n_contigs, n_events):
    rng = synth.rng_for(seed)
    targets, fai_text, recs, avg = synth.random_graph_case(rng, n_contigs, n_events)
    names = [t[0] for t in targets]
    lens = [t[1] for t in targets]
    d = tempfile.mkdtemp(prefix="palace_filter_")
    P = lambda n: os.path.join(d, n)
    open(P("g.fastg.fai"), "w").write(fai_text)
    o = orc.graph_default_opts()
    o.min_count = 3
    graph = orc.graph_run(recs, targets, P("g.fastg.fai"), avg, o).decode().splitlines(keepends=True)
    # a few depths in scientific notation, as `ostream << double` prints tiny / huge values
    sci = ["2.5e-05", "1e+06", "1.5e+02", "7.25e-01", "3e-07"]
    k = 0
    for i, ln in enumerate(graph):
        if ln.startswith("SEG") and i % 7 == 3:
            c = ln.split(" ")
            c[2] = sci[k % len(sci)]
            k += 1
            graph[i] = " ".join(c)
    junc = [ln for ln in graph if ln.startswith("JUNC")]
    if junc:                                       # a duplicated JUNC line and a self loop
        graph.append(junc[0])
        c = junc[0].split(" ")
        graph.append(f"JUNC {c[1]} + {c[1]} - 9 0\n")
    side = synth.filter_side_files(rng, names, lens)
    files = dict(graph="".join(graph), fastg_fai=fai_text, **side)
    for k_, v in files.items():
        open(P(k_), "w").write(v)
    env = dict(os.environ, PYTHONHASHSEED="0")
    args = [P("fastg_fai"), P("graph"), P("pre.txt"), f"{avg:.6g}", "0", P("hit_seqs"), P("node_scores"), P("blast"),
            "0.7", P("fasta_fai"), P("all_hit_segs.txt"), P("contigs_paths"), "0.7"]
    subprocess.run([sys.executable, os.path.join(REF, "filter_graph.py")] + args, check=True, env=env)
    out = dict(files)
    out["argv_depth"] = f"{avg:.6g}"
    out["pre"] = open(P("pre.txt")).read()
    out["all_hit_segs"] = open(P("all_hit_segs.txt")).read()
    return out


def cycle_case(seed):
    rng = synth.rng_for(seed)
    recs = [(f"iter {i % 3}\n", f"EDGE_{int(rng.integers(1, 9))}_length_100_cov_2.0+\tEDGE_7_length_50_cov_1.0-\n")
            for i in range(25)]
    lines = [x for r in recs for x in r] + ["self\n"]            # odd number of lines
    d = tempfile.mkdtemp(prefix="palace_cycle_")
    open(os.path.join(d, "c.txt"), "w").write("".join(lines))
    subprocess.run([sys.executable, os.path.join(REF, "remove_cycle_dup.py"), os.path.join(d, "c.txt"),
                    os.path.join(d, "o.txt")], check=True, stdout=subprocess.DEVNULL)
    return dict(cycle_in="".join(lines), cycle_out=open(os.path.join(d, "o.txt")).read())


def main():
    blob = {}
    for i, (seed, nc, ne) in enumerate([(101, 40, 3000), (102, 120, 9000), (103, 15, 1500)]):
        for k, v in one_case(seed, nc, ne).items():
            blob[f"case{i}_{k}"] = np.frombuffer(v.encode(), dtype=np.uint8)
    for k, v in cycle_case(7).items():
        blob[k] = np.frombuffer(v.encode(), dtype=np.uint8)
    blob["empty_cycle_out"] = np.zeros(0, dtype=np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "filter_cases.npz"), **blob)
    print({k: len(v) for k, v in blob.items()})


if __name__ == "__main__":
    main()
