"""oracle/graph_oracle.cpp against hand-derived expectations (no reference-run golden exists for
generateGraph: htslib is absent, see the oracle's header)."""
import dataclasses
import os

from oracle import binding as orc
from tests import graph_cases as gc


def run(records, tmp_path, targets=gc.TARGETS, fai=gc.FASTG_FAI, avg=gc.AVG_DEPTH, **opt):
    p = tmp_path / "g.fastg.fai"
    p.write_text(fai)
    o = orc.graph_default_opts()
    for k, v in opt.items():
        setattr(o, k, v)
    return orc.graph_run(records, targets, str(p), avg, o)


def test_hand_case(tmp_path):
    assert run(gc.records(), tmp_path) == gc.EXPECTED


def test_debug_lists_the_supporting_reads(tmp_path):
    assert run(gc.records(), tmp_path, debug=1) == gc.EXPECTED_DEBUG


def test_min_count_and_flags(tmp_path):
    recs = gc.records()
    out = run(recs, tmp_path, min_count=6)
    assert b"JUNC ctgA + ctgC" not in out and b"JUNC ctgA + ctgB + 7 0" in out
    # secondary / supplementary / unmapped records contribute nothing, not even depth
    extra = [dataclasses.replace(recs[0], flag=f, qname=f"x{f}") for f in (0x100, 0x800, 0x4)]
    assert run(extra + recs, tmp_path) == gc.EXPECTED


def test_mapq_zero_kills_score_but_still_marks_pair(tmp_path):
    recs = [dataclasses.replace(r, mapq=0) if r.qname.startswith("p") and r.tid == 0 else r for r in gc.records()]
    out = run(recs, tmp_path)
    assert b"JUNC ctgA + ctgC" not in out
    assert b"SEG ctgA 1.36 3" in out          # the mates still hit the processed-pairs quirk


def test_long_contig_underflow_gate(tmp_path):
    """'-' orientation measures distance to the far end: on a 120 kb contig exp() underflows to 0."""
    targets = [("ctgA", 120000), ("ctgB", 2000)]
    fai = "ctgA;\t1\n"
    from palace_amd.synth import BamRecord
    mk = lambda i, L: BamRecord(f"t{i}", 0x10, 1, 4, 60, "40S60M", nm=0, sa=f"ctgA,{L - 100},-,60S40M,60,0;")
    out_long = run([mk(i, 120000) for i in range(6)], tmp_path, targets, fai)
    assert b"JUNC" not in out_long
    out_short = run([mk(i, 50000) for i in range(6)], tmp_path, [("ctgA", 50000), ("ctgB", 2000)], fai)
    assert b"JUNC ctgA + ctgB + 6 0" in out_short


def test_depth_stage_hand_case():
    text, s, nr = orc.depth_mean(gc.depth_records(), gc.DEPTH_TARGETS)
    assert (s, nr, text) == (gc.DEPTH_SUM, gc.DEPTH_NR, gc.DEPTH_TEXT)
    # an integral mean is printed the way awk prints integers
    from palace_amd.synth import BamRecord
    text, s, nr = orc.depth_mean([BamRecord("a", 0, 0, 0, 60, "10M"), BamRecord("b", 0, 0, 0, 60, "10M")], [("c", 100)])
    assert (text, s, nr) == ("2", 20, 10)
    assert orc.depth_mean([BamRecord("u", 4, -1, -1, 0, "")], [("c", 100)])[0] is None


import itertools

import pytest

BOOLS = list(itertools.product([False, True], repeat=4))


def junc_lines(out):
    return "".join(l + "\n" for l in out.decode().splitlines() if l.startswith("JUNC"))


@pytest.mark.parametrize("swap", [False, True])
@pytest.mark.parametrize("rev1,end1,rev2,end2", BOOLS)
def test_paired_layout_every_orientation_and_region(tmp_path, rev1, end1, rev2, end2, swap):
    recs, want = gc.paired_case(rev1, end1, rev2, end2, swap)
    out = run(recs, tmp_path, gc.LAYOUT_TARGETS, "", 1.0)
    assert junc_lines(out) == (want or "")


@pytest.mark.parametrize("swap", [False, True])
@pytest.mark.parametrize("primary_first", [True, False])
@pytest.mark.parametrize("rev_p,end_p,rev_s,end_s", BOOLS)
def test_split_layout_every_orientation_and_region(tmp_path, rev_p, end_p, rev_s, end_s, primary_first, swap):
    recs, want = gc.split_case(rev_p, end_p, rev_s, end_s, primary_first, swap)
    out = run(recs, tmp_path, gc.LAYOUT_TARGETS, "", 1.0)
    assert junc_lines(out) == (want or "")
    assert sum(w is not None for w in [want]) in (0, 1)


def test_layout_tables_have_four_valid_rows_each():
    assert sum(gc.paired_case(*b)[1] is not None for b in BOOLS) == 4
    assert sum(gc.split_case(*b)[1] is not None for b in BOOLS) == 4
    assert sum(gc.split_case(*b, primary_first=False)[1] is not None for b in BOOLS) == 4


def test_columns_marshalling_equals_the_record_route_and_the_chain_runs_on_cpu(tmp_path):
    """GraphInput.from_columns (numpy, used at the bench workloads' full size) hands orc_graph_run the same records as the
    per-record route through BamRecord texts; and the checker's chain over a work directory (tests/oracle_chain.py) gives the
    native filter core's files on its graph (host logic only: no GPU in this test)."""
    import subprocess
    import sys

    import torch

    import bench
    from tests import oracle_chain as oc
    from bench import e2e
    from palace_amd.synth import BamRecord
    n_contigs, n_pairs = 20_000, 66_666
    gs = bench.make_graph_sample(torch, torch.device("cpu"), n_contigs, n_pairs)
    gs["side"] = bench.make_side_inputs(gs)
    P = e2e.e2e_paths(str(tmp_path))
    e2e.write_graph_inputs(gs, P, bam=False)
    text, names, lens, avg, _ = oc.oracle_graph(P)
    assert avg == gs["avg_depth"] and names == gs["names"]
    c = {k: v.numpy() for k, v in gs["col"].items()}
    so, sa = gs["sa_off"].numpy(), gs["sa"].numpy()
    recs = []
    for i in range(gs["n"]):
        s_txt = None
        if so[i + 1] > so[i]:
            it = sa[so[i]]
            s_txt = f"{names[it[0]]},{it[1]},{'-' if it[7] else '+'},{it[4]}S{it[6] - it[4]}M,{it[2]},{it[3]};"
        cig = f"{c['ref_len'][i]}M{c['clip_e'][i]}S" if c["clip_e"][i] else "150M"
        recs.append(BamRecord(f"q{int(c['qkey'][i]) & 0xffffffffffff:x}", int(c["flag"][i]) & 0xffff, int(c["tid"][i]), int(c["pos"][i]),
                              int(c["mapq"][i]), cig, int(c["mtid"][i]), int(c["mpos"][i]), nm=int(c["nm"][i]), sa=s_txt))
    want = orc.graph_run(recs, list(zip(names, lens.tolist())), P["fastg_fai"], avg)
    assert text == want and text.count(b"JUNC ") > 100
    part, *_ = oc.oracle_graph(P, 1000, 31000)                          # a record range marshals the same way
    assert part == orc.graph_run(recs[1000:31000], list(zip(names, lens.tolist())), P["fastg_fai"], avg)
    # the chain: Python filter (pinned to the reference script's goldens) == the native core, then matching + remove_cycle_dup
    o_graph = str(tmp_path / "o_graph.txt")
    open(o_graph, "wb").write(text)
    out = oc.oracle_stage04(P, o_graph, str(tmp_path / "o"), avg)
    core = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "palace_amd", "bin", "filter_graph")
    if os.path.exists(core):
        subprocess.run([sys.executable, os.path.join(oc.SCRIPTS, "filter_graph.py"), P["fastg_fai"], o_graph, str(tmp_path / "n_pre.txt"), f"{avg:.6g}", "0",
                        P["hit"], P["score"], P["blast"], "0.7", P["fasta_fai"], str(tmp_path / "n_hits.txt"), P["paths"], "0.7"], check=True,
                       env={k: v for k, v in os.environ.items() if k != "PALACE_FILTER_PY"})
        assert open(tmp_path / "n_pre.txt", "rb").read() == out["pre"] and open(tmp_path / "n_hits.txt", "rb").read() == out["allhit"]
    assert out["filt"].count(b"SEG ") > 1000 and out["filt"].count(b"JUNC ") > 50 and out["result"].count(b"\t") > 100
    assert out["result"] == out["lin"] + out["nodup"]
