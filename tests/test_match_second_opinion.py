"""oracle/match_oracle.cpp against tests/match_bruteforce.py (an independent Python statement of the same prose) on every
kind of tiny graph hypothesis finds: <= 12 segments, self loops, self-conjugate junctions, parallel junctions, equal weights.
CPU only; tests/test_gpu_cli.py::test_matching_second_opinion_on_the_gpu repeats it against the executable."""
import os

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import binding as orc
from tests.match_bruteforce import decompose


@st.composite
def tiny_graphs(draw):
    n = draw(st.integers(1, 12))
    names = ["EDGE_%d_length_%d_cov_2.0" % (i + 1, 100 + i) for i in range(n)]
    copies = [draw(st.integers(0, 4)) for _ in range(n)]
    juncs = [(draw(st.integers(0, n - 1)), draw(st.sampled_from("+-")), draw(st.integers(0, n - 1)), draw(st.sampled_from("+-")),
              draw(st.sampled_from([5, 5, 6, 9, 9, 20]))) for _ in range(draw(st.integers(0, 18)))]
    flags = draw(st.sampled_from([(False, False, False), (True, False, False), (False, True, False), (True, True, True), (False, False, True)]))
    return names, copies, juncs, flags


def graph_text(names, copies, juncs):
    return ("".join("SEG %s 1 %d 0 0.000 0\n" % (nm, c) for nm, c in zip(names, copies)) +
            "".join("JUNC %s %s %s %s %d 0\n" % (names[l], a, names[r], b, w) for l, a, r, b, w in juncs))


@settings(max_examples=250, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(tiny_graphs())
def test_oracle_equals_the_bruteforce_statement(tmp_path_factory, g):
    names, copies, juncs, (self_loops, break_cycles, aggressive) = g
    path = str(tmp_path_factory.mktemp("m") / "g.txt")
    open(path, "w").write(graph_text(names, copies, juncs))
    lin, cyc = orc.match_run(path, None, 10, self_loops=self_loops, break_cycles=break_cycles, aggressive=aggressive)
    want_lin, want_cyc = decompose(names, copies, juncs, 10, aggressive, self_loops, break_cycles)
    assert lin.decode() == want_lin
    assert cyc.decode() == want_cyc
