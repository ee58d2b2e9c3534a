"""`generateGraph --debug`'s per-read stderr text (generate_graph.cpp:454-458, :607-609, :711-853): the host-side generator of the
executable (palace_amd/host/debug_trace.hpp, run here through `hostdump bamtrace`, no GPU) against the oracle's
(oracle/graph_oracle.cpp, orc_graph_run_trace), byte for byte -- on random graph cases and on the adversarial records of the
generateGraph fuzz (one name on many records, SA lists with malformed / unknown / non-stitching items, empty CIGARs, every flag
combination, the option sets).  Both are restatements of the reference's text ("parity unpinned": generate_graph.cpp needs htslib)."""
import os
import subprocess
import tempfile

from hypothesis import HealthCheck, given, settings

from oracle import binding as orc
from palace_amd import synth
from tests.test_gpu_graph_fuzz import OPTS, cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTDUMP = os.path.join(ROOT, "palace_amd", "bin", "hostdump")


def product_trace(bam, fai, o):
    return subprocess.run([HOSTDUMP, "bamtrace", bam, fai, str(o.max_end), str(o.min_mapq), str(o.max_nm), str(o.enable_paired),
                           repr(o.max_span_frac)], stdout=subprocess.PIPE, check=True).stdout


def test_debug_trace_of_random_graph_cases(tmp_path):
    for seed in (77, 5, 9):
        rng = synth.rng_for(seed)
        targets, fai_text, recs, avg = synth.random_graph_case(rng, 80, 9000)
        bam, fai = str(tmp_path / "s.bam"), str(tmp_path / "g.fai")
        synth.write_bam(bam, targets, recs)
        open(fai, "w").write(fai_text)
        o = orc.graph_default_opts()
        graph, want = orc.graph_trace(recs, targets, fai, avg, o)
        assert graph == orc.graph_run(recs, targets, fai, avg, o)                  # (tracing changes nothing of the graph)
        assert want.count(b"=== Split-read: ") > 50 and want.count(b"  -> Passed eval with score=") > 20 and want.count(b"Score calculation: ") > 100
        assert product_trace(bam, fai, o) == want


@settings(max_examples=150, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(cases())
def test_debug_trace_of_adversarial_records(case):
    targets, fai_text, recs, extra = case
    with tempfile.TemporaryDirectory(prefix="palace_trace_") as d:
        bam, fai = os.path.join(d, "t.bam"), os.path.join(d, "g.fastg.fai")
        synth.write_bam(bam, targets, recs, block=700)
        open(fai, "w").write(fai_text)
        o = orc.graph_default_opts()
        o.min_count = 1
        for k, v in OPTS[extra].items():
            setattr(o, k, v)
        _, want = orc.graph_trace(recs, targets, fai, 1.0, o)
        assert product_trace(bam, fai, o) == want
