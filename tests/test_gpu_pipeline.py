"""Whole 04-match stage through the drop-in tools on a GPU (palace:555-600), checked against the
oracle chain; plus bench.py's multi-rank exchange code path rehearsed on one GPU."""
import json
import os
import subprocess
import sys

import pytest

from oracle import binding as orc
from palace_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "palace_amd", "bin")
SCRIPTS = os.path.join(ROOT, "palace_amd", "scripts")


def sh(cmd, **kw):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout


def test_stage4_chain_equals_oracle_chain(tmp_path):
    rng = synth.rng_for(77)
    targets, fai_text, recs, avg = synth.random_graph_case(rng, 80, 9000)
    names, lens = [t[0] for t in targets], [t[1] for t in targets]
    side = synth.filter_side_files(rng, names, lens)
    P = lambda n: str(tmp_path / n)
    synth.write_bam(P("s.bam"), targets, recs)
    for k, v in dict(fastg_fai=fai_text, **side).items():
        open(P(k), "w").write(v)
    depth = f"{avg:.6g}"
    # 4.3 generateGraph (min-count 3 so the toy keeps a few dozen junctions)
    sh([os.path.join(BIN, "generateGraph"), "--min-count", "3", P("s.bam"), P("fastg_fai"), P("graph.txt"), depth])
    o = orc.graph_default_opts(); o.min_count = 3
    want_graph = orc.graph_run(recs, targets, P("fastg_fai"), float(depth), o)
    assert open(P("graph.txt"), "rb").read() == want_graph
    # 4.4 filter + uniq
    sh([sys.executable, os.path.join(SCRIPTS, "filter_graph.py"), P("fastg_fai"), P("graph.txt"), P("pre.txt"), depth, "0",
        P("hit_seqs"), P("node_scores"), P("blast"), "0.7", P("fasta_fai"), P("all_hit_segs.txt"), P("contigs_paths"), "0.7"])
    open(P("filtered.txt"), "wb").write(sh(["uniq", P("pre.txt")]))
    assert open(P("filtered.txt")).read().count("JUNC") > 5
    # 4.5 matching + remove_cycle_dup + cat
    sh([os.path.join(BIN, "matching"), "-g", P("filtered.txt"), "-r", P("linear.txt"), "-c", P("cycle.txt"), "-s", "-i", "10",
        "-l", P("contigs_paths")])
    sh([sys.executable, os.path.join(SCRIPTS, "remove_cycle_dup.py"), P("cycle.txt"), P("cycle_nodup.txt")])
    all_result = open(P("linear.txt"), "rb").read() + open(P("cycle_nodup.txt"), "rb").read()
    lin, cyc = orc.match_run(P("filtered.txt"), P("contigs_paths"), 10, self_loops=True)
    assert open(P("linear.txt"), "rb").read() == lin and open(P("cycle.txt"), "rb").read() == cyc
    assert all_result.startswith(lin) and len(lin) > 0


def test_bench_exchange_path_rehearsal():
    """world_size 1 over RCCL: same results as the plain single-GPU run."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--contigs", "20000", "--refs", "200", "--steps", "1",
            "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--soak-seconds", "0"]
    a = json.loads(sh(base).decode().strip().splitlines()[-1])
    b = json.loads(sh(base, env=dict(os.environ, PALACE_FORCE_EXCHANGE="1")).decode().strip().splitlines()[-1])
    c = json.loads(sh(base, env=dict(os.environ, PALACE_FORCE_KEY_SPLIT="1")).decode().strip().splitlines()[-1])     # the key-space split's gather over RCCL
    for x in (b, c):
        assert a["config"]["refs_reported"] == x["config"]["refs_reported"] > 0
        assert a["config"]["graph"] == x["config"]["graph"]
        assert a["config"]["result_digest"] == x["config"]["result_digest"] and a["config"]["result_digest"]["graph_and_components"]
    assert a["config"]["graph"]["n_edges"] > 0


@pytest.mark.parametrize("world,port,shard", [(2, 29521, "0"), (4, 29522, "1"), (4, 29523, "0")])
def test_bench_multi_rank_rehearsal_on_one_gpu(world, port, shard):
    """The N-rank step (world 2: every rank counts all reads, Phase B / generateGraph / gathers sharded; world 4: reads
    sharded too, count-table exchange + merge) with all ranks on GPU 0 and the collectives over gloo (RCCL refuses two
    ranks on one device): the same refs and the same graph as the single-process run."""
    size = ["--contigs", "20000", "--refs", "200", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--soak-seconds", "0"]
    a = json.loads(sh([sys.executable, os.path.join(ROOT, "bench.py")] + size).decode().strip().splitlines()[-1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + size
    # (two flavours of four ranks: the reads sharded with the count-table exchange -- forced here, the bench itself never picks it --
    # and of four: the key space split with an all-gather of the plane slices -- what the bench does from four ranks on)
    out = sh(cmd, env=dict(os.environ, PALACE_BENCH_ONE_DEVICE="1", PALACE_BENCH_BACKEND="gloo", PALACE_BENCH_SHARD_READS=shard)).decode()
    b = json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == world
    assert ("reads/records/refs sharded" in b["config"]["parallelism"]) == (shard == "1")
    assert ("key space sharded" in b["config"]["parallelism"]) == (world == 4 and shard == "0")
    assert a["config"]["refs_reported"] == b["config"]["refs_reported"] > 0
    assert a["config"]["graph"] == b["config"]["graph"]
    assert a["config"]["result_digest"] == b["config"]["result_digest"] and a["config"]["result_digest"]["graph_and_components"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the ranks are started as child processes (before the parent
    touches a GPU), rank 0's JSON line comes through and the exit status is theirs.  Rehearsed on one GPU over gloo."""
    size = ["--contigs", "20000", "--refs", "200", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--soak-seconds", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PALACE_BENCH_ONE_DEVICE="1", PALACE_BENCH_BACKEND="gloo")
    a = json.loads(sh([sys.executable, os.path.join(ROOT, "bench.py")] + size).decode().strip().splitlines()[-1])
    out = sh([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + size, env=env).decode()
    b = json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2
    assert a["config"]["refs_reported"] == b["config"]["refs_reported"] > 0
    assert a["config"]["graph"] == b["config"]["graph"]
    assert a["config"]["result_digest"] == b["config"]["result_digest"] and a["config"]["result_digest"]["graph_and_components"]


def test_c_abi_table_exchange_on_a_one_rank_communicator():
    """include/palace_rccl.h driven from C++ (palace_amd/host/exchange_selftest_main.cpp): count -> pack -> grouped
    send/recv -> merge -> all-gather on an ncclComm_t of one rank leaves the table as it was, the rows broadcast too."""
    out = sh([os.path.join(BIN, "exchange_selftest"), "20000"]).decode()
    assert out.strip().splitlines()[-1].startswith("ok:"), out           # (RCCL prints its version banner first)


def test_bench_two_batches_in_flight_gives_the_same_results():
    """--batches-in-flight 2 (double-buffered count table, rows fetched a step late) reports what the plain step reports"""
    size = ["--contigs", "20000", "--refs", "200", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--soak-seconds", "0"]
    a = json.loads(sh([sys.executable, os.path.join(ROOT, "bench.py")] + size).decode().strip().splitlines()[-1])
    b = json.loads(sh([sys.executable, os.path.join(ROOT, "bench.py"), "--batches-in-flight", "2"] + size).decode().strip().splitlines()[-1])
    assert b["config"]["batches_in_flight"] == 2 and a["config"]["batches_in_flight"] == 1
    assert a["config"]["refs_reported"] == b["config"]["refs_reported"] > 0
    assert a["config"]["result_digest"] == b["config"]["result_digest"]
