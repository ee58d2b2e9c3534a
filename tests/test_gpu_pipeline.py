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
        for k in ("eref_rows", "graph_and_components"):
            assert a["config"]["result_digest"][k] == x["config"]["result_digest"][k] is not None
    assert a["config"]["graph"]["n_edges"] > 0


@pytest.mark.parametrize("world,port,scheme,rank0", [(4, 29522, "shard_reads", "1"), (4, 29525, "shard_reads", "0"), (4, 29523, "key_split", "auto"),
                                                    (4, 29526, "shard_counts", "0")])
def test_bench_multi_rank_rehearsal_on_one_gpu(world, port, scheme, rank0):
    """The N-rank step with all ranks on GPU 0 and the collectives over gloo (RCCL refuses two ranks on one device), under each
    Phase-A scheme (replicate: every rank counts all reads; shard_reads: reads sharded, count-table exchange + merge; key_split:
    the key space split, all-gather of the plane slices; shard_counts: reads sharded, partial counts of the DB's probe-index
    entries exchanged and summed, no plane moved; every scheme at two ranks: the all-schemes test below): the same refs and
    the same graph as the single-process run, from the exact candidate gather (first step) and from the padded one (later
    steps); and the weak record (every rank the whole one-GPU step) carries the same digest."""
    size = ["--contigs", "20000", "--refs", "200", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--soak-seconds", "0"]
    a = json.loads(sh([sys.executable, os.path.join(ROOT, "bench.py")] + size).decode().strip().splitlines()[-1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + size
    out = sh(cmd, env=dict(os.environ, PALACE_BENCH_ONE_DEVICE="1", PALACE_BENCH_BACKEND="gloo", PALACE_BENCH_SCHEME=scheme,
                           PALACE_BENCH_RANK0_READS=rank0)).decode()
    b = json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == world and b["scaling"] == "strong" and "failed_checks" not in b
    pm = b["config"]["parallelism_model"]
    assert set(pm["ms"]) == {"replicate", "key_split", "shard_reads"} and pm["world"] == world
    assert pm["choice_in_force"] == scheme and pm["step"]["scheme"] == scheme and pm["step"]["step_ms"] > 0
    if rank0 in ("0", "1"):
        assert pm["rank0_counts"] == (rank0 == "1") and ("rank 0 takes no reads" in b["config"]["parallelism"]) == (rank0 == "0")
    assert ("reads/records/refs sharded" in b["config"]["parallelism"]) == (scheme in ("shard_reads", "shard_counts"))
    assert ("partial counts of the DB's probe-index entries" in b["config"]["parallelism"]) == (scheme == "shard_counts")
    assert ("key space sharded" in b["config"]["parallelism"]) == (scheme == "key_split")
    # the plane crossed the "links" as counts + 16-bit keys in every step but the first (which sizes the room)
    assert (pm["sparse_gather"]["steps"] >= 2 and pm["sparse_gather"]["cap_keys_per_rank"] > 0) == (scheme in ("key_split", "shard_reads"))
    assert a["config"]["refs_reported"] == b["config"]["refs_reported"] > 0
    assert a["config"]["graph"] == b["config"]["graph"]
    assert a["config"]["result_digest"]["eref_rows"] == b["config"]["result_digest"]["eref_rows"]
    assert a["config"]["result_digest"]["graph_and_components"] == b["config"]["result_digest"]["graph_and_components"] is not None
    w = b["weak"]
    assert w["scaling"] == "weak" and w["n_gpus"] == world and w["value"] == pytest.approx(world * 20000 / (w["ms_per_step"] * 1e-3))
    assert w["result_digest"]["eref_rows"] == a["config"]["result_digest"]["eref_rows"]
    assert w["result_digest"]["graph_and_components"] == a["config"]["result_digest"]["graph_and_components"]


SIZE_SMALL = ["--contigs", "20000", "--refs", "200", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--soak-seconds", "0"]


def run_ranks(world, port, env_add, expect_rc=0):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + SIZE_SMALL
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, PALACE_BENCH_ONE_DEVICE="1", PALACE_BENCH_BACKEND="gloo", **env_add))
    lines = [l for l in p.stdout.decode().strip().splitlines() if l.startswith("{")]
    assert lines, p.stderr.decode()[-3000:]
    assert (p.returncode == 0) == (expect_rc == 0), (p.returncode, p.stderr.decode()[-3000:])
    return json.loads(lines[-1])


def test_bench_measures_every_scheme_when_none_is_forced():
    """`bench.py --gpus N` without a forced scheme (what the driver's scaling run starts): every Phase-A scheme that exists for N ranks
    is measured in a child process per rank, each held against the one-GPU step's digests (the weak leg's); `value` is the fastest
    valid one's and every scheme's time and verdict is in the line.  Two ranks on one GPU over gloo: all four schemes valid."""
    a = json.loads(sh([sys.executable, os.path.join(ROOT, "bench.py")] + SIZE_SMALL).decode().strip().splitlines()[-1])
    b = run_ranks(2, 29531, {})
    pm = b["parallelism_measured"]
    assert set(pm) == {"replicate", "key_split", "shard_reads", "shard_counts"} and "failed_checks" not in b
    assert all(v["valid"] is True and v["status"] == "ok" and v["ms_per_step"] > 0 for v in pm.values())
    best = min(pm, key=lambda k: pm[k]["ms_per_step"])
    assert b["ms_per_step"] == pm[best]["ms_per_step"] and best in b["config"]["parallelism"] and b["n_gpus"] == 2 and b["steps"] == 2
    assert b["value"] == pytest.approx(20000 / (b["ms_per_step"] * 1e-3))
    for k in ("eref_rows", "graph_and_components"):
        assert a["config"]["result_digest"][k] == b["config"]["result_digest"][k] == b["weak"]["result_digest"][k] is not None
    assert b["weak"]["scaling"] == "weak" and b["weak"]["n_gpus"] == 2


def test_bench_reports_a_scheme_that_hangs_under_its_name_and_still_measures_the_others():
    """one scheme's last rank never joins a collective: its children are killed at the time limit, it is a failed check under its name
    (exit status 3), and the line carries the fastest of the others"""
    b = run_ranks(2, 29541, {"PALACE_BENCH_ONLY_SCHEMES": "shard_reads,shard_counts", "PALACE_BENCH_TEST_HANG": "shard_reads", "PALACE_BENCH_SCHEME_TIMEOUT": "40"}, expect_rc=3)
    pm = b["parallelism_measured"]
    assert set(pm) == {"shard_reads", "shard_counts"}
    assert pm["shard_counts"]["valid"] is True and pm["shard_reads"]["valid"] is False and "killed" in pm["shard_reads"]["status"]
    assert len(b["failed_checks"]) == 1 and "shard_reads" in b["failed_checks"][0]
    assert "shard_counts" in b["config"]["parallelism"] and b["n_gpus"] == 2
    assert b["config"]["result_digest"]["eref_rows"] == b["weak"]["result_digest"]["eref_rows"]


def test_bench_all_schemes_at_four_ranks_include_rank_0_idle():
    """four ranks: the read-sharded schemes are measured with rank 0 counting AND with rank 0 taking no reads (stage 04 then has its device to itself)"""
    b = run_ranks(4, 29551, {"PALACE_BENCH_ONLY_SCHEMES": "shard_counts"})
    pm = b["parallelism_measured"]
    assert set(pm) == {"shard_counts", "shard_counts, rank 0 idle in Phase A"} and all(v["valid"] for v in pm.values()) and "failed_checks" not in b


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
    for k in ("eref_rows", "graph_and_components"):
        assert a["config"]["result_digest"][k] == b["config"]["result_digest"][k] is not None


def test_c_abi_table_exchange_on_a_one_rank_communicator():
    """include/palace_rccl.h driven from C++ (palace_amd/host/exchange_selftest_main.cpp): count -> pack -> grouped
    send/recv -> merge -> all-gather on an ncclComm_t of one rank leaves the table as it was, the rows broadcast too."""
    out = sh([os.path.join(BIN, "exchange_selftest"), "20000"]).decode()
    assert out.strip().splitlines()[-1].startswith("ok:"), out           # (RCCL prints its version banner first)


