"""The bench workloads at their own size (driver-visible: `pytest -m gpu`).

  configs[2]  500k contigs: `bench.py --contigs 500000` must exit 0 -- it exits non-zero when its own cross-checks fail (the
              executables on the generated files disagreeing with the HBM-resident step, the fused generateGraph process
              writing other files than the five-process chain, results that differ from step to step) -- and every read of
              the workload counted through the packed entry (the bench's data path) equals the oracle's table.
  headline    the 1M-contig workload BASELINE.json's metric is quoted on: the same, plus every present ref reported.
  configs[4]  the long-contig workload (100k contigs, N50 ~ 50 kb, 6.67 M records) through the same run.

And at each of those sizes the files the EXECUTABLES wrote for the sample (`_graph.txt`, `_filtered_graph_pre.txt`, `_filtered_graph.txt`,
all_hit_segs.txt, linear, cycle, cycle_nodup, `_all_result.txt`) against the checker's chain over ALL records of the sample
(tests/oracle_chain.py: oracle/graph_oracle.cpp -> the golden-pinned Python filter_graph -> uniq -> oracle/match_oracle.cpp ->
remove_cycle_dup), byte for byte; the bench itself asserts that the HBM-resident step's depth sums / copy numbers / edges read as that
`_graph.txt` and its decomposition as that `_all_result.txt`, and that the one-process generateGraph writes the same files.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(contigs, work=None, workload="default"):
    """bench.py as the driver runs it (its own cross-checks decide the exit status); work: the directory its files -> files leg
    writes the sample's files into and leaves them in"""
    env = dict(os.environ)
    if work is not None:
        env.update(PALACE_BENCH_WORK_DIR=str(work), PALACE_BENCH_KEEP="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--contigs", str(contigs), "--workload", workload, "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--soak-seconds", "0.5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=900, env=env)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    line = json.loads(lines[0])
    assert p.returncode == 0, (line.get("failed_checks"), p.stderr[-2000:])
    return line


def check_line(line, contigs):
    assert "failed_checks" not in line
    c = line["config"]
    assert c["workload"].startswith(f"{contigs}-contig synthetic sample") and line["n_gpus"] == 1 and line["steps"] == 2
    d = c["result_digest"]
    assert d["identical_over_untimed_steps"] is True and d["untimed_steps_compared"] >= 10
    e = line["e2e"]
    assert "error" not in e
    assert e["agrees_with_resident_step"] is True and e["all_result_identical_to_resident_step"] is True
    assert e["graph_txt_identical_to_resident_step"] is True
    assert e["one_process_stage04"]["files_identical_to_the_chain"] is True
    assert e["refs_reported"] == c["refs_reported"] and e["junc_lines"] == c["graph"]["n_junc"] > 1000
    r = line["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9
    n_reads = int(c["workload"].split(", ")[1].split(" reads")[0])
    ref_bp = int(c["workload"].split("phage refs (")[1].split(" bp")[0])
    # 864 B per 150-bp read; with Phase B's look-ups fused into the count kernel, their 1 B per ref position and channel too
    # (all three channels on one GPU; none for a rank that exchanges plane slices)
    assert r["algorithmic_bytes_per_launch"] in (864 * n_reads, 864 * n_reads + ref_bp - 31 * 5000, 864 * n_reads + 3 * (ref_bp - 31 * 5000))
    st = line["roofline_stages"]
    assert set(st) == {"phase_b", "classify", "resolve", "stage04"}
    for v in st.values():
        assert v["bound"] == "hbm" and v["ms_per_step"] > 0 and 0 < v["frac"] < 1 and v["algorithmic_bytes_per_step"] > 0
    assert st["classify"]["algorithmic_bytes_per_step"] >= 52 * n_reads
    assert line["value"] == pytest.approx(contigs / (line["ms_per_step"] * 1e-3))
    return c


def files_equal_the_oracle_chain(work, line):
    """every record of the sample through the checker's chain; each file the executables wrote must hold exactly that"""
    import shutil

    from tests import oracle_chain as oc
    from bench import e2e
    P = e2e.e2e_paths(str(work))
    for k in ("fq1", "fq2", "bam"):                                    # (room: the chain below reads the decoded columns, not these)
        os.remove(P[k])
    want_graph, names, lens, avg, _ = oc.oracle_graph(P)
    got_graph = open(P["graph"], "rb").read()
    assert got_graph == want_graph, "generateGraph's _graph.txt differs from the oracle's over all records"
    assert want_graph.count(b"\nJUNC ") == line["config"]["graph"]["n_junc"] > 1000
    o_graph = os.path.join(str(work), "o_graph.txt")
    open(o_graph, "wb").write(want_graph)
    want = oc.oracle_stage04(P, o_graph, os.path.join(str(work), "o"), avg)
    for k in ("pre", "filt", "allhit", "lin", "cyc", "nodup", "result"):
        got = open(P[k], "rb").read()
        if k in ("pre", "filt"):                                       # (SEG block: a set iteration in the reference script)
            assert oc.split_graph(got) == oc.split_graph(want[k]), k
        assert got == want[k], f"{k}: the executables' file differs from the oracle chain's"
        assert open(P[k] + ".fused", "rb").read() == want[k], f"{k}: the one-process generateGraph's file differs from the oracle chain's"
    n_seg, n_junc = want["filt"].count(b"SEG "), want["filt"].count(b"JUNC ")
    assert n_seg == line["config"]["graph"]["n_segs_filtered"] and n_junc == line["config"]["graph"]["n_kept_junc"] > 100
    assert want["result"].count(b"\n") > 1000 and want["result"].count(b"\t") > 1000
    shutil.rmtree(str(work), ignore_errors=True)


def resident_rows(work):
    """the eref rows (n_intervals, el, ref_len, 0 per ref) of the bench's HBM-resident step as it was timed, and how that step counted"""
    rows = np.load(os.path.join(str(work), "eref_rows_resident_step.npy"))
    how = json.load(open(os.path.join(str(work), "eref_rows_resident_step.json")))
    return rows, how


def count_packed_equals_oracle(contigs, timed_rows=None):
    """every read of the workload through palace_eref_pack_reads + palace_eref_count_reads_packed (three planes) against the
    oracle's byte table: plane populations over the whole key space, and look-ups at random keys and at the keys of a present ref.
    timed_rows: the rows of the bench's resident step (the path the headline number times: every look-up of Phase B inside the
    count launch, no plane written) -- held, for ALL refs, against the oracle's scan (extract_ref.cpp:813-903, 504-617) on the
    oracle's table of ALL reads; and the same rows once more through the C ABI in this process (fused count, indexed scan)."""
    import torch

    import bench
    from oracle import binding as orc
    from palace_amd import capi, coder
    dev = torch.device("cuda", 0)
    sample = bench.make_sample(torch, dev, contigs, 5000)
    torch.cuda.synchronize()
    n_side, P, L = sample["n_reads_side"], (lambda t: t.data_ptr()), capi.lib()
    tot = 2 * n_side * bench.READ_LEN
    hdr = coder.header_from_picks(np.random.Generator(np.random.PCG64(bench.SEED)).integers(0, 6, size=32))
    cc = orc.header_to_cc(hdr)
    rng = np.random.Generator(np.random.PCG64(5))
    i = int(sample["present"][0])
    ro = sample["ref_off"][i:i + 2].cpu().numpy()
    ref_keys = orc.index_ref(sample["ref_bases"][int(ro[0]):int(ro[1])].cpu().numpy(), cc).reshape(-1)
    probe = np.unique(np.concatenate([rng.integers(0, 2**32, size=400000, dtype=np.uint64).astype(np.uint32), ref_keys.astype(np.uint32)]))
    with capi.Ctx(0) as ctx:
        ctx.eref_set_coder(hdr)
        nb = int(L.palace_eref_packed_bytes(tot))
        packed = [torch.zeros(nb, dtype=torch.uint8, device=dev) for _ in range(3)]
        torch.cuda.synchronize()
        capi._check(L.palace_eref_pack_reads(ctx.h, P(sample["r12"]), P(sample["read_off"]), 2 * n_side, None, tot, *(P(t) for t in packed)), "pack")
        ctx.eref_table_reset()
        capi._check(L.palace_eref_count_reads_packed(ctx.h, *(P(t) for t in packed), tot, 2 * n_side), "count")
        ctx.sync()
        pops, got = ctx.eref_table_popcounts(), ctx.eref_table_lookup(probe)
        fused_rows = None
        if timed_rows is not None:                                     # the timed path itself, once more through the C ABI (bench/step.py:245-278, 300-358)
            import ctypes
            n_refs = sample["n_refs"]
            ix = ctypes.c_void_p()
            capi._check(L.palace_eref_probe_index_build(ctx.h, P(sample["ref_bases"]), P(sample["ref_off"]), n_refs, sample["ref_total"], ctypes.byref(ix)), "probe index")
            capi._check(L.palace_eref_attach_probe_index(ctx.h, ix), "attach")
            ctx.eref_set_option("final_count", 1)
            ctx.eref_set_option("probe_all_sets", 1)
            rows_d = torch.zeros((n_refs, 4), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            one_min, three_min = capi.window_minimums(0.9, 0.85)
            ctx.eref_table_reset()
            capi._check(L.palace_eref_count_reads_packed(ctx.h, *(P(t) for t in packed), tot, 2 * n_side), "fused count")
            capi._check(L.palace_eref_scan_refs_indexed(ctx.h, ix, P(sample["ref_bases"]), P(sample["ref_off"]), n_refs, sample["ref_total"],
                                                        one_min, three_min, P(rows_d)), "indexed scan")
            ctx.sync()
            fused_rows = rows_d.cpu().numpy()
            ctx.eref_set_option("probe_all_sets", 0)
            ctx.eref_set_option("final_count", 0)
            capi._check(L.palace_eref_attach_probe_index(ctx.h, None), "detach")
            ctx.eref_probe_index_free(ix)
            ctx.eref_table_reset()
    b12 = sample["r12"].cpu().numpy()
    off = sample["read_off"].cpu().numpy()
    ref_b = sample["ref_bases"].cpu().numpy() if timed_rows is not None else None
    ref_o = sample["ref_off"].cpu().numpy() if timed_rows is not None else None
    del sample, packed
    torch.cuda.empty_cache()
    table = orc.CountTable()
    table.clear()
    table.count_mt(b12, off, cc, max(1, min(os.cpu_count() or 1, 16)))
    want = table.lookup(probe)
    want_pops = [0, 0, 0]                                               # entries with count >= 1, >= 2, >= 3 over the whole key space
    for lo in range(0, 1 << 32, 1 << 28):
        v = table.view[lo:lo + (1 << 28)]
        for k in range(3):
            want_pops[k] += int(np.count_nonzero(v > k))
    if timed_rows is not None:
        # the oracle's Phase B over every ref of the DB on that table: index (extract_ref.cpp:711-738), look-ups (:858-870), windows (:504-617)
        from concurrent.futures import ThreadPoolExecutor

        def scan(i):
            s_ = ref_b[int(ref_o[i]):int(ref_o[i + 1])]
            _, n_int, el, _ = orc.scan_ref(orc.index_ref(s_, cc), len(s_), table, 0.9, 0.85)
            return n_int, el, len(s_)
        with ThreadPoolExecutor(max(1, min(os.cpu_count() or 1, 16))) as ex:           # (ctypes calls release the interpreter lock)
            want_rows = np.array(list(ex.map(scan, range(len(ref_o) - 1))), dtype=np.int64)
    table.free()
    if timed_rows is not None:
        assert timed_rows.shape == (len(want_rows), 4)
        assert np.array_equal(timed_rows[:, :3].astype(np.int64), want_rows), "the bench's timed eref rows differ from the oracle's scan of every ref"
        assert np.array_equal(fused_rows[:, :3].astype(np.int64), want_rows), "fused count + indexed scan through the C ABI differ from the oracle"
        printed = (want_rows[:, 1] > 0) & (want_rows[:, 1].astype(np.float32) / want_rows[:, 2].astype(np.float32) > np.float32(0.75))
        assert printed.sum() >= 150 and (want_rows[:, 0] > 0).sum() >= printed.sum()
    assert np.array_equal(got, want)
    assert (want == 3).sum() > 10000 and (want == 1).sum() > 10000 and (want == 0).sum() > 10000
    assert list(pops) == want_pops
    assert pops[0] > pops[1] > pops[2] > 0


def test_config2_500k_contigs_bench_checks_oracle_chain_and_oracle_table(tmp_path):
    line = run_bench(500_000, tmp_path / "w")
    c = check_line(line, 500_000)
    assert 150 <= c["refs_reported"] <= c["refs_present"] == 200        # (half the read depth of the 1M workload: a few refs fall short)
    rows, how = resident_rows(tmp_path / "w")
    assert how["probe_all_sets"] is True and how["reads"] == "packed"   # the default step: Phase B's look-ups inside the count launch
    files_equal_the_oracle_chain(tmp_path / "w", line)
    count_packed_equals_oracle(500_000, rows)


def test_headline_1m_contigs_bench_checks_oracle_chain_and_oracle_table(tmp_path):
    line = run_bench(1_000_000, tmp_path / "w")
    c = check_line(line, 1_000_000)
    assert c["refs_reported"] == c["refs_present"] == 200
    assert line["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    rows, how = resident_rows(tmp_path / "w")
    assert how["probe_all_sets"] is True and how["reads"] == "packed"   # the configuration the headline number is timed in
    files_equal_the_oracle_chain(tmp_path / "w", line)
    count_packed_equals_oracle(1_000_000, rows)


def test_config4_long_contigs_bench_checks_and_oracle_chain(tmp_path):
    """100k contigs with N50 ~ 50 kb and 6.67 M records: half of the cross-contig pairs reach the exp-underflow gate (G5)"""
    line = run_bench(100_000, tmp_path / "w", workload="long")
    c = check_line(line, 100_000)
    assert c["workload_kind"] == "long" and c["refs_reported"] == c["refs_present"] == 200
    files_equal_the_oracle_chain(tmp_path / "w", line)
