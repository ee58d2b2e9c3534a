"""The bench workloads at their own size (driver-visible: `pytest -m gpu`).

  configs[2]  500k contigs: `bench.py --contigs 500000` must exit 0 -- it exits non-zero when its own cross-checks fail (the
              executables on the generated files disagreeing with the HBM-resident step, the fused generateGraph process
              writing other files than the five-process chain, results that differ from step to step) -- and every read of
              the workload counted through the packed entry (the bench's data path) equals the oracle's table.
  headline    the 1M-contig workload BASELINE.json's metric is quoted on: the same, plus every present ref reported.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(contigs):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--contigs", str(contigs), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--soak-seconds", "0.5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=900)
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
    assert e["one_process_stage04"]["files_identical_to_the_chain"] is True
    assert e["refs_reported"] == c["refs_reported"] and e["junc_lines"] == c["graph"]["n_junc"] > 1000
    r = line["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9
    n_reads = int(c["workload"].split(", ")[1].split(" reads")[0])
    ref_bp = int(c["workload"].split("phage refs (")[1].split(" bp")[0])
    # 864 B per 150-bp read; with Phase B's channel-0 probe fused into the count kernel, its 1 B per ref position too
    assert r["algorithmic_bytes_per_launch"] in (864 * n_reads, 864 * n_reads + ref_bp - 31 * 5000)
    st = line["roofline_stages"]
    assert set(st) == {"phase_b", "classify", "resolve", "stage04"}
    for v in st.values():
        assert v["bound"] == "hbm" and v["ms_per_step"] > 0 and 0 < v["frac"] < 1 and v["algorithmic_bytes_per_step"] > 0
    assert st["classify"]["algorithmic_bytes_per_step"] >= 52 * n_reads
    assert line["value"] == pytest.approx(contigs / (line["ms_per_step"] * 1e-3))
    return c


def count_packed_equals_oracle(contigs):
    """every read of the workload through palace_eref_pack_reads + palace_eref_count_reads_packed (three planes) against the
    oracle's byte table: plane populations over the whole key space, and look-ups at random keys and at the keys of a present ref"""
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
    b12 = sample["r12"].cpu().numpy()
    off = sample["read_off"].cpu().numpy()
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
    table.free()
    assert np.array_equal(got, want)
    assert (want == 3).sum() > 10000 and (want == 1).sum() > 10000 and (want == 0).sum() > 10000
    assert list(pops) == want_pops
    assert pops[0] > pops[1] > pops[2] > 0


def test_config2_500k_contigs_bench_checks_and_oracle_table():
    line = run_bench(500_000)
    c = check_line(line, 500_000)
    assert 150 <= c["refs_reported"] <= c["refs_present"] == 200        # (half the read depth of the 1M workload: a few refs fall short)
    count_packed_equals_oracle(500_000)


def test_headline_1m_contigs_bench_checks_and_oracle_table():
    line = run_bench(1_000_000)
    c = check_line(line, 1_000_000)
    assert c["refs_reported"] == c["refs_present"] == 200
    assert line["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    count_packed_equals_oracle(1_000_000)
