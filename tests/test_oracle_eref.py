"""Pin oracle/eref_oracle.c against outputs of the compiled reference (tests/golden/eref_toy.npz,
made by tests/golden/make_eref_golden.py from oracle/_ref/eref_ref)."""
import hashlib
import os

import numpy as np
import pytest

from oracle import binding as orc


@pytest.fixture(scope="module")
def toy(golden_eref, tmp_path_factory):
    d = tmp_path_factory.mktemp("eref_toy")
    fa = str(d / "db.fa")
    open(fa, "wb").write(golden_eref["db_fasta"].tobytes())
    hdr = golden_eref["index_header"]
    orc.build_index_file(fa, hdr, fa + ".k32.index.dat", fa + ".genome.len.txt")
    cc = orc.header_to_cc(hdr)
    table = orc.CountTable()
    table.count(golden_eref["r1_bases"], golden_eref["r1_offsets"], cc)
    table.count(golden_eref["r2_bases"], golden_eref["r2_offsets"], cc)
    yield dict(fa=fa, cc=cc, table=table, g=golden_eref)
    table.free()


def test_index_file_matches_reference(toy):
    g = toy["g"]
    idx = open(toy["fa"] + ".k32.index.dat", "rb").read()
    assert len(idx) == int(g["index_size"])
    assert idx[:400] == g["index_header"].tobytes()
    assert hashlib.sha256(idx[400:]).digest() == g["index_body_sha256"].tobytes()
    head = g["index_head_slice"].tobytes()
    assert idx[400:400 + len(head)] == head


def test_genome_len_file_matches_reference(toy):
    assert open(toy["fa"] + ".genome.len.txt", "rb").read() == toy["g"]["genome_len_txt"].tobytes()


@pytest.mark.parametrize("key,hr,pr", [("stdout_090_085", 0.9, 0.85), ("stdout_080_050", 0.8, 0.5),
                                       ("stdout_095_090", 0.95, 0.9)])
def test_stdout_matches_reference(toy, key, hr, pr):
    out = orc.scan_index_file(toy["fa"] + ".k32.index.dat", toy["table"], hr, pr)
    assert out == toy["g"][key].tobytes()


def test_glibc_rand_restatement_matches_libc():
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 12345):
        libc.srand(seed)
        want = [libc.rand() for _ in range(1000)]
        assert orc.glibc_rand_stream(seed, 1000).tolist() == want


def test_sample_ratio():
    assert orc.sample_ratio(5 * 10**8) == 200          # <= 1e9 fq1 bases: every read is kept
    assert orc.sample_ratio(2 * 10**9) == 50
