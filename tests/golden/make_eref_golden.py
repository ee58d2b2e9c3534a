#!/usr/bin/env python3
"""Generate tests/golden/eref_toy.npz by RUNNING the compiled reference (oracle/_ref/eref_ref).

Run in the build container only (needs /root/reference compiled via `make -C oracle ref`,
~21 GB RAM and ~2.5 min per reference invocation: SURVEY.md F6).  The GPU box never runs this;
it only reads the committed .npz.  What is stored is data: the seeded inputs and the outputs the
reference produced for them -- no reference source text.

Pinning choices:
  * threads=1, the only configuration with defined semantics (F5);
  * MALLOC_PERTURB_=255 so fresh heap memory is zero-filled, which defines the never-written
    tail of record_ref_hit (extract_ref.cpp:856) as 0;
  * the index (and with it the time(0)-seeded coder permutation, extract_ref.cpp:1088) is built
    by the first run; its 400-byte header is kept so both sides use the same permutation.
"""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from palace_amd import synth  # noqa: E402

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "eref_ref")
WORK = "/tmp/palace_golden_eref"
SEED = 20261003


def build_toy():
    rng = synth.rng_for(SEED)
    refs = []          # (header, seq, [(lo, hi, depth, divergence)])
    def R(header, L, segs):
        refs.append([header, synth.random_dna(rng, L), segs])
    R("phiA_full complete genome", 9000, [(0, 9000, 14, 0.0)])
    R("phiB_gap1200|x", 12000, [(0, 4000, 14, 0.0), (5200, 12000, 14, 0.0)])
    R("phiC_two_intervals", 15000, [(0, 3000, 14, 0.0), (8000, 15000, 14, 0.0)])
    R("phiD_below_ratio", 10000, [(2000, 7000, 14, 0.0)])
    R("phiE_absent", 8000, [])
    R("phiF_tiny", 40, [])                      # len<=32+..: 40>32 so it IS indexed (see below)
    R("phiG_len32", 32, [])                     # len<=32: not written to the index (extract_ref.cpp:697)
    R("phiH_short300", 300, [(0, 300, 14, 0.0)])
    R("phiI_withN/1 extra", 7000, [(0, 7000, 14, 0.0)])
    R("phiJ_lowdepth", 6000, [(0, 6000, 2, 0.0)])
    R("phiK_diverged\ttabbed", 20000, [(0, 20000, 16, 0.02)])
    R("phiL_gap900", 11000, [(0, 5000, 14, 0.0), (5900, 11000, 14, 0.0)])
    R("phiM_last", 5000, [(0, 5000, 14, 0.0)])
    # N run + lower-case stretch in phiI
    s = refs[8][1]
    s[3000:3050] = ord("N")
    s[5000:5400] = np.frombuffer(s[5000:5400].tobytes().lower(), dtype=np.uint8)
    s[6000] = ord("R")
    names = [r[0] for r in refs]
    seqs = [r[1] for r in refs]
    db = synth.PhageDB(names, seqs)

    r1, r2 = [], []
    for header, seq, segs in refs:
        for lo, hi, depth, div in segs:
            tmpl = synth.mutate(rng, seq, div) if div > 0 else seq
            n_pairs = max(1, int(depth * (hi - lo) / 200))
            rl = 100
            if hi - lo < 400:
                a, b = synth.sample_pairs(rng, tmpl, n_pairs, rl, hi - lo, 1, 0.004, lo, hi)
            else:
                a, b = synth.sample_pairs(rng, tmpl, n_pairs, rl, 320, 30, 0.004, lo, hi)
            r1 += a
            r2 += b
    # background reads + edge cases (ragged lengths, N, lower case, <32, ==32, long)
    bg = synth.random_dna(rng, 400000)
    for L in [100] * 1500 + [150] * 300 + [31, 32, 33, 1, 250, 399]:
        st = int(rng.integers(0, len(bg) - 400))
        a = bg[st:st + L].copy()
        b = synth.revcomp(bg[st + 50:st + 50 + L])
        if rng.random() < 0.05 and L > 40:
            a[int(rng.integers(0, L))] = ord("N")
        if rng.random() < 0.05:
            b = np.frombuffer(b.tobytes().lower(), dtype=np.uint8)
        r1.append(a)
        r2.append(b)
    order = rng.permutation(len(r1))
    r1 = [r1[i] for i in order]
    r2 = [r2[i] for i in order]
    return db, synth.reads_from_list(r1), synth.reads_from_list(r2)


def run_ref(args, env):
    t0 = time.time()
    p = subprocess.run([REF_BIN] + args, env=env, stdout=subprocess.PIPE, check=True)
    print(f"  reference run {args[4:]} took {time.time() - t0:.0f} s, {len(p.stdout)} B stdout", flush=True)
    return p.stdout


def main():
    os.makedirs(WORK, exist_ok=True)
    for f in os.listdir(WORK):
        os.remove(os.path.join(WORK, f))
    db, r1, r2 = build_toy()
    fa, q1, q2 = (os.path.join(WORK, n) for n in ("db.fa", "r_1.fq", "r_2.fq"))
    db.write_fasta(fa)
    r1.write_fastq(q1, "1")
    r2.write_fastq(q2, "2")
    env = dict(os.environ, MALLOC_PERTURB_="255")
    tmp = os.path.join(WORK, "tmp.txt")
    out_a = run_ref([q1, q2, fa, tmp, "0.9", "0.85", "1"], env)      # builds the index
    out_b = run_ref([q1, q2, fa, tmp, "0.8", "0.5", "1"], env)       # cached index
    out_c = run_ref([q1, q2, fa, tmp, "0.95", "0.9", "1"], env)
    idx = open(fa + ".k32.index.dat", "rb").read()
    glen = open(fa + ".genome.len.txt", "rb").read()
    assert os.path.getsize(tmp) == 0
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "eref_toy.npz"),
        db_fasta=np.frombuffer(open(fa, "rb").read(), dtype=np.uint8),
        r1_bases=r1.bases, r1_offsets=r1.offsets, r2_bases=r2.bases, r2_offsets=r2.offsets,
        index_header=np.frombuffer(idx[:400], dtype=np.uint8),
        index_head_slice=np.frombuffer(idx[400:400 + 4 + 12 * 2000], dtype=np.uint8),
        index_sha256=np.frombuffer(hashlib.sha256(idx).digest(), dtype=np.uint8),
        index_body_sha256=np.frombuffer(hashlib.sha256(idx[400:]).digest(), dtype=np.uint8),
        index_size=np.int64(len(idx)),
        genome_len_txt=np.frombuffer(glen, dtype=np.uint8),
        stdout_090_085=np.frombuffer(out_a, dtype=np.uint8),
        stdout_080_050=np.frombuffer(out_b, dtype=np.uint8),
        stdout_095_090=np.frombuffer(out_c, dtype=np.uint8),
    )
    print(out_a.decode(), out_b.decode(), out_c.decode(), sep="\n---\n")


if __name__ == "__main__":
    main()
