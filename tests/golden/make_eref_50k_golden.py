#!/usr/bin/env python3
"""Generate tests/golden/eref_50k.npz by RUNNING the compiled reference (oracle/_ref/eref_ref, the unmodified
bin/extract_ref.cpp built by oracle/Makefile) on the BASELINE config[1] eref input: 5 000 phage refs (200 Mb) and the
read pairs of a 50k-contig sample (palace_amd.synth.eref_config_inputs, seed 20261003).

Build container only (needs /root/reference, ~21 GB of RAM, ~1.5 min).  Stored: the 400-byte coder header the
reference drew for its index (time-seeded, extract_ref.cpp:1088), the reference's stdout at threads=1 for two
ratio pairs, and the sha256 of the three input files -- data only, so that the GPU-side test can regenerate the
inputs from the seed, prove they are the same bytes, and compare results byte for byte."""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from palace_amd import synth  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "eref_ref")
WORK = "/tmp/palace_golden_eref50k"
SEED, N_REFS, N_PAIRS = 20261003, 5000, 166_666


def main():
    os.makedirs(WORK, exist_ok=True)
    fa, fq1, fq2 = synth.eref_config_inputs(SEED, N_REFS, N_PAIRS)
    paths = [os.path.join(WORK, n) for n in ("db.fa", "r_1.fq", "r_2.fq")]
    for p, b in zip(paths, (fa, fq1, fq2)):
        open(p, "wb").write(b)
    for ext in (".k32.index.dat", ".genome.len.txt"):
        if os.path.exists(paths[0] + ext):
            os.remove(paths[0] + ext)
    env = dict(os.environ, MALLOC_PERTURB_="255")
    outs = {}
    for tag, hr, pr in (("build_090_085", "0.9", "0.85"), ("stdout_090_085", "0.9", "0.85"), ("stdout_080_050", "0.8", "0.5")):
        t0 = time.time()
        outs[tag] = subprocess.run([REF_BIN, paths[1], paths[2], paths[0], os.path.join(WORK, "tmp.txt"), hr, pr, "1"],
                                   stdout=subprocess.PIPE, check=True, env=env).stdout
        print(f"{tag}: {time.time() - t0:.1f} s, {outs[tag].count(10)} lines", flush=True)
    assert outs["build_090_085"] == outs["stdout_090_085"], "index-building run and cached-index run disagree"
    header = open(paths[0] + ".k32.index.dat", "rb").read(400)
    sha = {n: hashlib.sha256(b).hexdigest() for n, b in zip(("db_fa", "fq1", "fq2"), (fa, fq1, fq2))}
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "eref_50k.npz"),
                        index_header=np.frombuffer(header, dtype=np.uint8),
                        stdout_090_085=np.frombuffer(outs["stdout_090_085"], dtype=np.uint8),
                        stdout_080_050=np.frombuffer(outs["stdout_080_050"], dtype=np.uint8),
                        genome_len_sha256=np.array(hashlib.sha256(open(paths[0] + ".genome.len.txt", "rb").read()).hexdigest()),
                        index_sha256=np.array(hashlib.sha256(open(paths[0] + ".k32.index.dat", "rb").read()).hexdigest()),
                        sha256_db_fa=np.array(sha["db_fa"]), sha256_fq1=np.array(sha["fq1"]), sha256_fq2=np.array(sha["fq2"]),
                        params=np.array([SEED, N_REFS, N_PAIRS], dtype=np.int64))
    print("wrote tests/golden/eref_50k.npz", sha)


if __name__ == "__main__":
    main()
