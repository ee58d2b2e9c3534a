#!/usr/bin/env python3
"""Generate tests/golden/eref_1m.npz by RUNNING the compiled reference (oracle/_ref/eref_ref, the unmodified
bin/extract_ref.cpp built by oracle/Makefile) on an eref input of the HEADLINE configuration's size: 5 000 phage refs
(200 Mb) and the 3 333 333 read pairs x 150 bp of a 1M-contig sample (5e8 fq1 bases), 200 refs
present at ~12x, the other reads from a 1 Gb contig pool (palace_amd.synth.eref_config_inputs, seed 20261003).

Build container only (needs /root/reference, ~25 GB of RAM, ~30 min: three reference runs at threads=1).  Stored: the
400-byte coder header the reference drew for its index (time-seeded, extract_ref.cpp:1088), the reference's stdout for two
ratio pairs, and the sha256 of the three input files -- data only: the GPU-side test regenerates the inputs from the
seed, proves they are the same bytes, and compares the executable AND the resident (fused-probe) step byte for byte."""
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
WORK = "/tmp/palace_golden_eref1m"
SEED, N_REFS, N_PAIRS, N_PRESENT, POOL = 20261003, 5000, 3_333_333, 200, 1_000_000_000


def inputs():
    return synth.eref_config_inputs(SEED, N_REFS, N_PAIRS, pool_bases=POOL, n_present=N_PRESENT, n_phage_pairs=N_PAIRS // 10)


def main():
    os.makedirs(WORK, exist_ok=True)
    t0 = time.time()
    fa, fq1, fq2 = inputs()
    print(f"inputs generated in {time.time() - t0:.1f} s", flush=True)
    paths = [os.path.join(WORK, n) for n in ("db.fa", "r_1.fq", "r_2.fq")]
    for p, b in zip(paths, (fa, fq1, fq2)):
        open(p, "wb").write(b)
    sha = {n: hashlib.sha256(b).hexdigest() for n, b in zip(("db_fa", "fq1", "fq2"), (fa, fq1, fq2))}
    del fa, fq1, fq2
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
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "eref_1m.npz"),
                        index_header=np.frombuffer(header, dtype=np.uint8),
                        stdout_090_085=np.frombuffer(outs["stdout_090_085"], dtype=np.uint8),
                        stdout_080_050=np.frombuffer(outs["stdout_080_050"], dtype=np.uint8),
                        sha256_db_fa=np.array(sha["db_fa"]), sha256_fq1=np.array(sha["fq1"]), sha256_fq2=np.array(sha["fq2"]),
                        params=np.array([SEED, N_REFS, N_PAIRS, N_PRESENT, POOL], dtype=np.int64))
    print("wrote tests/golden/eref_1m.npz", sha)


if __name__ == "__main__":
    main()
