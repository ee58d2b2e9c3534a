#!/usr/bin/env python3
"""Generate tests/golden/eref_e3.npz by RUNNING the compiled reference (oracle/_ref/eref_ref) on an input that is large
enough for its read subsampling (row E3) to switch on by itself: sum of fq1 sequence bases = 1.26e9 > 1e9, so
cal_sam_ratio (extract_ref.cpp:1124-1148) gives 79 and every sequence line of fq1, then fq2, draws rand() % 100 from the
seed-1 glibc stream (extract_ref.cpp:955-960, 1239-1240; the second run finds the index, so random_coder does not reseed).
100 of 200 refs are present at 3-10x, so that dropping a fifth of the reads changes which k-mers reach count 3.

Build container only (~21 GB of RAM, ~1.5 h at threads=1).  Stored: coder header, stdout of the cached-index runs, input
sha256 -- data only."""
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
WORK = "/tmp/palace_golden_e3"
PARAMS = dict(seed=20261004, n_refs=200, n_pairs=8_400_000, n_present=100, n_phage_pairs=67_000)


def main():
    os.makedirs(WORK, exist_ok=True)
    fa, fq1, fq2 = synth.eref_config_inputs(**PARAMS)
    paths = [os.path.join(WORK, n) for n in ("db.fa", "r_1.fq", "r_2.fq")]
    sha = {}
    for p, b, k in zip(paths, (fa, fq1, fq2), ("db_fa", "fq1", "fq2")):
        open(p, "wb").write(b)
        sha[k] = hashlib.sha256(b).hexdigest()
    del fa, fq1, fq2
    for ext in (".k32.index.dat", ".genome.len.txt"):
        if os.path.exists(paths[0] + ext):
            os.remove(paths[0] + ext)
    env = dict(os.environ, MALLOC_PERTURB_="255")
    # a first, tiny run only builds the index (so that the timed-seed coder exists and later runs keep srand(1))
    tiny = os.path.join(WORK, "tiny.fq")
    open(tiny, "wb").write(b"@t\nACGT\n+\nIIII\n")
    subprocess.run([REF_BIN, tiny, tiny, paths[0], os.path.join(WORK, "tmp.txt"), "0.9", "0.85", "1"], stdout=subprocess.DEVNULL, check=True, env=env)
    outs = {}
    for tag, hr, pr in (("stdout_090_085", "0.9", "0.85"), ("stdout_080_050", "0.8", "0.5")):
        t0 = time.time()
        outs[tag] = subprocess.run([REF_BIN, paths[1], paths[2], paths[0], os.path.join(WORK, "tmp.txt"), hr, pr, "1"],
                                   stdout=subprocess.PIPE, check=True, env=env).stdout
        print(f"{tag}: {time.time() - t0:.1f} s, {outs[tag].count(10)} lines", flush=True)
    header = open(paths[0] + ".k32.index.dat", "rb").read(400)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "eref_e3.npz"),
                        index_header=np.frombuffer(header, dtype=np.uint8),
                        stdout_090_085=np.frombuffer(outs["stdout_090_085"], dtype=np.uint8),
                        stdout_080_050=np.frombuffer(outs["stdout_080_050"], dtype=np.uint8),
                        sha256_db_fa=np.array(sha["db_fa"]), sha256_fq1=np.array(sha["fq1"]), sha256_fq2=np.array(sha["fq2"]),
                        params=np.array([PARAMS[k] for k in ("seed", "n_refs", "n_pairs", "n_present", "n_phage_pairs")], dtype=np.int64))
    print("wrote tests/golden/eref_e3.npz", sha)


if __name__ == "__main__":
    main()
