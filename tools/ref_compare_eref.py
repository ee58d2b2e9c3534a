#!/usr/bin/env python3
"""One-off, GPU box: run the COMPILED REFERENCE eref (oracle/_ref/eref_ref, built from the unmodified
extract_ref.cpp in the build container) and this repository's eref on the same mid-size synthetic input and
compare stdout byte for byte; print wall times.  Not part of bench.py or the tests (the reference run needs
~21 GB of RAM and minutes).  usage: ref_compare_eref.py <workdir> [n_refs] [n_pairs] [fast]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from palace_amd import synth  # noqa: E402


def main():
    work = sys.argv[1]
    n_refs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    n_pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 166666
    os.makedirs(work, exist_ok=True)
    print(f"generating {n_refs} refs and {n_pairs} read pairs ...", flush=True)
    rng = synth.rng_for(20261003)
    lens = rng.integers(20000, 60001, size=n_refs)
    refs = [synth.random_dna(rng, int(L)) for L in lens]
    fa = os.path.join(work, "db.fa")
    with open(fa, "wb") as f:
        for i, s in enumerate(refs):
            f.write(b">phage_%d synthetic\n" % (i + 1))
            b = s.tobytes()
            f.write(b"\n".join(b[k:k + 80] for k in range(0, len(b), 80)) + b"\n")
    present = rng.choice(n_refs, size=max(1, n_refs // 100), replace=False)
    pool = synth.random_dna(rng, 60_000_000)
    n_ph = n_pairs // 5
    comp = np.zeros(256, dtype=np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    ar = np.arange(150)

    def cut(src, st):
        return src[st[:, None] + ar[None, :]]

    ins = np.clip(rng.normal(400, 40, size=n_pairs), 150, 800).astype(np.int64)
    st_pool = rng.integers(0, len(pool) - 1000, size=n_pairs - n_ph)
    r1 = [cut(pool, st_pool)]
    r2 = [cut(pool, st_pool + ins[: n_pairs - n_ph] - 150)]
    which = rng.integers(0, len(present), size=n_ph)
    per = [np.nonzero(which == k)[0] for k in range(len(present))]
    a1 = np.zeros((n_ph, 150), dtype=np.uint8)
    a2 = np.zeros((n_ph, 150), dtype=np.uint8)
    for k, idx in enumerate(per):
        s = refs[present[k]]
        st = rng.integers(0, len(s) - 900, size=len(idx))
        a1[idx] = cut(s, st)
        a2[idx] = cut(s, st + ins[n_pairs - n_ph:][idx] - 150)
    for a in (a1, a2):
        m = rng.random(a.shape) < 0.005
        a[m] = synth.ACGT[rng.integers(0, 4, size=int(m.sum()))]
    r1.append(a1)
    r2.append(a2)
    r1 = np.concatenate(r1)
    r2 = comp[np.concatenate(r2)[:, ::-1]]
    perm = rng.permutation(n_pairs)
    for tag, arr in (("1", r1[perm]), ("2", r2[perm])):
        with open(os.path.join(work, f"r_{tag}.fq"), "wb") as f:
            q = b"I" * 150
            f.write(b"".join(b"@r%d/%s\n" % (i, tag.encode()) + arr[i].tobytes() + b"\n+\n" + q + b"\n" for i in range(n_pairs)))
        print(f"  wrote r_{tag}.fq", flush=True)
    args = [os.path.join(work, "r_1.fq"), os.path.join(work, "r_2.fq"), fa, os.path.join(work, "tmp.txt"), "0.9", "0.85"]
    out = {}
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "eref_ref")
    if len(sys.argv) > 4 and sys.argv[4] == "fast":
        # one reference run only (it builds the index with its time-seeded coder and scans with it; our eref then reads
        # that index), with a heartbeat so that a long silent run is not taken for a hang
        print("inputs written; running the reference (threads=1, builds the index) ...", flush=True)
        t0 = time.time()
        with open(os.path.join(work, "ref.out"), "wb") as fo:
            pr = subprocess.Popen([ref_bin] + args + ["1"], stdout=fo, env=dict(os.environ, MALLOC_PERTURB_="255"))
            while pr.poll() is None:
                time.sleep(5)
                if int(time.time() - t0) % 60 < 5:
                    print(f"  reference running, {time.time() - t0:.0f} s", flush=True)
        t_ref = time.time() - t0
        ref_out = open(os.path.join(work, "ref.out"), "rb").read()
        t0 = time.time()
        ours = subprocess.run([os.path.join(ROOT, "palace_amd", "bin", "eref")] + args + ["16"], stdout=subprocess.PIPE, check=True).stdout
        t_ours = time.time() - t0
        print(f"refs={n_refs} ({int(lens.sum())} bp) read pairs={n_pairs} x150; lines reported: ours {ours.count(10)}, reference {ref_out.count(10)}")
        print(f"reference eref (index build + run, threads=1): {t_ref:.1f} s (rc {pr.returncode}); this repository's eref (CLI wall): {t_ours:.2f} s")
        print(f"stdout byte-identical to the reference: {ours == ref_out}")
        sys.exit(0 if ours == ref_out else 1)
    t0 = time.time()
    out["ref_build"] = subprocess.run([ref_bin] + args + ["1"], stdout=subprocess.PIPE, check=True,
                                      env=dict(os.environ, MALLOC_PERTURB_="255")).stdout
    t_ref_build = time.time() - t0
    t0 = time.time()
    out["ref_cached"] = subprocess.run([ref_bin] + args + ["1"], stdout=subprocess.PIPE, check=True,
                                       env=dict(os.environ, MALLOC_PERTURB_="255")).stdout
    t_ref = time.time() - t0
    t0 = time.time()
    out["ref_t16"] = subprocess.run([ref_bin] + args + ["16"], stdout=subprocess.PIPE, check=True,
                                    env=dict(os.environ, MALLOC_PERTURB_="255")).stdout
    t_ref16 = time.time() - t0
    t0 = time.time()
    ours = subprocess.run([os.path.join(ROOT, "palace_amd", "bin", "eref")] + args + ["16"], stdout=subprocess.PIPE, check=True).stdout
    t_ours = time.time() - t0
    same = ours == out["ref_cached"]
    same16 = sorted(out["ref_t16"].splitlines()) == sorted(ours.splitlines())
    print(f"refs={n_refs} ({int(lens.sum())} bp) read pairs={n_pairs} x150; lines reported: ours {ours.count(10)}, reference {out['ref_cached'].count(10)}")
    print(f"reference eref (index build + run, threads=1): {t_ref_build:.1f} s; cached index threads=1: {t_ref:.1f} s; threads=16: {t_ref16:.1f} s")
    print(f"this repository's eref (CLI wall, incl. text parsing + H2D): {t_ours:.2f} s")
    print(f"stdout byte-identical to the reference (threads=1, cached index): {same}; same line set as reference threads=16: {same16}")
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
