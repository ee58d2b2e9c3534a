#!/usr/bin/env python3
"""Headline bench: contigs/s over eref + generateGraph + matching on the 1M-contig synthetic
(BASELINE.json `metric`), inputs resident in HBM, one process per GPU.

  python bench.py [--gpus N --steps K --warmup W] [--contigs 1000000] [--workload default|long]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(`python bench.py --gpus N` without a launcher starts its N ranks itself, as child processes, before this process
touches a GPU, relays rank 0's JSON line and exits with their status.)

One "step" = one full pass of the hot path over the whole synthetic sample:
  eref:   zero the count table, count every read of both FASTQ sides, scan every phage ref
  (generateGraph and matching stages are added to the step as they land; `config.stages` names
   what the printed number covers.)
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant
kernel (live HIP-event timing on the kernel's own stream) and `cpu_baseline` (the oracle, i.e.
the CPU restatement of the reference algorithm, timed on a bounded sample on this host).
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench.sample import HBM_PEAK_GBS, READ_LEN, SEED  # noqa: E402,F401  (bench/: the parts of this bench)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("default", "long"), default="default",
                    help="default: BASELINE configs[1..3] shape (log-normal contigs, median 800 bp); long: configs[4] "
                         "(100k contigs, N50 ~ 50 kb, tail > 120 kb, evidence that reaches the exp-underflow gate)")
    ap.add_argument("--contigs", type=int, default=None)
    ap.add_argument("--refs", type=int, default=5000)
    ap.add_argument("--reads", choices=("packed", "ascii"), default="packed",
                    help="form of the reads resident in HBM: packed = two bits per base + 32-mer start mask (what the eref executable's "
                         "parser threads produce; palace_eref_count_reads_packed), ascii = a byte per base (palace_eref_count_reads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files -> files leg (CLI chain on generated files)")
    ap.add_argument("--soak-seconds", type=float, default=2.0,
                    help="after the K timed steps keep stepping (untimed for `value`) until this much wall time has passed")
    ap.add_argument("--cpu-sample-frac", type=float, default=0.05, help="share of the reads the CPU baseline's eref term is timed on (the compiled reference: ~30 s)")
    a = ap.parse_args()
    if a.contigs is None:
        a.contigs = 100_000 if a.workload == "long" else 1_000_000
    return a


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (torch.distributed.run, one rank
    per GPU, rendezvous on 127.0.0.1) before this process has touched a GPU, let them print the JSON line, return their
    exit status.  (Never exec from a process that initialised the GPU.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def scheme_plan(world):
    """The N-rank configurations one `--gpus N` run measures (palace_amd/multigpu.py): the weak leg (every rank the whole one-GPU step
    on a full sample: its digests are what every scheme's result must equal), then every Phase-A scheme that exists for this many
    ranks, the read-sharded ones also with rank 0 taking no reads (stage 04 then has rank 0's device to itself)."""
    plan = [("weak", {"PALACE_BENCH_LEG": "weak"})]
    strong = lambda scheme, r0: (scheme + ("" if r0 else ", rank 0 idle in Phase A"), {"PALACE_BENCH_LEG": "strong", "PALACE_BENCH_SCHEME": scheme,
                                                                                      "PALACE_BENCH_RANK0_READS": "1" if r0 else "0"})
    plan.append(strong("replicate", True))
    if 64 % world == 0:
        plan.append(strong("key_split", True))
    for scheme in ("shard_reads", "shard_counts"):
        plan.append(strong(scheme, True))
        if world > 2:
            plan.append(strong(scheme, False))
    only = os.environ.get("PALACE_BENCH_ONLY_SCHEMES")                 # rehearsals: a comma-separated subset of the labels' schemes
    if only:
        keep = set(only.split(","))
        plan = [pl for pl in plan if pl[0] == "weak" or pl[1]["PALACE_BENCH_SCHEME"] in keep]
    return plan


def run_all_schemes(args) -> int:
    """One rank of a `--gpus N` run (N > 1) that was not told which scheme to use: this process never touches a GPU.  It starts one
    CHILD process per configuration of scheme_plan() -- the same bench.py, the same rank, a rendezvous port of its own, the scheme
    forced -- and waits for it with a time limit: a configuration that hangs in a collective, faults or exits non-zero is killed and
    reported under its name, and the others are still measured.  Every child times exactly --steps steps.  Rank 0 prints ONE line:
    the line of the fastest configuration whose results equal the one-GPU step's (`value` is that child's), with every configuration's
    time and verdict under `parallelism_measured`; a configuration that failed is a failed check (exit status 3), never a silent skip."""
    import signal
    import subprocess
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"])
    base = int(os.environ.get("MASTER_PORT", "29500"))
    limit = float(os.environ.get("PALACE_BENCH_SCHEME_TIMEOUT", "240"))
    plan, res = scheme_plan(world), {}
    for k, (label, env_add) in enumerate(plan):
        port = base + 64 * (k + 1) if base + 64 * (len(plan) + 1) < 65536 else base - 64 * (k + 1)      # (a rendezvous port of its own, away from the launcher's)
        env = dict(os.environ, PALACE_BENCH_CHILD="1", MASTER_PORT=str(port), **env_add)
        for v in ("TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RUN_ID"):      # the child's rank 0 hosts the store of ITS group on ITS port
            env.pop(v, None)
        t0 = time.perf_counter()
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL,
                             start_new_session=True)
        try:
            out, _ = p.communicate(timeout=limit)
            status = "ok" if p.returncode == 0 else f"exit status {p.returncode}"
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            out, _ = p.communicate()
            status = f"no result within {limit:.0f} s (killed)"
        line = None
        if rank == 0:
            for l in (out or b"").decode(errors="replace").splitlines():
                if l.startswith("{"):
                    line = json.loads(l)
        res[label] = dict(status=status, seconds=round(time.perf_counter() - t0, 1), line=line)
        print(f"[bench rank {rank}] {label}: {status} in {res[label]['seconds']} s", file=sys.stderr, flush=True)
    if rank != 0:
        return 0                                           # (the verdict is rank 0's to print: a launcher that sees another rank fail first tears rank 0 down)
    weak = res["weak"]["line"]
    want = (weak or {}).get("result_digest") or {}
    measured, failures, best = {}, [], None
    if weak is None:
        failures.append(f"weak leg: {res['weak']['status']}")
    for label, r in res.items():
        if label == "weak":
            continue
        line = r["line"]
        if line is None:
            measured[label] = dict(status=r["status"], valid=False)
            failures.append(f"scheme '{label}': {r['status']}")
            continue
        d = line["config"]["result_digest"]
        same = bool(want) and d.get("eref_rows") == want.get("eref_rows") and d.get("graph_and_components") == want.get("graph_and_components")
        bad = list(line.get("failed_checks") or []) + ([] if same else ["results differ from the one-GPU step's (digests)" if want else "no one-GPU digest to compare with"])
        measured[label] = dict(status=r["status"], valid=not bad, ms_per_step=line["ms_per_step"], contigs_per_s=line["value"],
                               eref_count_ms=(line.get("stage_ms") or {}).get("eref_count_both_sides"), **({"failed": bad} if bad else {}))
        if bad:
            failures.append(f"scheme '{label}': " + "; ".join(bad))
        elif best is None or line["ms_per_step"] < res[best]["line"]["ms_per_step"]:
            best = label
    if best is None:
        out = dict(metric=json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"], value=None, unit="contigs/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   higher_is_better=True, scaling="strong", vs_baseline=None, data="synthetic", config={"workload": "no configuration of the N-GPU step produced a valid result"})
    else:
        out = res[best]["line"]
        out.pop("failed_checks", None)
        out["config"]["parallelism"] = f"{world} GPUs, one sample: {best} (the fastest valid one of the configurations measured in this run)"
    out["parallelism_measured"] = measured
    out["parallelism_measured_note"] = ("every configuration is a child process per rank (own process group, the scheme forced) timing exactly --steps steps of the SAME sample; "
                                        "valid = its eref rows and graph / component digests equal the one-GPU step's (the weak leg's); `value` is the fastest valid one's")
    if weak is not None:
        out["weak"] = weak
    if failures:
        out["failed_checks"] = failures
    sys.stdout.flush()
    os.write(1, (json.dumps(out) + "\n").encode())
    if failures:
        print("bench.py: FAILED CHECKS: " + "; ".join(failures), file=sys.stderr)
        return 3
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 and not os.environ.get("PALACE_BENCH_CHILD") and os.environ.get("PALACE_BENCH_SCHEME", "auto") == "auto"
            and os.environ.get("PALACE_BENCH_ALL_SCHEMES", "1") == "1"):
        raise SystemExit(run_all_schemes(args))
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to fd 1 when a process group
    # initialises, so everything written to fd 1 before the final line is routed to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from bench.sample import progress
    progress("start")
    import torch
    progress("torch imported")
    from types import SimpleNamespace
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PALACE_BENCH_ONE_DEVICE") == "1":   # rehearsal only: every rank on GPU 0 (with PALACE_BENCH_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_exchange = os.environ.get("PALACE_FORCE_EXCHANGE") == "1"      # rehearse the N>1 code path on one GPU (reads sharded, table exchange)
    force_key_split = os.environ.get("PALACE_FORCE_KEY_SPLIT") == "1"    # ... and the key-space split with its gather (one rank: its share is everything)
    if world > 1 or force_exchange or force_key_split:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("PALACE_BENCH_BACKEND", "nccl")           # "nccl" is RCCL; gloo only for rehearsals
        if backend == "nccl":
            # the collectives of a step are small and sit between kernels of the two context streams: RCCL's own stream at high
            # priority, so that they are not queued behind the workgroups of a saturating count launch
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, pg_options=opts)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    E = SimpleNamespace(torch=torch, dev=dev, local=local, dist=dist, rank=rank, world=world,
                        force_exchange=force_exchange, force_key_split=force_key_split)
    from bench.step import measure
    if os.environ.get("PALACE_BENCH_TEST_HANG") and os.environ.get("PALACE_BENCH_TEST_HANG") == os.environ.get("PALACE_BENCH_SCHEME") and rank == world - 1:
        time.sleep(1e6)                                    # tests only: the last rank of this scheme's child never joins a collective
    if os.environ.get("PALACE_BENCH_LEG") == "weak" and world > 1:        # a child of run_all_schemes: the weak leg alone
        out, failures = measure(args, E, "weak")
    else:
        out, failures = measure(args, E, "strong" if world > 1 else "single")
    if world > 1 and os.environ.get("PALACE_BENCH_WEAK", "1") == "1" and not os.environ.get("PALACE_BENCH_LEG"):
        # the other reading of "N GPUs": one independent sample per GPU, no collective in the data path.  Every rank runs the
        # whole 1-GPU step on the (same) full sample; aggregate = N samples per max-over-ranks time.
        weak, _ = measure(args, E, "weak")
        if rank == 0:
            out["weak"] = weak
    if rank == 0:
        if failures:
            out["failed_checks"] = failures
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()
    if failures:
        print("bench.py: FAILED CHECKS: " + "; ".join(failures), file=sys.stderr)
        sys.exit(3)



if __name__ == "__main__":
    main()
