#!/usr/bin/env python3
"""The two tables of DESIGN.md section 6 from palace_amd/multigpu.py's cost model (markdown on stdout): Phase A + its exchange per
scheme, and the whole step (stream A of a rank against stream B of rank 0) under the scheme the model picks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from palace_amd import multigpu as mg   # noqa: E402

READ_LEN = 150
reads_of = lambda contigs: 2 * (int(5e8 * contigs / 1_000_000) // READ_LEN)
samples = [("1M contigs (6.67 M reads)", 1_000_000, reads_of(1_000_000)), ("5M contigs (33.3 M reads)", 5_000_000, reads_of(5_000_000))]

print("| sample | W | replicate | key_split | shard_reads | picked |\n|---|---|---|---|---|---|")
for name, nc, nr in samples:
    for W in (2, 4, 8):
        m = mg.phase_a_model(nr, W)
        ms = m["ms"]
        print(f"| {name if W == 2 else ''} | {W} | {ms['replicate']:.1f} | {ms.get('key_split', float('nan')):.1f} | {ms['shard_reads']:.1f} | {m['choice']} |")
print()
print("| sample | W | scheme | stream A | stream B (rank 0) | step | vs one GPU |\n|---|---|---|---|---|---|---|")
for name, nc, nr in samples + [("long contigs (100k, reads of the 1M config)", 100_000, reads_of(1_000_000))]:
    one = mg.step_model(nc, nr, 1)
    for W in (1, 2, 4, 8):
        b = mg.best_step(nc, nr, W) if W > 1 else dict(one, scheme="—")
        sch = b["scheme"] + ("" if b.get("rank0_counts", True) or W == 1 else ", rank 0 idle in Phase A")
        print(f"| {name if W == 1 else ''} | {W} | {sch} | {b['stream_a_ms']:.1f} | {b['stream_b_rank0_ms']:.1f} | {b['step_ms']:.1f} | {one['step_ms'] / b['step_ms']:.2f} × |")
print()
print("opt-in scheme shard_counts (reads sharded, partial counts of the DB's probe-index entries exchanged, no plane moved):")
print("| sample | W | rank 0 | stream A | stream B (rank 0) | step | vs one GPU | the picked scheme above |\n|---|---|---|---|---|---|---|---|")
for name, nc, nr in samples + [("long contigs (100k, reads of the 1M config)", 100_000, reads_of(1_000_000))]:
    one = mg.step_model(nc, nr, 1)
    for W in (2, 4, 8):
        b = min((mg.step_model(nc, nr, W, "shard_counts", r0) for r0 in (True, False)), key=lambda c: c["step_ms"])
        old = mg.best_step(nc, nr, W)
        print(f"| {name if W == 2 else ''} | {W} | {'counts' if b['rank0_counts'] else 'idle in Phase A'} | {b['stream_a_ms']:.1f} | {b['stream_b_rank0_ms']:.1f} | {b['step_ms']:.1f} | "
              f"{one['step_ms'] / b['step_ms']:.2f} × | {one['step_ms'] / old['step_ms']:.2f} × |")
