#!/usr/bin/env python3
"""profiles/phase_a_traffic.json from the two PMC passes of tools/prof_full.sh (each a ONE-step bench run, no warm-up):
HBM bytes per step and stage = sum over the stage's dispatches of FETCH_SIZE x 2 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads) + WRITE_SIZE, counters in KiB.  `bytes_per_launch` = the count launch
(Phase A), the figure of bench.py's `roofline.traffic`; `stages` = the other stages of the step for `roofline_stages`.
The file carries the library's build id (palace_version()); bench.py quotes it only for the build AND the workload it was
measured on.
With packed reads (bench.py --reads packed, the default) the kernels that make the bit streams run once at set-up, not per
launch: they are listed but left out of the sum.
usage: traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <source tag> [contigs] [workload] [reads]"""
import collections
import re
import csv
import ctypes
import json
import os
import sys

SETUP = ("mark_read_ends_kernel", "mark_dropped_kernel", "eref_streams_kernel")
STAGES = {
    "phase_a": SETUP + ("eref_usable_kernel", "eref_bin1_sort_kernel", "eref_bin2_kernel", "eref_lds_count_kernel"),
    "phase_b": ("eref_probe_sets_kernel", "eref_ehits_scatter_kernel", "eref_sentinel_words_kernel", "eref_gather_hits_kernel", "eref_need_kernel", "eref_ref_kernel", "eref_window_kernel", "seq_prefix_kernel"),
    "classify": ("graph_depth_select_kernel", "graph_classify_kernel"),
    "resolve": ("resolve_split_kernel", "resolve_pair_insert_kernel", "resolve_pair_apply_kernel", "compact_edges_kernel", "copy_number_kernel"),
    "stage04": ("st4_", "dec_", "scan_apply_kernel", "scan_partials_kernel", "scan_prefix_kernel"),
}
# kernels that only run while the bench sets up (index build, packing, by_rank, path arcs): never part of a step
NOT_A_STEP = ("eref_probe_index_kernel", "eref_bucket_prefix_kernel", "st4_by_rank_kernel", "st4_path_arcs_kernel")


def kernel_of(name):
    for part in name.replace("(anonymous namespace)::", "").split("palace::")[1:]:
        return part.split("<")[0].split("(")[0].strip()
    return None


FUSED = {"seen": 0}                                     # the count kernel ran with Phase B's probe in it (its third template flag)


def sums(path, counter):
    """kernel -> (sum over dispatches in bytes, dispatches)"""
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        m = re.search(r"eref_lds_count_kernel<true, true, (?:\(int\))?([12])>", r["Kernel_Name"].replace("(bool)1", "true"))
        if m:                                                   # 1: channel 0 rode along; 2: every entry set did and no plane was written
            FUSED["seen"] = max(int(FUSED["seen"]), int(m.group(1)))
        if r["Counter_Name"] == counter:
            k = kernel_of(r["Kernel_Name"])
            if k:
                agg[k][0] += float(r["Counter_Value"]) * 1024.0
                agg[k][1] += 1
    return agg


def build_id():
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        lib = ctypes.CDLL(os.path.join(root, "palace_amd", "libpalace_hip.so"))
        lib.palace_version.restype = ctypes.c_char_p
        return lib.palace_version().decode()
    except Exception as e:                                  # (a file without id is never quoted)
        return None


def main():
    fetch, write, out, tag = sys.argv[1:5]
    contigs = int(sys.argv[5]) if len(sys.argv) > 5 else 1_000_000
    workload = sys.argv[6] if len(sys.argv) > 6 else "default"
    reads = sys.argv[7] if len(sys.argv) > 7 else "packed"
    setup_only = SETUP if reads == "packed" else ()
    f, w = sums(fetch, "FETCH_SIZE"), sums(write, "WRITE_SIZE")
    stage_of = lambda k: next((s for s, pats in STAGES.items() if any(k == p or (p.endswith("_") and k.startswith(p)) for p in pats)), None)
    per_kernel, stages = {}, collections.defaultdict(lambda: {"fetch_x2": 0.0, "write": 0.0, "dispatches": 0})
    for k in sorted(set(f) | set(w)):
        if k in NOT_A_STEP:
            continue
        fx, wr, n = 2 * f[k][0] if k in f else 0.0, w[k][0] if k in w else 0.0, max(f[k][1] if k in f else 0, w[k][1] if k in w else 0)
        st = stage_of(k)
        if st == "phase_a":                                   # per launch: the per-dispatch mean (a step has one dispatch of each)
            per_kernel[k] = {"fetch_x2": fx / max(1, n), "write": wr / max(1, n)}
        if st and k not in setup_only:
            stages[st]["fetch_x2"] += fx if st != "phase_a" else fx / max(1, n)
            stages[st]["write"] += wr if st != "phase_a" else wr / max(1, n)
            stages[st]["dispatches"] += n
    total = stages["phase_a"]["fetch_x2"] + stages["phase_a"]["write"]
    json.dump({"bytes_per_launch": total, "per_kernel_bytes": per_kernel, "once_at_setup": list(setup_only),
               "stages": {s: dict(v, bytes=v["fetch_x2"] + v["write"]) for s, v in stages.items()},
               "source": tag, "contigs": contigs, "workload": workload, "reads": reads, "build": build_id(), "fused_probe": FUSED["seen"],
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of a one-step run; FETCH_SIZE x 2 + WRITE_SIZE "
                         "(KiB counters), summed over the dispatches of a stage"},
              open(out, "w"), indent=1)
    print(f"{total / 1e9:.2f} GB per count launch", {s: round((v['fetch_x2'] + v['write']) / 1e9, 3) for s, v in stages.items()})


if __name__ == "__main__":
    main()
