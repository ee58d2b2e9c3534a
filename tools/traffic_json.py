#!/usr/bin/env python3
"""profiles/phase_a_traffic.json from the two PMC passes of tools/prof_full.sh: HBM bytes per count_reads launch =
sum over the launch's kernels of FETCH_SIZE x 2 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the bytes
of wide coalesced reads) + WRITE_SIZE, counters in KiB, per-dispatch means.  bench.py quotes the number only for the
workload it was measured on.
With packed reads (bench.py --reads packed, the default) the kernels that make the bit streams run once at set-up, not per
launch: they are listed but left out of the sum.
usage: traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <source tag> [contigs] [workload] [reads]"""
import collections
import csv
import json
import sys

KERNELS = ("mark_read_ends_kernel", "mark_dropped_kernel", "eref_streams_kernel", "eref_usable_kernel", "eref_bin1_sort_kernel",
           "eref_bin2_kernel", "eref_lds_count_kernel")


def means(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            for k in KERNELS:
                if "palace::" + k in r["Kernel_Name"]:
                    agg[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * 1024.0 for k, v in agg.items()}


def main():
    fetch, write, out, tag = sys.argv[1:5]
    contigs = int(sys.argv[5]) if len(sys.argv) > 5 else 1_000_000
    workload = sys.argv[6] if len(sys.argv) > 6 else "default"
    reads = sys.argv[7] if len(sys.argv) > 7 else "packed"
    setup_only = ("mark_read_ends_kernel", "mark_dropped_kernel", "eref_streams_kernel") if reads == "packed" else ()
    f, w = means(fetch, "FETCH_SIZE"), means(write, "WRITE_SIZE")
    per = {k: {"fetch_x2": 2 * f.get(k, 0.0), "write": w.get(k, 0.0)} for k in KERNELS if k in f or k in w}
    total = sum(v["fetch_x2"] + v["write"] for k, v in per.items() if k not in setup_only)
    json.dump({"bytes_per_launch": total, "per_kernel_bytes": per, "once_at_setup": list(setup_only), "source": tag, "contigs": contigs,
               "workload": workload, "reads": reads,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE x 2 + WRITE_SIZE (KiB counters)"},
              open(out, "w"), indent=1)
    print(f"{total / 1e9:.2f} GB per launch", {k: round((v['fetch_x2'] + v['write']) / 1e9, 2) for k, v in per.items()})


if __name__ == "__main__":
    main()
