#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/...) into a small text summary for profiles/.
usage: rocprof_summary.py <out.md> --stats <kernel_stats.csv> [--trace <kernel_trace.csv>] [--pmc NAME=<counter_collection.csv> ...] [--note TEXT]"""
import collections
import csv
import sys


def short(name):
    """kernel name without its parameter list; kernels of an unnamed namespace keep their own name"""
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")



def main():
    out, args = sys.argv[1], sys.argv[2:]
    lines = []
    i = 0
    while i < len(args):
        if args[i] == "--note":
            lines.append(args[i + 1] + "\n")
        elif args[i] == "--stats":
            lines.append("## rocprofv3 --kernel-trace --stats (palace kernels + top others)\n")
            lines.append("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|")
            rows = list(csv.DictReader(open(args[i + 1])))
            keep = [r for r in rows if "palace::" in r["Name"]] + [r for r in rows if "palace::" not in r["Name"]][:4]
            for r in keep:
                nm = short(r["Name"])[:70]
                lines.append(f"| {nm} | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | "
                             f"{float(r['TotalDurationNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")
            lines.append("")
        elif args[i] == "--trace":
            # per-kernel durations from the kernel trace: median next to the mean (a profiled run now and then
            # shows one call several times longer than the rest; the median is what the unprofiled bench sees)
            import statistics
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(args[i + 1])):
                dur[short(r["Kernel_Name"])[:70]].append(
                    (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            lines.append("## rocprofv3 --kernel-trace: per-kernel duration (palace kernels)\n")
            lines.append("| kernel | calls | median ms | mean ms | max ms |\n|---|---|---|---|---|")
            for k, v in sorted(dur.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1])):
                if "palace::" in k:
                    lines.append(f"| {k} | {len(v)} | {statistics.median(v):.4f} | {sum(v)/len(v):.4f} | {max(v):.4f} |")
            lines.append("")
        elif args[i] == "--pmc":
            name, path = args[i + 1].split("=", 1)
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] == name and "palace::" in r["Kernel_Name"]:
                    agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            lines.append(f"## rocprofv3 --pmc {name} (own pass; per-dispatch mean, counter unit = KiB)\n")
            lines.append("| kernel | dispatches | mean value (KiB) | mean GB |\n|---|---|---|---|")
            for k, v in agg.items():
                m = sum(v) / len(v)
                lines.append(f"| {k[:70]} | {len(v)} | {m:.1f} | {m*1024/1e9:.3f} |")
            lines.append("")
        i += 2
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
