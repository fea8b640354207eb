#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` kernel_stats.csv into a short
table (kernel names shortened) for profiles/.  Usage: summarize_rocprof.py <kernel_stats.csv> <out.md> [title]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?(ltxmi::(?:\w+::)*\w+(?:<[^>]*>)?)", name)
    if m:
        return m.group(1)
    m = re.match(r"(?:void )?(at::native::\w+)", name)
    if m:
        inner = re.search(r"(\w+_kernel_cuda|\w+Functor|CatArrayBatchedCopy\w*|uniform_kernel|normal_kernel)", name)
        return m.group(1) + ("[" + inner.group(1) + "]" if inner else "")
    return name[:80]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else src
    rows = list(csv.DictReader(open(src)))
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats --output-format csv`; total kernel time "
                f"{total / 1e6:.2f} ms\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | "
                    f"{float(r['AverageNs']) / 1e3:.1f} | {int(r['MinNs']) / 1e3:.1f} | {int(r['MaxNs']) / 1e3:.1f} | "
                    f"{float(r['Percentage']):.2f} |\n")


if __name__ == "__main__":
    main()
