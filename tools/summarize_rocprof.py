#!/usr/bin/env python3
"""tools/summarize_rocprof.py <kernel_stats.csv> [bench-json-line-file] -> markdown summary for profiles/."""
import csv
import json
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows[:40]:
        name = r["Name"].replace("|", "/")
        if len(name) > 110:
            name = name[:107] + "..."
        print("| `%s` | %s | %.3f | %.2f | %.2f | %.2f | %.1f |" % (
            name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    print("\ntotal kernel time: %.3f ms over %d distinct kernels" % (tot / 1e6, len(rows)))
    if len(sys.argv) > 2:
        for line in open(sys.argv[2]):
            if line.startswith("{") and "selfplay_moves" in line:
                d = json.loads(line)
                print("\nbench line of the profiled run (profiling slows the run; do not compare `value` with an un-profiled run):\n")
                print("```json\n%s\n```" % json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
