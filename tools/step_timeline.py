#!/usr/bin/env python3
"""tools/step_timeline.py <kernel_trace.csv> -- the kernels of one lock-step outside the simulation loop (everything
that is not k_mlp_recurrent / k_backprop_traverse), with start offsets, durations and gaps, from a rocprofv3
--kernel-trace CSV of bench.py.  The step is delimited by two consecutive k_actor_record_search launches."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_actor_record_search" in r["Kernel_Name"]]
    a, b = idx[-3], idx[-2]
    t0 = int(rows[a]["Start_Timestamp"])
    print("step span %.1f us, %d kernels" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, b - a))
    prev_end, busy, n = None, 0.0, 0
    for r in rows[a:b]:
        s, e, name = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]
        if "k_mlp_recurrent" in name or "k_backprop_traverse" in name:
            prev_end = e
            continue
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("%9.1f %7.1f gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name[:100]))
        busy += (e - s) / 1e3
        n += 1
        prev_end = e
    print("outside the simulation loop: %d kernels, %.1f us of kernel time" % (n, busy))


if __name__ == "__main__":
    main()
