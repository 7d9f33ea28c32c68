#!/usr/bin/env python3
"""tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <workload> -> profiles/pmc_traffic.json

HBM traffic per launch of the hand-written kernels from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes: counters are in KiB; on gfx950 FETCH_SIZE counts 64 B
per 128-B request for wide (16 B/lane) coalesced reads, so it is doubled before use.  The doubling is calibrated for wide
streaming reads only; k_backprop's narrow reads are reported with the same formula and flagged."""
import collections
import csv
import json
import os
import sys

KERNELS = {"k_search": ("k_search<", "k_search_half<", "k_search_turn<"), "k_move_tail_a": "k_move_tail_a", "k_move_tail_b": "k_move_tail_b", "k_traverse": "k_traverse(", "k_backprop": "k_backprop<", "k_backprop_traverse": "k_backprop_traverse<", "k_mlp_recurrent": "k_mlp_recurrent",
           "k_env_observe": "k_env_observe", "k_env_rules": "k_env_rules", "k_env_reset_rows": "k_env_reset_rows", "k_prepare": "k_prepare",
           "k_select_action": "k_select_action", "k_rows_scatter": "k_rows_scatter", "k_actor_draw": "k_actor_draw",
           "k_actor_record_search": "k_actor_record_search", "k_actor_record_step_slots": "k_actor_record_step_slots",
           "k_actor_flush": "k_actor_flush", "k_actor_begin_move_draw": "k_actor_begin_move_draw"}


def mean_by_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        for key, needle in KERNELS.items():
            if any(n in r["Kernel_Name"] for n in ((needle,) if isinstance(needle, str) else needle)):
                agg[key].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fetch, write, workload = mean_by_kernel(sys.argv[1]), mean_by_kernel(sys.argv[2]), sys.argv[3]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_path = os.path.join(root, "profiles", "pmc_traffic.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    entry = {"_note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction for wide coalesced "
                      "reads; uncalibrated for k_backprop's narrow reads), mean over the launches of the profiled run",
             "_detail": {}}
    for k in KERNELS:
        if k in fetch and k in write:
            f, nf = fetch[k]
            w, nw = write[k]
            entry[k] = (2 * f + w) * 1024
            entry["_detail"][k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "launches": min(nf, nw)}
    try:  # (run in the authoring container on the merged gpurun_out/ files: the commit the measured tree was built from)
        import subprocess
        entry["_commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        entry["_commit"] = None
    data[workload] = entry
    json.dump(data, open(out_path, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
