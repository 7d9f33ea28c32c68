#!/usr/bin/env python3
"""tools/net_error_report.py -- the error of the inference paths bench.py times, per dtype and game, as one JSON document:
tests/netgold.py::golden_net_error (against the reference nets' fp32 outputs and against the reference under fp16 autocast)
and ::search_divergence (what the format does to visit counts).  Runs on the GPU box; `--out profiles/rNN_net_error.json`."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--roots", type=int, default=1024)
    args = ap.parse_args()
    import torch
    from tests.netgold import golden_net_error, search_divergence
    doc = {}
    for game in ("Hanabi-Small", "Hanabi-Full"):
        for name, dt in (("fp32", torch.float32), ("fp16", torch.float16), ("bf16", torch.bfloat16)):
            e = golden_net_error(game, dt)
            e["search"] = search_divergence(game, dt, roots=args.roots)
            doc["%s/%s" % (game, name)] = e
            flat = {k: "%.2e/%.2e" % (v["max"], v["mean"]) for k, v in e.items() if isinstance(v, dict) and "max" in v}
            print(game, name, "worst %.3g" % e["worst"], "vs autocast %.3g" % e["vs_reference_autocast"]["worst"],
                  "(reference autocast vs fp32 %.3g)" % e["reference_autocast_vs_fp32"]["worst"], flat, e["search"], flush=True)
            print("   wide rms, ours / reference autocast:", {k: "%.2e / %.2e" % (v["rms"], e["wide"]["reference_autocast_vs_fp32"][k]["rms"])
                                                             for k, v in e["wide"]["got_vs_fp32"].items()}, flush=True)
    text = json.dumps(doc, indent=1, sort_keys=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
