"""Every entry of profiles/r04_config5_loop.json from one run on the GPU box: tools/loop_bench.py under the settings named below,
one child process each (python tools/write_loop_profile.py [out.json])."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = {
    "one_rank_ratio_0.008": [],
    "one_rank_ratio_0.008_learner_blocks_through_autograd": ["--eager-blocks"],
    "one_rank_selfplay_and_ingest_only": ["--ratio", "0", "--rounds", "10"],
    "two_ranks_on_one_gpu_over_gloo_rehearsal": ["--gpus", "2", "--backend", "gloo", "--share-device", "--envs", "1024"],
    "one_rank_soak_30_rounds_small_replay_frequent_handovers": ["--rounds", "30", "--replay-capacity", "300000", "--checkpoint-interval", "100",
                                                                  "--target-interval", "50"],
    "Hanabi-Small_one_rank_ratio_0.008": ["--game", "Hanabi-Small"],
    "Hanabi-Full_2p_one_rank_ratio_0.008": ["--game", "Hanabi-Full"],
}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "config5_loop.json")
    rev = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    doc = {"_note": "tools/write_loop_profile.py on one MI355X box (gpurun)%s; BASELINE.json configs[4]: Hanabi-Full 5p, 50 simulations, "
                    "self-play + reanalyze + learner batch 256.  Each entry: `python tools/loop_bench.py` + its `args`.  r03's synchronous host "
                    "loop: profiles/r03_config5_loop.json (52 learner steps/s, 6.5 k moves/s; self-play + ingest alone 0.44 M moves/s; "
                    "hand-over 843 ms)." % (", tree at commit " + rev if rev else "")}
    for name, extra in RUNS.items():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "loop_bench.py")] + extra, cwd=ROOT, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            doc[name] = {"args": extra, "failed": r.returncode, "stderr_tail": r.stderr[-2000:]}
        else:
            doc[name] = dict(json.loads(lines[-1]), args=extra)
        print(name, doc[name].get("learner_steps_per_s"), doc[name].get("selfplay_moves_per_s"), flush=True)
        with open(out, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
