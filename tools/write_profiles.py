#!/usr/bin/env python3
"""tools/write_profiles.py <tag> -- profiles/<tag>_bench_kernel_stats_full{4096,8192}.md, profiles/<tag>_k_search_pmc.md and
profiles/pmc_traffic.json from what tools/profile_round.sh <tag> and tools/pmc_search.sh left under gpurun_out/ (run in the
authoring container after the gpurun call; the tree must be the one that was measured: the commit is recorded)."""
import csv
import glob
import io
import json
import os
import re
import subprocess
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

R01 = {  # round 1's values of the same counters (profiles/r01_k_search_pmc.md, r01_k_search_half_pmc.md; bf16 nets then, fp16 now)
    "cyc": ("4.5-4.7 M", "5.1-5.3 M"), "req": ("3.247e8", "3.242e8"), "rate": ("21.9 TB/s, 63 %", "53 %"), "lat": ("258", "224"),
    "tcc": ("89-92 %", "92 %"), "ta": ("69 %", "-"), "pend": ("23 %", "-"), "hit": ("95.6 %", "93.9 %"), "nmfma": ("3.83e7", "7.67e7"),
    "mfma": ("13 %", "23 %"), "wait": ("36 %", "35 %"), "lds": ("48 %", "47 %"), "hbm": ("1.58 GB", "-")}


def head():
    h = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    dirty = subprocess.call(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", "hanabizero_amd", "bench.py"]) != 0
    return h + ("+" if dirty else "")


def stats(tag, w, commit, note):
    import summarize_rocprof
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_stats_%s" % (tag, w), "**", "*kernel_stats.csv"), recursive=True)
    assert len(files) == 1, files
    js = os.path.join(ROOT, "gpurun_out", "%s_stats_%s.json" % (tag, w))
    d = json.loads([l for l in open(js) if l.startswith("{")][-1])
    rf = d["roofline"]
    buf = io.StringIO()
    sys.argv = ["summarize_rocprof.py", files[0], js]
    with redirect_stdout(buf):
        summarize_rocprof.main()
    # the dispatches bench.py's own figure is about: the last 32 launches of the search kernel in the trace (search_in_step's timed ones)
    tr = glob.glob(os.path.join(os.path.dirname(files[0]), "*kernel_trace.csv"))
    instep = ""
    if tr:
        rows = [r for r in csv.DictReader(open(tr[0])) if "k_search" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-32:]]
        instep = "rocprofv3's durations of those same 32 dispatches (the last 32 of the search kernel in the kernel trace): avg %.1f us, min %.1f us, max %.1f us.\n" % (
            sum(dur) / len(dur), min(dur), max(dur))
    cmd = "python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --workload %s" % w
    out = """# %s rocprofv3 --kernel-trace --stats: %s

Command (on the MI355X box, `tools/profile_round.sh %s`): `cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/%s_stats_%s -- %s`
(tree built from commit %s: %s).

State: Hanabi-Full 2p, %d envs, 49 simulations per move in ONE launch of the persistent search kernel (%d trees per workgroup); the whole
lock-step (8 kernels: three hipBLASLt GEMMs + `k_mlp_recurrent16` = the root inference, `k_prepare`, the search kernel,
`k_move_tail_a`, `k_move_tail_b`) is one hipGraph.  Search-kernel rows: 14 from the timed / warm-up / capture steps, 40 from bench.py's
roofline pass (further lock-steps of the live actor enqueued kernel by kernel, a HIP event on either side of the search launch:
`bench.py::search_in_step`), the rest from the launch-per-phase search that measures the mean path length; `k_backprop_traverse`,
`k_traverse`, `k_backprop`, `k_mlp_recurrent<.., 4, 4>` and the `__amd_rocclr_copyBuffer` rows come ONLY from that last pass and from the
tail timing's snapshots.
bench.py's own HIP-event figure for the same kernel in this run: avg %.1f us, shortest %.1f us (an event pair adds ~5 us of its own).
%s(The per-kernel averages of the table below are over ALL dispatches, the process's first launches and the warm-up steps included.)

%s""" % (tag, cmd, tag, tag, w, cmd, commit, note, d["config"]["envs_per_gpu"], rf["trees_per_workgroup"], rf["avg_launch_us"],
         rf["min_launch_us"], instep, buf.getvalue())
    open(os.path.join(ROOT, "profiles", "%s_bench_kernel_stats_%s.md" % (tag, w)), "w").write(out)
    row = [l for l in buf.getvalue().splitlines() if "k_search" in l][0]
    print(w, row[:160])


def pmc(tag, commit, note):
    def load(p):
        d = {}
        for line in open(p):
            m = re.match(r"(\S+)\s+launches\s+(\d+)\s+mean\s+(\S+)", line)
            if m:
                d[m.group(1)] = float(m.group(3))
        return d

    def col(d):
        cyc = d["TCC_CYCLE_sum"] / 128
        return dict(cyc=cyc, req=d["TCP_TCC_READ_REQ_sum"], gb=d["TCP_TCC_READ_REQ_sum"] * 128 / 1e9,
                    lat=d["TCP_TCC_READ_REQ_LATENCY_sum"] / d["TCP_TCC_READ_REQ_sum"], tcc=d["TCC_BUSY_avr"] / cyc, ta=d["TA_BUSY_avr"] / cyc,
                    hit=d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), mfma=d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
                    nmfma=d["SQ_INSTS_MFMA"], wait=d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], pend=d["TCP_PENDING_STALL_CYCLES_sum"] / 256 / cyc,
                    lds=d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"])
    A = col(load(os.path.join(ROOT, "gpurun_out", "%s_pmc_summary_4096.txt" % tag)))
    B = col(load(os.path.join(ROOT, "gpurun_out", "%s_pmc_summary_8192.txt" % tag)))
    tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    # the launch's wall time from the stats pass of the same tree -> the shader clock the chip ran at under this kernel
    clk = []
    for w, c in (("full4096", A), ("full8192", B)):
        js = os.path.join(ROOT, "gpurun_out", "%s_stats_%s.json" % (tag, w))
        d = json.loads([l for l in open(js) if l.startswith("{")][-1])
        clk.append(c["cyc"] / (d["roofline"]["avg_launch_us"] * 1e-6) / 1e9)
    r = lambda k, i: R01[k][i]
    rate = lambda c, g: c["gb"] / (c["cyc"] / (g * 1e9)) / 1e3
    md = """# %s hardware counters of the persistent search kernels (Hanabi-Full, 49 simulations per launch)

Command (on the MI355X box, repo root): `bash tools/pmc_search.sh` (4096 envs: `k_search<ElF16>`, 256 workgroups x 16 trees) and
`bash tools/pmc_search.sh --workload full8192` (`k_search_half<ElF16, 16>`, 256 workgroups x 32 trees): one `rocprofv3 --pmc <group>
--kernel-include-regex k_search --output-format csv` pass per counter group (no trace options) over `python bench.py --steps 3 --warmup 1
--no-cpu-baseline --no-roofline --no-also`; means over the 7 launches of each pass; tree at commit %s (%s).
`_sum` = over the 256 TCPs / 128 L2 channels, `_avr` = per instance.  Round 1's values of the same counters (profiles/r01_k_search_pmc.md,
r01_k_search_half_pmc.md) in brackets.

| counter | 4096 envs (16 trees / workgroup) | 8192 envs (32 trees / workgroup) |
|---|---|---|
| launch, shader cycles (`TCC_CYCLE_sum` / 128) | %.2f M [%s] | %.2f M [%s] |
| launch duration (bench.py's HIP events, stats pass of the same tree) -> mean shader clock under this kernel | %.2f GHz | %.2f GHz |
| `TCP_TCC_READ_REQ_sum` (128-B L1 -> L2 requests) | %.3e = %.1f GB [%s] | %.3e = %.1f GB [%s] |
| -> L1 <- L2 bytes per shader cycle per CU (64 = the L1's fill width) | %.1f | %.1f |
| -> L1 <- L2 rate over the launch at that clock | %.1f TB/s [%s at an assumed 2.4 GHz] | %.1f TB/s [%s] |
| mean L1-miss latency (`..LATENCY_sum` / requests) | %.0f cycles [%s] | %.0f cycles [%s] |
| `TCC_BUSY_avr` / launch (L2 channels busy) | %.0f %% [%s] | %.0f %% [%s] |
| `TA_BUSY_avr` / launch (texture addresser) | %.0f %% [%s] | %.0f %% [%s] |
| `TCP_PENDING_STALL_CYCLES_sum` / 256 / launch | %.0f %% [%s] | %.0f %% [%s] |
| L2 hit rate (`TCC_HIT` / (`HIT` + `MISS`)) | %.1f %% [%s] | %.1f %% [%s] |
| `SQ_INSTS_MFMA` | %.3e [%s] | %.3e [%s] |
| MFMA pipes busy (`SQ_VALU_MFMA_BUSY_CYCLES` / (1024 SIMDs x launch)) | %.1f %% [%s] | %.1f %% [%s] |
| `SQ_WAIT_INST_ANY` / `SQ_WAVE_CYCLES` | %.0f %% [%s] | %.0f %% [%s] |
| `SQ_LDS_BANK_CONFLICT` / `SQ_LDS_IDX_ACTIVE` | %.0f %% [%s] | %.0f %% [%s] |
| HBM traffic per launch (profiles/pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, separate passes) | %.2f GB [%s] | %.2f GB [%s] |

Reading.  The request count is what it was (the weight stream: every workgroup pulls all 3.18 MB per simulation); the launches got shorter in
cycles because fewer of them pass with the stream standing still (DESIGN.md section 4: hand-scheduled k-loop, start values through the scalar
cache, no workgroup barrier between the inference's passes, fewer instructions in the tree phases).  Two things bound what is left.  (1) tools/l2_stream_bench.hip: a kernel that does nothing but
this stream, with the product's MFMAs and layer boundaries, reaches 54-55 B per shader cycle per CU (of the L1's 64); the product's inference
phases run at 47 (16 rows) / 44 (32 rows) -- tools/mlp_loop_bench.py -- and the tree phases (bound by instruction issue) stream nothing.  (2) The shader clock:
under these kernels the chip runs at the clock in row 2, not at 2.4 GHz (the synthetic stream holds 2.39 GHz at 16 rows and 2.2 GHz at 32; random
rather than constant weights alone cost it 11 %% of its rate at 32 rows): L1 and L2 are clocked with the shaders, so every GB/s figure of this
kernel scales with it.  The MFMA pipes are busy %.1f %% / %.1f %% of the time: with 16 / 32 rows per weight fragment the matrix cores cannot be the bound.
""" % (tag, commit, note, A["cyc"] / 1e6, r("cyc", 0), B["cyc"] / 1e6, r("cyc", 1), clk[0], clk[1],
       A["req"], A["gb"], r("req", 0), B["req"], B["gb"], r("req", 1),
       A["req"] * 128 / 256 / A["cyc"], B["req"] * 128 / 256 / B["cyc"],
       rate(A, clk[0]), r("rate", 0), rate(B, clk[1]), r("rate", 1),
       A["lat"], r("lat", 0), B["lat"], r("lat", 1), A["tcc"] * 100, r("tcc", 0), B["tcc"] * 100, r("tcc", 1),
       A["ta"] * 100, r("ta", 0), B["ta"] * 100, r("ta", 1), A["pend"] * 100, r("pend", 0), B["pend"] * 100, r("pend", 1),
       A["hit"] * 100, r("hit", 0), B["hit"] * 100, r("hit", 1), A["nmfma"], r("nmfma", 0), B["nmfma"], r("nmfma", 1),
       A["mfma"] * 100, r("mfma", 0), B["mfma"] * 100, r("mfma", 1), A["wait"] * 100, r("wait", 0), B["wait"] * 100, r("wait", 1),
       A["lds"] * 100, r("lds", 0), B["lds"] * 100, r("lds", 1), tr["full4096"]["k_search"] / 1e9, r("hbm", 0),
       tr["full8192"]["k_search"] / 1e9, r("hbm", 1), A["mfma"] * 100, B["mfma"] * 100)
    open(os.path.join(ROOT, "profiles", "%s_k_search_pmc.md" % tag), "w").write(md)
    print(md[:2600])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    note = sys.argv[2] if len(sys.argv) > 2 else "hand-scheduled k-loop, scalar-cache start values, blockwise layer boundaries"
    commit = head()
    for w in ("full4096", "full8192"):
        f = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_pmc_FETCH_SIZE_%s" % (tag, w), "**", "*counter_collection.csv"), recursive=True)
        wr = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_pmc_WRITE_SIZE_%s" % (tag, w), "**", "*counter_collection.csv"), recursive=True)
        assert len(f) == 1 and len(wr) == 1, (f, wr)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), f[0], wr[0], w], stdout=subprocess.DEVNULL)
    for w in ("full4096", "full8192"):
        stats(tag, w, commit, note)
    if os.path.exists(os.path.join(ROOT, "gpurun_out", "%s_pmc_summary_4096.txt" % tag)):
        pmc(tag, commit, note)
    else:
        print("no counter summaries of tools/pmc_search.sh for %s under gpurun_out/: profiles/%s_k_search_pmc.md not written" % (tag, tag))


if __name__ == "__main__":
    main()
