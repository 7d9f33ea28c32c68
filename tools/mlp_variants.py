#!/usr/bin/env python3
"""tools/mlp_variants.py -- experiment harness for the fused MLP kernel: builds scratch variants of hz_mlp.hip with
-D switches (epilogues / final stage / staging removed, other ring depths), times each with hipGraph replays on random
inputs, prints us per launch.  Variants with parts removed compute garbage: they only say what that part costs."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STRIP = ["-DHZ_MLP_X_NOEPI", "-DHZ_MLP_X_NOFINAL", "-DHZ_MLP_X_NOSTAGE"]
RING8 = ["-DHZ_RING_WIDE=8"]  # (only workgroups of <= 8 waves have the registers for it: no effect on 16 x 2)
NOAV = ["-DHZ_MLP_X_NOAV"]
VARIANTS = [("baseline", []), ("no acc start loads", NOAV), ("no epilogues", ["-DHZ_MLP_X_NOEPI"]),
            ("no acc start loads, no epilogues", NOAV + ["-DHZ_MLP_X_NOEPI"]), ("no acc start / epi / final / staging", NOAV + STRIP)]
SHAPES = [(16, 2), (8, 4)]


def main():
    import bench
    from hanabizero_amd._lib import MlpHeader
    from hanabizero_amd.config import make_config
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cfg = make_config("Hanabi-Full", simulations=50, stack=4)
    eng = bench.build_engine(cfg, torch.bfloat16, "cuda")
    S = 8
    pool = torch.rand(S, N, eng.H, device="cuda").to(torch.bfloat16)
    ix = torch.randint(0, S, (N,), device="cuda", dtype=torch.int32)
    act = torch.randint(0, eng.A, (N,), device="cuda", dtype=torch.int32)
    h = torch.empty(N, eng.H, dtype=torch.bfloat16, device="cuda")
    r, v, p = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, eng.A, device="cuda")
    V, I, I64 = C.c_void_p, C.c_int, C.c_int64
    src = os.path.join(ROOT, "hanabizero_amd", "csrc")
    for name, flags, f in [(n + " %dx%d" % sh, fl, eng.fused_shape(*sh)) for sh in SHAPES for n, fl in VARIANTS
                           if not (sh[0] > 8 and "ring 8" in n)]:
        mt = f.rows_per_wg(N)
        out = os.path.join(ROOT, "gpurun_out", "libmlp_var_%s.so" % "".join(c for c in name if c.isalnum()))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-w", "-I" + src, "-I" + os.path.join(ROOT, "include"), "-o", out,
                               os.path.join(src, "hz_mlp.hip"), os.path.join(src, "hz_tree.hip")] + flags)
        lib = C.CDLL(out)
        lib.hz_mlp_recurrent.argtypes = [C.POINTER(MlpHeader), V, V, V, V, V, I64, V, I64, V, V, V, V, V, I, I, V]

        def launch():
            rc = lib.hz_mlp_recurrent(C.byref(f.header), f.jobs.data_ptr(), f.weights.data_ptr(), f.biases.data_ptr(),
                                      f.act_table.data_ptr(), pool.data_ptr(), pool.stride(1), ix.data_ptr(),
                                      pool.stride(0), act.data_ptr(), h.data_ptr(), r.data_ptr(), v.data_ptr(),
                                      p.data_ptr(), N, mt, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3):
                launch()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.replay()
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
        print("%-40s %7.2f us/launch" % (name, e0.elapsed_time(e1) * 1e3 / 100), flush=True)


if __name__ == "__main__":
    main()
