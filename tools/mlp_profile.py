#!/usr/bin/env python3
"""tools/mlp_profile.py -- diagnostic build of the fused MLP kernel (-DHZ_MLP_PROFILE): where a workgroup's cycles go.
Builds scratch copies with hipcc, runs them on the GPU, prints per-phase shader cycles of workgroup 100 (s_memtime).
Never quote the run time of this build: the stamps serialise what the real kernel overlaps."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    from _build import build
    out = build("libmlp_prof.so", {"hz_mlp.hip": ["-DHZ_MLP_PROFILE"] + [f for f in sys.argv[5:] if f.startswith("-D")]})
    import bench
    from hanabizero_amd._lib import MlpHeader
    from hanabizero_amd.config import make_config
    game = sys.argv[1] if len(sys.argv) > 1 else "Hanabi-Full"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    waves, tiles = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4, 4)
    cfg = make_config(game, simulations=50, stack=4)
    eng = bench.build_engine(cfg, torch.bfloat16, "cuda")
    f = eng.fused_shape(waves, tiles)
    if os.environ.get("HZ_PROFILE_UNIFORM"):  # n uniform 512 x 512 layers instead of the real chain (tools/mlp_chain_probe.py)
        from mlp_chain_probe import uniform_chain
        f = uniform_chain(eng, waves, tiles, int(os.environ["HZ_PROFILE_UNIFORM"]))
    hid = torch.rand(N, eng.H, device="cuda").to(torch.bfloat16)
    act = torch.randint(0, eng.A, (N,), device="cuda", dtype=torch.int32)
    h = torch.empty(N, eng.H, dtype=torch.bfloat16, device="cuda")
    r, v, p = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, eng.A, device="cuda")
    V, I, I64 = C.c_void_p, C.c_int, C.c_int64
    lib = C.CDLL(out)
    lib.hz_mlp_recurrent.argtypes = [C.POINTER(MlpHeader), V, V, V, V, V, I64, V, I64, V, V, V, V, V, I, I, V]
    lib.hz_mlp_profile_read.argtypes = [V]
    mt = f.rows_per_wg(N)
    for _ in range(5):
        rc = lib.hz_mlp_recurrent(C.byref(f.header), f.jobs.data_ptr(), f.weights.data_ptr(), f.biases.data_ptr(),
                                  f.act_table.data_ptr(), hid.data_ptr(), eng.H, None, 0, act.data_ptr(), h.data_ptr(),
                                  r.data_ptr(), v.data_ptr(), p.data_ptr(), N, mt, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    prof = np.zeros(128, np.uint64)
    lib.hz_mlp_profile_read(prof.ctypes.data_as(V))
    print("%s N=%d rows/WG=%d  (shader cycles of workgroup 100)" % (game, N, mt))
    for w in range(waves):
        o = prof[w * 8:w * 8 + 8]
        print("  wave %d: staging %6d | barriers %6d | job prologues %6d | k-loops %6d | epilogues %6d | final %6d | total %6d | k-steps %4d" % (
            w, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]))


    # per pass: when wave 0 got past the pass's barrier, and the weight bytes the workgroup streams in that pass
    lib.hz_mlp_profile_read_passes.argtypes = [V]
    pp = np.zeros(34, np.uint64)
    lib.hz_mlp_profile_read_passes(pp.ctypes.data_as(V))
    pp = pp.astype(np.int64)
    import ctypes
    from hanabizero_amd._lib import MlpJob
    raw = f.jobs.cpu().numpy().tobytes()
    tab = (MlpJob * (len(raw) // ctypes.sizeof(MlpJob))).from_buffer_copy(raw)
    t_prev = pp[32]
    print("  pass: ticks since the previous pass began | jobs | weight KB of the PREVIOUS pass | B/tick")
    prev_kb = 0.0
    for j in range(f.n_jobs):
        ents = [tab[j * waves + w] for w in range(waves)]
        kb = sum(e.ks for e in ents) * tiles * 1024 / 1024.0
        dt = pp[j] - t_prev
        print("  %2d: %7d | %2d jobs ks=%2d %s| %7.1f | %5.1f" % (j, dt, sum(1 for e in ents if e.ks), max(e.ks for e in ents),
              "B " if ents[0].flags & 4 else "  ", prev_kb, prev_kb * 1024 / max(dt, 1)))
        t_prev, prev_kb = pp[j], kb
    print("  end: %7d (last pass %.1f KB + final stage)" % (pp[33] - t_prev, prev_kb))


    # per job: when each wave got past the barrier / started its k-loop / ended it / ended its epilogue (relative to the first wave past the barrier)
    lib.hz_mlp_profile_read_timeline.argtypes = [V]
    tl = np.zeros(16 * 8 * 4, np.uint32)
    lib.hz_mlp_profile_read_timeline(tl.ctypes.data_as(V))
    tl = tl.reshape(16, 8, 4).astype(np.int64)[:waves]
    print("  per job (cycles after the first wave passed the job's barrier): barrier passed min..max | k-loop start min..max | k-loop end min..max | epilogue end min..max | next barrier released - last k-loop end")
    for j in range(min(8, f.n_jobs)):
        t0 = tl[:, j, 0].min()
        nxt = tl[:, j + 1, 0].min() - tl[:, j, 2].max() if j + 1 < min(8, f.n_jobs) else -1
        print("  job %d: %5d..%5d | %5d..%5d | %5d..%5d | %5d..%5d | %5d   k-loop per wave: %s" % (
            j, tl[:, j, 0].min() - t0, tl[:, j, 0].max() - t0, tl[:, j, 1].min() - t0, tl[:, j, 1].max() - t0, tl[:, j, 2].min() - t0,
            tl[:, j, 2].max() - t0, tl[:, j, 3].min() - t0, tl[:, j, 3].max() - t0, nxt, " ".join("%d" % x for x in (tl[:, j, 2] - tl[:, j, 1]))))


if __name__ == "__main__":
    main()
