import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hanabizero_amd._lib import check, lib
def graph_time(fn, reps=30):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rows, cols, k, epi in ((256,512,32,0),(256,512,128,0),(256,512,512,0),(256,512,2048,0),(256,32,512,0),(32,32,32,0),(256,512,512,2),(512,512,256,2)):
    x = torch.randn(rows, k, device="cuda").bfloat16(); w = torch.randn(cols, k, device="cuda").bfloat16()
    o = torch.zeros(rows, cols, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
    def f():
        for _ in range(20):
            check(lib.hz_gemm_nt(x.data_ptr(), k, w.data_ptr(), k, None, rows, cols, k, o.data_ptr(), cols, epi, None, 1, st()), "g")
    def t():
        for _ in range(20):
            torch.mm(x, w.t())
    print(rows, cols, k, epi, "hz %.1f us   torch.mm %.1f us" % (1e6*graph_time(f)/20, 1e6*graph_time(t)/20))
