import torch, time
torch.manual_seed(0)
N=4096
def bench(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
for dt in (torch.bfloat16,):
    x=torch.randn(N,512,device='cuda',dtype=dt); w=torch.randn(512,512,device='cuda',dtype=dt); b=torch.randn(512,device='cuda',dtype=dt)
    print("addmm", bench(lambda: torch.addmm(b,x,w)))
    print("addmm+relu_", bench(lambda: torch.relu_(torch.addmm(b,x,w))))
    try:
        y=torch._addmm_activation(b,x,w,use_gelu=False)
        ref=torch.relu(torch.addmm(b,x,w))
        print("_addmm_activation ok maxdiff", (y.float()-ref.float()).abs().max().item(), bench(lambda: torch._addmm_activation(b,x,w,use_gelu=False)))
    except Exception as e: print("no _addmm_activation", e)
    g=torch.cuda.CUDAGraph()
    s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): y=torch.relu_(torch.addmm(b,x,w))
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20):
            y=torch.relu_(torch.addmm(b,x,w))
    print("graph 20x(addmm+relu) per pair us", bench(lambda: g.replay(), 50)/20)
    x3=torch.randn(3,N,256,device='cuda',dtype=dt); w3=torch.randn(3,256,256,device='cuda',dtype=dt); b3=torch.randn(3,1,256,device='cuda',dtype=dt)
    print("baddbmm 3x", bench(lambda: torch.baddbmm(b3,x3,w3)))
    z=torch.randn(N,768,device='cuda',dtype=dt)
    zs=z.view(N,3,256).transpose(0,1)
    print("baddbmm strided", bench(lambda: torch.baddbmm(b3,zs,w3)))
    x2=torch.randn(N,256,device='cuda',dtype=dt); w2=torch.randn(256,256,device='cuda',dtype=dt); b2=torch.randn(256,device='cuda',dtype=dt)
    print("3 x addmm 256", bench(lambda: (torch.addmm(b2,x2,w2),torch.addmm(b2,x2,w2),torch.addmm(b2,x2,w2))))
    xk=torch.randn(N,544,device='cuda',dtype=dt); wk=torch.randn(544,512,device='cuda',dtype=dt)
    print("addmm K=544", bench(lambda: torch.addmm(b,xk,wk)))
    xk=torch.randn(N,576,device='cuda',dtype=dt); wk=torch.randn(576,512,device='cuda',dtype=dt)
    print("addmm K=576", bench(lambda: torch.addmm(b,xk,wk)))
    w768=torch.randn(512,768,device='cuda',dtype=dt); b768=torch.randn(768,device='cuda',dtype=dt)
    print("addmm 512->768", bench(lambda: torch.addmm(b768,x,w768)))
    # event overhead
    a=torch.cuda.Event(enable_timing=True); c=torch.cuda.Event(enable_timing=True)
    a.record(); c.record(); torch.cuda.synchronize(); print("empty event pair us", a.elapsed_time(c)*1e3)
    a.record(); y=torch.addmm(b,x,w); c.record(); torch.cuda.synchronize(); print("event around 1 addmm us", a.elapsed_time(c)*1e3)
    a.record()
    for _ in range(10): y=torch.addmm(b,x,w)
    c.record(); torch.cuda.synchronize(); print("event around 10 addmm us/each", a.elapsed_time(c)*1e2)
