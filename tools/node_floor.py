import torch, time
x = torch.zeros(64, device='cuda'); y = torch.zeros(1<<20, device='cuda')
def bench(fn, n):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/50/n*1e6
print('tiny dependent kernels in graph: us/node', bench(lambda: x.add_(1), 200))
print('4MB kernels in graph: us/node', bench(lambda: y.add_(1), 200))
# eager back to back
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(2000): x.add_(1)
torch.cuda.synchronize(); print('eager tiny us', (time.perf_counter()-t)/2000*1e6)
