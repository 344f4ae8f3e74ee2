import sys, torch
n = int(sys.argv[1])
x = torch.zeros(1024, device="cuda")
y = torch.ones(1024, device="cuda")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    x.add_(y)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(n):
        x.add_(y)
print("captured", n, flush=True)
g.replay(); torch.cuda.synchronize()
print("ok", n, float(x[0]))
