import torch, time
n = 113*1024*1024//8
bufs = [torch.empty(n, dtype=torch.float64, device='cuda') for _ in range(6)]
src = torch.randn(n, dtype=torch.float64, device='cuda')
def t(fn, it=30):
    for _ in range(3): fn(0)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(it): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/it
tf = t(lambda i: bufs[i%6].fill_(1.5))
tc = t(lambda i: bufs[i%6].copy_(src))
tm = t(lambda i: torch.mul(src, 2.0, out=bufs[i%6]))
print("fill  %.1f us  %.2f TB/s written" % (tf*1e6, n*8/tf/1e12))
print("copy  %.1f us  %.2f TB/s written (%.2f total)" % (tc*1e6, n*8/tc/1e12, 2*n*8/tc/1e12))
print("mul   %.1f us  %.2f TB/s written" % (tm*1e6, n*8/tm/1e12))
