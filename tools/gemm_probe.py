import torch, time
dev="cuda"
def t(fn, it=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it
a=torch.randn(12,4096,device=dev,dtype=torch.float16)
w=torch.randn(12288,4096,device=dev,dtype=torch.float16)
try:
    o=torch.mm(a,w.t(),out_dtype=torch.float32); print("out_dtype ok", o.dtype, (o-a.float()@w.float().t()).abs().max().item())
except Exception as ex: print("out_dtype FAIL", repr(ex)[:200])
# rotate through many weight copies to defeat L2/MALL (256MB)
for (M,N,K) in [(12,12288,4096),(12,4096,4096),(12,22016,4096),(12,4096,11008),(1,12288,4096),(1,32000,4096),(32,12288,4096),(64,22016,4096),(256,22016,4096)]:
    n=max(2, int(600e6/(N*K*2))+1)
    ws=[torch.randn(N,K,device=dev,dtype=torch.float16) for _ in range(n)]
    x=torch.randn(M,K,device=dev,dtype=torch.float16)
    i=[0]
    def f():
        i[0]=(i[0]+1)%n
        return torch.mm(x,ws[i[0]].t())
    ms=t(f)
    print(f"mm M={M} N={N} K={K}: {ms*1e3:.1f} us  weight-stream {N*K*2/ms/1e6:.0f} GB/s  {2*M*N*K/ms/1e9:.1f} TF")
    del ws
