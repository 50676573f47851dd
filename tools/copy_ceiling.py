import torch, time
n = 904396800 // 2
a = torch.empty(n, dtype=torch.float16, device="cuda").normal_()
b = torch.empty_like(a)
for name, fn in (("torch copy_ (contiguous 904 MB)", lambda: b.copy_(a)),):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{name}: median {ts[len(ts)//2]*1e3:.1f} us  min {ts[0]*1e3:.1f} us -> {2*n*2/ (ts[len(ts)//2]*1e-3)/1e9:.0f} GB/s (read+write)")
# strided destination like the staged arena: [L*2*H, S, D] into [L*2*H, max_ctx, D]
L2H, S, D, cap = 32*2*32, 1725, 128, 4096
src = torch.empty((L2H, S, D), dtype=torch.float16, device="cuda").normal_()
dst = torch.empty((L2H, cap, D), dtype=torch.float16, device="cuda")
fn = lambda: dst[:, :S].copy_(src)
for _ in range(5): fn()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"torch strided copy_ [2048,1725,128] -> [2048,4096,128]: median {ts[10]*1e3:.1f} us -> {2*src.numel()*2/(ts[10]*1e-3)/1e9:.0f} GB/s")
