"""Which fp16 roundings of the many-row (encode / no-cache) path cost how much at full depth?  A torch fp32 emulation of
the Llama forward at the 7b shape with selectable fp16 roundings, compared with the all-fp32 forward (fp16 weights are
the inputs in every variant).  Prints max |dlogit| per variant."""
import math, sys
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda:0"
L, H, D, hid, inter, V = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 32, 128, 4096, 11008, 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
g = torch.Generator(device=dev); g.manual_seed(5)
def W(n, k): return (torch.randn((n, k), generator=g, device=dev) * 0.02).half().float()
def G(n): return (1 + 0.1 * torch.randn(n, generator=g, device=dev)).half().float()
layers = [dict(ln1=G(hid), wqkv=W(3 * hid, hid), wo=W(hid, hid), ln2=G(hid), wgu=W(2 * inter, hid), wd=W(hid, inter)) for _ in range(L)]
embed, norm, head = W(V, hid), G(hid), W(V, hid)
ids = torch.randint(3, V, (T,), generator=g, device=dev)
pos = torch.arange(T, device=dev).float()
inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, device=dev).float() / D))
ang = pos[:, None] * inv[None]; cos, sin = torch.cat([ang.cos()] * 2, 1), torch.cat([ang.sin()] * 2, 1)
r16 = lambda t: t.half().float()
def rms(x, w): return w * (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-5))
def rope(t): return t * cos[None] + torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1) * sin[None]
def fwd(act16, q16, kv16, p16, o16):
    x = embed[ids]
    for lw in layers:
        h = rms(x, lw["ln1"]); h = r16(h) if act16 else h
        qkv = h @ lw["wqkv"].t()
        q, k, v = (t.view(T, H, D).transpose(0, 1) for t in qkv.split(hid, dim=1))
        q, k = rope(q), rope(k)
        if q16: q = r16(q)
        if kv16: k, v = r16(k), r16(v)
        s = (q @ k.transpose(1, 2)) / math.sqrt(D)
        s = s.masked_fill(torch.triu(torch.ones(T, T, device=dev, dtype=torch.bool), 1), float("-inf"))
        p = torch.softmax(s, -1)
        if p16:                                   # the kernel rounds the un-normalised weights exp(s - m)
            m = s.max(-1, keepdim=True).values; e = torch.exp(s - m); p = r16(e) / e.sum(-1, keepdim=True)
        a = (p @ v).transpose(0, 1).reshape(T, hid); a = r16(a) if o16 else a
        x = x + a @ lw["wo"].t()
        h = rms(x, lw["ln2"]); h = r16(h) if act16 else h
        gu = h @ lw["wgu"].t()
        act = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]; act = r16(act) if act16 else act
        x = x + act @ lw["wd"].t()
    return rms(x, norm) @ head.t()
ref = fwd(False, False, False, False, False)
print(f"L={L} T={T} max|logit| {ref.abs().max():.2f}")
for name, cfg in [("all fp16 roundings (fast dense path)", (1, 1, 1, 1, 1)), ("split GEMM inputs + attn out; fp16 q,k,v,p", (0, 1, 1, 1, 0)),
                  ("+ q split", (0, 0, 1, 1, 0)), ("+ q, p split (only K/V fp16)", (0, 0, 1, 0, 0)),
                  ("only q fp16", (0, 1, 0, 0, 0)), ("only p fp16", (0, 0, 0, 1, 0)), ("only K/V fp16", (0, 0, 1, 0, 0)),
                  ("only GEMM inputs fp16", (1, 0, 0, 0, 1))]:
    print(f"{name:48s} max|dlogit| = {(fwd(*cfg) - ref).abs().max():.2e}")
