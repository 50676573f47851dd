"""dev: the wide staged-key attention's partials, slice by slice, against a torch fp64 reference (which slice / row group is off?)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n
n.load()
H, S, q = (int(a) for a in (sys.argv[1:4] + ["40", "8258", "259"][len(sys.argv) - 1:]))
dev = "cuda:0"; D = 128; Hkv = H
g = torch.Generator(device=dev); g.manual_seed(1)
cap = S + q + 7
Q = torch.randn((q, H, D), device=dev, generator=g)
K = (0.7 * torch.randn((Hkv, cap, D), device=dev, generator=g)).half()
V = torch.randn((Hkv, cap, D), device=dev, generator=g).half()
qh = Q.half(); ql = (Q - qh.float()).half()
arena = torch.stack([K, V]).contiguous()
Kn = 0.7 * torch.randn((Hkv, q, D), device=dev, generator=g); Vn = torch.randn((Hkv, q, D), device=dev, generator=g)     # the pass's own rows: fp32
arena[0, :, S:S + q] = Kn.half(); arena[1, :, S:S + q] = Vn.half()
klo = (Kn - Kn.half().float()).half().unsqueeze(0).contiguous(); vlo = (Vn - Vn.half().float()).half().unsqueeze(0).contiguous()
if os.environ.get("NAN_TAIL"): arena[:, :, S + q:] = float("nan")
wsb = n.attn_workspace_bytes(1, H, D, q, S + q)
ws = torch.zeros(max(wsb, 4) // 4, dtype=torch.float32, device=dev)
mt = (q + 15) // 16
oh = torch.empty((mt, H * D // 32, 64, 8), dtype=torch.float16, device=dev); ol = torch.empty_like(oh)
n.attn_fwd(qh.view(q, H * D), q * H * D, H * D, arena[0].unsqueeze(0), arena[1].unsqueeze(0), 2 * Hkv * cap * D, cap * D, None, 0, 0,
           1, H, Hkv, D, q, S, 1.0 / np.sqrt(D), ws, out_frag=(oh, ol), q_lo=ql.view(q, H * D), kv_lo=(klo, vlo, Hkv * q * D, q * D, -1))
torch.cuda.synchronize()
got = (n.from_act_frags(oh, q).float() + n.from_act_frags(ol, q).float()).view(q, H, D)
# reference, fp64
Qd = (qh.double() + ql.double()).permute(1, 0, 2)              # [H, q, D]
Kd, Vd = arena[0, :, :S + q].double(), arena[1, :, :S + q].double()
Kd[:, S:] = Kn.double(); Vd[:, S:] = Vn.double()
s = torch.einsum("hqd,hkd->hqk", Qd, Kd) / np.sqrt(D)
mask = torch.arange(S + q, device=dev)[None, :] > (S + torch.arange(q, device=dev))[:, None]
s = s.masked_fill(mask[None], float("-inf"))
ref = torch.einsum("hqk,hkd->hqd", torch.softmax(s, -1), Vd).permute(1, 0, 2)
err = (got.double() - ref).abs()
print(f"H={H} S={S} q={q}: max err {err.max():.3e}  (max|ref| {ref.abs().max():.3f})")
print("err by row group:", [f"{err[i*16:(i+1)*16].max():.1e}" for i in range(mt)])
print("err by head (first 8):", [f"{err[:, h].max():.1e}" for h in range(min(H, 8))])
# partials: how many?
for ns in range(2, 17):
    if H * ns * q * (D + 2) * 4 == wsb:
        break
else:
    ns = None
print("workspace partials per row:", ns, "bytes", wsb)
if ns and os.environ.get("PC_ATTN_NO_WIDE") != "1":
    po = ws[:H * ns * q * D].view(H, ns, q, D); ml = ws[H * ns * q * D:H * ns * q * (D + 2)].view(H, ns, q, 2)
    T = (S + 63) // 64; tps = (T + ns - 2) // (ns - 1)
    for sp in range(ns):
        a, e = (sp * tps * 64, min((sp + 1) * tps * 64, S)) if sp < ns - 1 else (S, S + q)
        ss = s[:, :, a:e]
        m = ss.max(-1).values
        p = torch.exp(ss - m[..., None]); l = p.sum(-1)
        o = torch.einsum("hqk,hkd->hqd", p, Vd[:, a:e]) / l[..., None]
        og = po[:, sp].double() / ml[:, sp, :, 1:2].double()
        d = (og - o).abs()
        print(f"  slice {sp} keys [{a},{e}): max err of O/l {d.max():.2e}; by group {[f'{d[:, i*16:(i+1)*16].max():.0e}' for i in range(mt)]}")
