"""Time pc_attn_fwd / pc_attn_fwd_ex (+ combine) on the cached-prefill shape, with and without new-row K/V residuals.
usage: python tools/attn_probe.py [q_len] [past] [H] [D]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "prompt-cache_amd"))
import torch
from promptcache_amd import _native as n

q_len = int(sys.argv[1]) if len(sys.argv) > 1 else 12
past = int(sys.argv[2]) if len(sys.argv) > 2 else 1725
H = int(sys.argv[3]) if len(sys.argv) > 3 else 32
D = int(sys.argv[4]) if len(sys.argv) > 4 else 128
dev = "cuda:0"
B, Hkv, cap = 1, H, past + q_len + 64
T = q_len
torch.manual_seed(0)
arena = torch.randn((B, 2, Hkv, cap, D), device=dev).half()
q16 = torch.randn((T, H * D), device=dev).half(); q16l = (torch.randn((T, H * D), device=dev) * 1e-3).half()
klo = (torch.randn((B, Hkv, q_len, D), device=dev) * 1e-3).half(); vlo = klo.clone()
mt = (T + 15) // 16
fh = torch.zeros((mt, H * D // 32, 64, 8), dtype=torch.float16, device=dev); fl = torch.zeros_like(fh)
ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, past + q_len), 4) // 4, dtype=torch.float32, device=dev)
past_dev = torch.tensor([past], dtype=torch.int32, device=dev)

def run(kv_lo):
    n.attn_fwd(q16, q_len * H * D, H * D, arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D, None, 0, 0,
               B, H, Hkv, D, q_len, past, 1.0 / D ** 0.5, ws, past_len_dev=past_dev, out_frag=(fh, fl), q_lo=q16l, kv_lo=kv_lo)

for name, lo in (("plain", None), ("kv_lo", (klo, vlo, Hkv * q_len * D, q_len * D, -1))):
    for _ in range(20):
        run(lo)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 300
    e0.record()
    for _ in range(N):
        run(lo)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:6s} q={q_len} past={past}: {e0.elapsed_time(e1) / N * 1e3:.2f} us per attn+combine")
