"""dev: per-wave wall-clock stamps of ONE tile of the wide attention kernel's workgroup (0, 0, 0) (pc_dev_attn_trace):
per unit (A: the wave's first two row groups, B: its third) [start, QK done, softmax done, PV done], end of arithmetic, DMA drained, barrier passed,
next tile issued.  python tools/wide_trace.py [H S q]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n
H, S, q = (int(a) for a in (sys.argv[1:4] + ["40", "8258", "259"][len(sys.argv) - 1:]))
lib = n.load(); dev = "cuda:0"; Hkv, D = H, 128
cap = S + q + 64
arena = torch.randn((2, Hkv, cap, D), device=dev).half()
q16 = torch.randn((q, H * D), device=dev).half(); q16l = (torch.randn((q, H * D), device=dev) * 2 ** -11).half()
lo = torch.zeros((2, Hkv, 320, D), device=dev).half()
ah = torch.empty(((q + 15) // 16, H * D // 32, 64, 8), dtype=torch.float16, device=dev); al = torch.empty_like(ah)
ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q, S + q), 4) // 4, dtype=torch.float32, device=dev)
trace = torch.zeros(8 * 16, dtype=torch.int64, device=dev)
def run():
    n.attn_fwd(q16, q * H * D, H * D, arena[0], arena[1], 2 * Hkv * cap * D, cap * D, None, 0, 0, 1, H, Hkv, D, q, S, 1.0 / D ** 0.5, ws,
               out_frag=(ah, al), q_lo=q16l, kv_lo=(lo[0], lo[1], Hkv * 320 * D, 320 * D, -1))
for _ in range(3): run()
torch.cuda.synchronize()
lib.pc_dev_attn_trace(trace.data_ptr()); run(); lib.pc_dev_attn_trace(None)
torch.cuda.synchronize()
t = trace.view(8, 16).cpu().numpy()
base = t[:, 0].min()
names = ["A:start", "A:qk", "A:soft", "A:pv", "B:start", "B:qk", "B:soft", "B:pv"] + ["-"] * 4 + ["arith", "drained", "barrier", "issued"]
print("shader clock cycles (s_memtime); columns relative to the first wave's start")
print("wave " + " ".join(f"{x:>8s}" for x in names))
for w in range(8):
    print(f"{w:4d} " + " ".join(f"{(int(v) - int(base)) if v else 0:8d}" for v in t[w]))
