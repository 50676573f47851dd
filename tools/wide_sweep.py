"""Round 6: where the wide staged-key attention beats the ring kernel: launch pair time by (heads, staged keys, new rows), both paths,
same process.  python tools/wide_sweep.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = [(H, S, q) for H, S in ((40, 8258), (32, 4390), (32, 2048)) for q in (66, 130, 144, 160, 259, 288, 300, 400, 512)]
for H, S, q in shapes:
    out = []
    for nw in ("1", "0"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn_mid.py"), str(H), str(S), str(q)], capture_output=True, text=True,
                           env=dict(os.environ, PC_ATTN_NO_WIDE=nw, PC_ATTN_WIDE_MIN="1024"))
        try:
            out.append(float(r.stdout.split(":")[-1].split("us")[0]))
        except ValueError:
            print(f"H={H} S={S} q={q} NO_WIDE={nw}: FAILED\n{r.stderr[-600:]}", flush=True)
            out.append(float("nan"))
    print(f"H={H} S={S} q={q}: ring {out[0]:.1f} us, wide {out[1]:.1f} us  ({out[1] / out[0]:.2f}x)", flush=True)
