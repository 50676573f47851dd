"""LLM.int8 projections of the 7b layer, one launch shape at a time, in a hipGraph over L layers of cold weights:
round 4's quantiser launch + pc_gemm (a8c) against pc_gemm_q8 (quantiser inside).  us per layer-launch, M = 1 and 12.

    python tools/q8_micro.py [--layers 12] [--rows 1,12] [--down 4,4;2,2]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402
from promptcache_amd.model.llama_hip import prime_graph_capture  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=12)
ap.add_argument("--rows", default="1,12")
ap.add_argument("--down", default="4,4;1,1;2,1")
ap.add_argument("--which", default="qkv,o,gu,down")
args = ap.parse_args()
dev = "cuda:0"
n.load()
H, Hkv, D, hid, inter = 32, 32, 128, 4096, 11008
L = args.layers
torch.manual_seed(0)


def wimg(N, K, perm=None):
    w = torch.randn((N, K), device=dev) * 0.02
    q, sc = n.quantize_rows_int8(w)
    if perm is not None:
        return n.to_weight_frags_i8(q[perm].contiguous()), sc[perm].contiguous(), q.t().contiguous()
    return n.to_weight_frags_i8(q), sc, q.t().contiguous()


perm = n.qkv_rope_row_perm(H + 2 * Hkv, D).to(dev)
perm32 = perm.to(torch.int32)
which = args.which.split(",")
W = {}
if "qkv" in which:
    W["qkv"] = [wimg(3 * hid, hid, perm) for _ in range(L)]
if "o" in which:
    W["o"] = [wimg(hid, hid) for _ in range(L)]
if "gu" in which:
    W["gu"] = [wimg(2 * inter, hid) for _ in range(L)]
if "down" in which:
    W["down"] = [wimg(hid, inter) for _ in range(L)]
gam = (1.0 + 0.1 * torch.randn(hid, device=dev)).half()
eps = 1e-5


def timed(fn, reps=30):
    prime_graph_capture(torch.device(dev))
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] / L * 1e6


for T in [int(v) for v in args.rows.split(",")]:
    x = torch.randn((T, hid), device=dev) * 1.2
    planes = lambda k: torch.zeros((1, k // 32, 64, 8), dtype=torch.float16, device=dev)
    img = lambda k: torch.zeros((1, k // 64, 64, 16), dtype=torch.int8, device=dev)
    xs = torch.zeros(16, dtype=torch.float32, device=dev)
    fl = torch.zeros((4, 16384), dtype=torch.uint8, device=dev)
    zero_h, zero_i = planes(hid), planes(inter)
    cs = torch.zeros((T, D // 2, 2), dtype=torch.float32, device=dev)
    pos = torch.arange(T, dtype=torch.int32, device=dev)
    inv = (1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float64) / D))).float().to(dev)
    n.rope_table(pos, inv, cs, T, D)
    cap = T + 8
    arena = torch.zeros((1, 2, Hkv, cap, D), dtype=torch.float16, device=dev)
    qh = torch.zeros((T, H * D), dtype=torch.float16, device=dev); ql = torch.zeros_like(qh)
    res = {}
    if "qkv" in which:
        xh, xq, x8 = planes(hid), planes(hid), img(hid)

        def old_qkv():
            for wf, sc, qt in W["qkv"]:
                n.rmsnorm_quant_i8(x, gam, eps, T, hid, xh, xq, xs, fl[0], fl[1], codes8=x8)
                n.gemm_qkv_rope_a8c(wf, sc, xq, zero_h, xs, fl[0], xh, qt, perm32, T, hid, cs, qh, ql, H * D, arena[:, 0], arena[:, 1],
                                    2 * Hkv * cap * D, cap * D, 1, H, Hkv, D, T, 0, cap, codes8=x8)

        def new_qkv():
            for wf, sc, qt in W["qkv"]:
                n.gemm_q8(epilogue=n.EPI_QKV_ROPE, wf=wf, w_scale=sc, w_codes_t=qt, row_perm=perm32, x=x, norm_weight=gam, eps=eps, M=T, K=hid,
                          cs=cs, q_hi=qh, q_lo=ql, q_token_stride=H * D, k_arena=arena[:, 0], v_arena=arena[:, 1],
                          arena_batch_stride=2 * Hkv * cap * D, arena_head_stride=cap * D, B=1, H=H, Hkv=Hkv, D=D, q_len=T, past_len=0, cap=cap)
        res["qkv"] = (timed(old_qkv), timed(new_qkv))
    if "o" in which:
        ah = (torch.randn((1, hid // 32, 64, 8), device=dev) * 0.5).half()
        aq, a8 = planes(hid), img(hid)
        y = torch.zeros((T, hid), dtype=torch.float32, device=dev)

        def old_o():
            for wf, sc, qt in W["o"]:
                n.quant_act_i8(ah, True, T, hid, aq, xs, fl[1], fl[2], codes8=a8)
                n.gemm_skinny_a8c(wf, sc, aq, zero_h, xs, fl[1], ah, qt, T, hid, hid, n.EPI_ADD, y=y, ldy=hid, codes8=a8)

        def new_o():
            for wf, sc, qt in W["o"]:
                n.gemm_q8(epilogue=n.EPI_ADD, wf=wf, w_scale=sc, w_codes_t=qt, xf_hi=ah, M=T, N=hid, K=hid, y=y, ldy=hid, flags_clear=fl[3],
                          clear_bytes=16384)
        res["o"] = (timed(old_o), timed(new_o))
    ch, cl = planes(inter), planes(inter)
    pm = torch.zeros((inter // 16, 16), dtype=torch.float32, device=dev)
    if "gu" in which:
        xh, xq, x8 = planes(hid), planes(hid), img(hid)

        def old_gu():
            for wf, sc, qt in W["gu"]:
                n.rmsnorm_quant_i8(x, gam, eps, T, hid, xh, xq, xs, fl[2], fl[3], codes8=x8)
                n.gemm_skinny_a8c(wf, sc, xq, zero_h, xs, fl[2], xh, qt, T, 2 * inter, hid, n.EPI_SILU, of_hi=ch, of_lo=cl, codes8=x8)

        def new_gu():
            for wf, sc, qt in W["gu"]:
                n.gemm_q8(epilogue=n.EPI_SILU, wf=wf, w_scale=sc, w_codes_t=qt, x=x, norm_weight=gam, eps=eps, M=T, N=2 * inter, K=hid,
                          of_hi=ch, row_max_out=pm, flags_out=fl[3])
        res["gu"] = (timed(old_gu), timed(new_gu))
    if "down" in which:
        ch.copy_((torch.randn((1, inter // 32, 64, 8), device=dev) * 0.5).half())
        cq, c8 = planes(inter), img(inter)
        y = torch.zeros((T, hid), dtype=torch.float32, device=dev)
        act = n.from_act_frags(ch, 16).float()
        pm.copy_(act.abs().view(16, inter // 16, 16).amax(dim=2).t())
        fl[3].zero_()
        scr = torch.empty(n.gemm_skinny_ks_scratch_bytes(hid, 8) // 4, dtype=torch.float32, device=dev)
        ctr = torch.zeros(hid // 16, dtype=torch.int32, device=dev)

        def old_down():
            for wf, sc, qt in W["down"]:
                n.quant_act_i8(ch, True, T, inter, cq, xs, fl[0], fl[1], codes8=c8)
                n.gemm_skinny_a8c(wf, sc, cq, zero_i, xs, fl[0], ch, qt, T, hid, inter, n.EPI_ADD, y=y, ldy=hid, codes8=c8)
        r = [timed(old_down)]
        for cfg in args.down.split(";"):
            tiles, slices = (int(v) for v in cfg.split(","))

            def new_down():
                for wf, sc, qt in W["down"]:
                    n.gemm_q8(epilogue=n.EPI_ADD, wf=wf, w_scale=sc, w_codes_t=qt, xf_hi=ch, row_max=pm, row_max_units=inter // 16, flags_in=fl[3],
                              M=T, N=hid, K=inter, y=y, ldy=hid, ks_tiles=tiles, kslices=slices, ks_scratch=scr, ks_scratch_bytes=scr.numel() * 4,
                              ks_counters=ctr)
            r.append((cfg, timed(new_down)))
        res["down"] = r
    print(f"rows {T}: " + "  ".join(f"{k} {v}" if k == "down" else f"{k} old {v[0]:.1f} new {v[1]:.1f}" for k, v in res.items()), flush=True)
