"""pc_gemm_chain (csrc/pc_gemm.hip): o_proj -> gate|up -> down_proj (-> the next layer's q|k|v) as ONE persistent launch with
in-kernel grid barriers, against the four stand-alone launches it replaces (pc_gemm_skinny epilogue 1, pc_gemm_skinny_norm
epilogue 2, pc_gemm_skinny epilogue 1, pc_gemm_qkv_rope_norm): same tiles, K split and reduction order -> BIT-identical.
The stand-alone launches are themselves checked against the oracle in tests/test_gpu_kernels.py."""
import numpy as np
import pytest
import torch

def _has_chain():
    from promptcache_amd import _native
    return _native.has("pc_gemm_chain")


# (round 6: pc_gemm_chain left the product library -- bit-identical to the four launches and slower in every round it was measured;
# it is built with PC_BUILD_FLAGS=-DPC_DEV_SWEEPS only, csrc/pc_dev.h)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _has_chain(), reason="pc_gemm_chain exists only in -DPC_DEV_SWEEPS builds (csrc/pc_dev.h)")]
DEV = "cuda:0"


def _n():
    from promptcache_amd import _native
    _native.load()
    return _native


def _inv_freq(D, theta):
    return torch.from_numpy((1.0 / (theta ** (np.arange(0, D, 2, dtype=np.float64) / D))).astype(np.float32))


class Layer:
    def __init__(self, n, hid, inter, H, Hkv, D, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)

        def w(r, c, s):
            return (s * torch.randn((r, c), generator=g)).half().to(DEV)
        self.wo = n.to_weight_frags(w(hid, H * D, 0.02))
        self.wgu = n.to_weight_frags(w(2 * inter, hid, 0.02))
        self.wdown = n.to_weight_frags(w(hid, inter, 0.02))
        perm = n.qkv_rope_row_perm(H + 2 * Hkv, D).to(DEV)
        self.wqkv = n.to_weight_frags(w((H + 2 * Hkv) * D, hid, 0.02)[perm].contiguous())
        self.ln1 = (1.0 + 0.1 * torch.randn(hid, generator=g)).half().to(DEV)
        self.ln2 = (1.0 + 0.1 * torch.randn(hid, generator=g)).half().to(DEV)


def _separate(n, L, x, ah, al, ch, cl, T, hid, inter, H, Hkv, D, B, q_len, past, cap, cs, q, ql, arena, with_qkv, eps,
              past_dev=None, kv_lo=None, lo_base=-1):
    n.gemm_skinny(L.wo, ah, al, T, hid, H * D, n.EPI_ADD, y=x, ldy=hid)
    n.gemm_skinny_norm(L.wgu, x, L.ln2, eps, T, 2 * inter, hid, n.EPI_SILU, of_hi=ch, of_lo=cl)
    n.gemm_skinny(L.wdown, ch, cl, T, hid, inter, n.EPI_ADD, y=x, ldy=hid)
    if with_qkv:
        n.gemm_qkv_rope_norm(L.wqkv, x, L.ln1, eps, T, hid, cs, q, ql, H * D, arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D,
                             B, H, Hkv, D, q_len, past, cap, past_dev, kv_lo=kv_lo, lo_base=lo_base)


def _chain(n, L, x, ah, al, ch, cl, T, hid, inter, H, Hkv, D, B, q_len, past, cap, cs, q, ql, arena, with_qkv, eps, sync,
           past_dev=None, kv_lo=None, lo_base=-1):
    qkv = None
    if with_qkv:
        qkv = dict(wqkv_f=L.wqkv, ln1=L.ln1, cs=cs, q_hi=q, q_lo=ql, q_ts=H * D, k_arena=arena[:, 0], v_arena=arena[:, 1],
                   a_bs=2 * Hkv * cap * D, a_hs=cap * D, B=B, H=H, Hkv=Hkv, D=D, q_len=q_len, past_len=past, cap=cap,
                   past_len_dev=past_dev, kv_lo=kv_lo, lo_base=lo_base)
    n.gemm_chain(L.wo, ah, al, H * D, x, T, hid, L.wgu, L.ln2, eps, inter, ch, cl, L.wdown, sync, qkv=qkv)


def _state(n, T, hid, inter, H, Hkv, D, B, q_len, past, cap, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (1.5 * torch.randn((T, hid), generator=g)).to(DEV)
    attn = torch.randn((T, H * D), generator=g).to(DEV)
    ah, al = n.to_act_frags(attn)
    ch = torch.zeros((1, inter // 32, 64, 8), dtype=torch.float16, device=DEV)
    cl = torch.zeros_like(ch)
    pos = torch.randint(0, 3000, (T,), generator=g, dtype=torch.int32).to(DEV)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(pos, _inv_freq(D, 10000.0).to(DEV), cs, T, D)
    q = torch.zeros((T, H * D), dtype=torch.float16, device=DEV)
    ql = torch.zeros_like(q)
    arena = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    return x, ah, al, ch, cl, cs, q, ql, arena


SHAPE_7B = dict(hid=4096, inter=11008, H=32, Hkv=32, D=128)
SHAPE_13B = dict(hid=5120, inter=13824, H=40, Hkv=40, D=128)


@pytest.mark.parametrize("shape", [SHAPE_7B, SHAPE_13B], ids=["7b", "13b"])
@pytest.mark.parametrize("B,q_len,with_qkv", [(1, 12, True), (1, 1, True), (2, 8, True), (1, 12, False), (1, 5, False)])
def test_chain_is_bit_identical_to_the_separate_launches(shape, B, q_len, with_qkv):
    n = _n()
    hid, inter, H, Hkv, D = (shape[k] for k in ("hid", "inter", "H", "Hkv", "D"))
    T, past = B * q_len, 37
    cap = past + q_len + 3
    eps = 1e-5
    L = Layer(n, hid, inter, H, Hkv, D, seed=5)
    sync = n.chain_sync_state(DEV)
    for rep in range(3):                     # the sync state carries over from launch to launch (no reset)
        a = _state(n, T, hid, inter, H, Hkv, D, B, q_len, past, cap, seed=100 + rep)
        b = [t.clone() for t in a]
        _separate(n, L, a[0], a[1], a[2], a[3], a[4], T, hid, inter, H, Hkv, D, B, q_len, past, cap, a[5], a[6], a[7], a[8], with_qkv, eps)
        _chain(n, L, b[0], b[1], b[2], b[3], b[4], T, hid, inter, H, Hkv, D, B, q_len, past, cap, b[5], b[6], b[7], b[8], with_qkv, eps, sync)
        torch.cuda.synchronize()
        assert n.chain_sync_error(sync) == 0
        names = ("x", "attn_hi", "attn_lo", "act_hi", "act_lo", "cs", "q_hi", "q_lo", "arena")
        for name, ta, tb in zip(names, a, b):
            assert torch.equal(ta, tb), (name, rep, float((ta.float() - tb.float()).abs().max()))
        assert torch.isfinite(a[0]).all()


def test_chain_replays_inside_a_hip_graph_and_with_a_device_side_past_length():
    """The product's use: captured once, replayed with new inputs; past length read from device memory; a residual tail."""
    n = _n()
    hid, inter, H, Hkv, D = (SHAPE_7B[k] for k in ("hid", "inter", "H", "Hkv", "D"))
    B, q_len, past, eps = 1, 3, 50, 1e-5
    T, cap = B * q_len, 80
    L = Layer(n, hid, inter, H, Hkv, D, seed=9)
    sync = n.chain_sync_state(DEV)
    past_dev = torch.tensor([past, 40], dtype=torch.int32, device=DEV)
    tail = torch.zeros((2, B, Hkv, 32, D), dtype=torch.float16, device=DEV)
    kv_lo = (tail[0], tail[1], Hkv * 32 * D, 32 * D)
    st = _state(n, T, hid, inter, H, Hkv, D, B, q_len, past, cap, seed=1)
    ref = [t.clone() for t in st]
    tail_ref = torch.zeros_like(tail)
    x_in = st[0].clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _chain(n, L, st[0], st[1], st[2], st[3], st[4], T, hid, inter, H, Hkv, D, B, q_len, past, cap, st[5], st[6], st[7], st[8], True,
               eps, sync, past_dev=past_dev, kv_lo=kv_lo, lo_base=-2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _chain(n, L, st[0], st[1], st[2], st[3], st[4], T, hid, inter, H, Hkv, D, B, q_len, past, cap, st[5], st[6], st[7], st[8], True,
               eps, sync, past_dev=past_dev, kv_lo=kv_lo, lo_base=-2)
    for step in range(4):
        past_dev[0] = past + step
        st[0].copy_(x_in + step)
        ref[0].copy_(x_in + step)
        g.replay()
        _separate(n, L, ref[0], ref[1], ref[2], ref[3], ref[4], T, hid, inter, H, Hkv, D, B, q_len, past, cap, ref[5], ref[6],
                  ref[7], ref[8], True, eps, past_dev=past_dev,
                  kv_lo=(tail_ref[0], tail_ref[1], Hkv * 32 * D, 32 * D), lo_base=-2)
        torch.cuda.synchronize()
        assert n.chain_sync_error(sync) == 0
        for ta, tb in zip(st, ref):
            assert torch.equal(ta, tb), step
        assert torch.equal(tail, tail_ref)


def test_chain_rejects_what_it_cannot_run():
    n = _n()
    hid, inter, H, D = 512, 1024, 4, 128
    wf = torch.zeros(16, dtype=torch.float16, device=DEV)
    x = torch.zeros((17, hid), dtype=torch.float32, device=DEV)
    sync = n.chain_sync_state(DEV)
    with pytest.raises(RuntimeError):         # more than 16 rows
        n.gemm_chain(wf, wf, wf, H * D, x, 17, hid, wf, wf, 1e-5, inter, wf, wf, wf, sync)
    with pytest.raises(RuntimeError):         # tile widths without an instantiation: the caller falls back to separate launches
        n.gemm_chain(wf, wf, wf, H * D, x, 4, hid, wf, wf, 1e-5, inter, wf, wf, wf, sync)
