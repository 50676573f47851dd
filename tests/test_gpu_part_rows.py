"""pc_gemm_part_rows: o_proj + residual of a 2..16-row cached step on the attention's split-KV partials (pc_attn defer_merge) -- K cut
across workgroups, the reduction inside the launch, every lane merging its own operand fragments.  Against the three-launch form
(pc_attn + merge launch + pc_gemm): the merged operands are the same bits, the fp32 sums differ in their order only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _n():
    from promptcache_amd import _native
    _native.load()
    if not _native.has("pc_gemm_part_rows"):
        pytest.skip("pc_gemm_part_rows exists in -DPC_DEV_SWEEPS builds only (measured slower than the merge launch + o_proj: csrc/pc_dev.h)")
    return _native


@pytest.mark.parametrize("S,H,D,q_len,ksl,live", [(1725, 32, 128, 12, 8, None), (1725, 32, 128, 16, 8, 12), (1725, 32, 128, 12, 4, None),
                                                  (1725, 32, 128, 12, 2, None), (900, 40, 128, 7, 8, None), (900, 40, 128, 16, 4, 9),
                                                  (3000, 40, 128, 2, 2, None), (600, 8, 64, 5, 4, None)])
def test_o_proj_on_the_partials_of_a_few_rows(S, H, D, q_len, ksl, live):
    n = _n()
    rng = np.random.default_rng(S + H + q_len + ksl)
    Hkv, K, N = H, H * D, (H * D if H * D >= 2048 else 512)
    q = torch.from_numpy(rng.standard_normal((q_len, K)).astype(np.float16)).to(DEV)
    ql = torch.from_numpy((rng.standard_normal((q_len, K)) * 1e-3).astype(np.float16)).to(DEV)
    cap = S + q_len + 8
    kv = torch.from_numpy(rng.standard_normal((1, 2, Hkv, cap, D)).astype(np.float16)).to(DEV)
    w = torch.from_numpy((0.03 * rng.standard_normal((N, K))).astype(np.float16)).to(DEV)
    wf = n.to_weight_frags(w)
    ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q_len, S + q_len), 4) // 4, dtype=torch.float32, device=DEV)
    args = (q, q_len * K, K, kv[:, 0], kv[:, 1], 2 * Hkv * cap * D, cap * D, None, 0, 0, 1, H, Hkv, D, q_len, S, 1.0 / np.sqrt(D), ws)
    ah, al = (torch.zeros((1, K // 32, 64, 8), dtype=torch.float16, device=DEV) for _ in range(2))
    assert n.attn_fwd(*args, out_frag=(ah, al), q_lo=ql) == 1
    rows = live or q_len
    rows_dev = None if live is None else torch.tensor([live], dtype=torch.int32, device=DEV)
    base = torch.from_numpy(rng.standard_normal((q_len, N)).astype(np.float32)).to(DEV)
    y_a = base.clone()
    n.gemm_skinny(wf, ah, al, q_len, N, K, n.EPI_ADD, y=y_a, ldy=N, rows_dev=rows_dev)
    ah2, al2 = torch.zeros_like(ah), torch.zeros_like(al)
    ns = n.attn_fwd(*args, out_frag=(ah2, al2), q_lo=ql, defer_merge=True)
    if ns == 1:
        pytest.skip("this launch shape does not split the keys: nothing is deferred")
    assert 2 <= ns <= 8 and float(ah2.abs().max()) == 0.0
    sc = torch.empty(n.gemm_skinny_ks_scratch_bytes(N, 8) // 4, dtype=torch.float32, device=DEV)
    ctr = torch.zeros(N // 16, dtype=torch.int32, device=DEV)
    y_b = base.clone()
    for _ in range(3):                                                   # (the counters return to zero: launches repeat)
        y_b.copy_(base)
        n.gemm_part_rows(wf, ws, ws[H * ns * q_len * D:], ns, H, D, N, q_len, y_b, N, ksl, sc, ctr, rows_dev=rows_dev)
    torch.cuda.synchronize()
    assert int(ctr.abs().sum()) == 0
    merged = (n.from_act_frags(ah, q_len).float() + n.from_act_frags(al, q_len).float()).cpu().numpy().astype(np.float64)
    ref = base.cpu().numpy() + merged @ w.float().cpu().numpy().astype(np.float64).T
    got_a, got_b = y_a.cpu().numpy(), y_b.cpu().numpy()
    assert np.abs(got_b[:rows] - ref[:rows]).max() < 1e-3, np.abs(got_b[:rows] - ref[:rows]).max()
    assert np.abs(got_b[:rows] - got_a[:rows]).max() < 2e-4                 # the same operands, another fp32 order
    assert np.array_equal(got_b[rows:], base.cpu().numpy()[rows:])          # rows behind the live ones are not touched
    # deterministic: bit-equal from launch to launch
    y_c = base.clone()
    n.gemm_part_rows(wf, ws, ws[H * ns * q_len * D:], ns, H, D, N, q_len, y_c, N, ksl, sc, ctr, rows_dev=rows_dev)
    assert torch.equal(y_b, y_c)


def test_argument_errors():
    n = _n()
    t = torch.zeros(4096, dtype=torch.float32, device=DEV)
    h = torch.zeros(4096, dtype=torch.float16, device=DEV)
    c = torch.zeros(64, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="kslices"):
        n.gemm_part_rows(h, t, t, 4, 8, 64, 512, 4, t, 512, 3, t, c)
    with pytest.raises(RuntimeError, match="1..16 rows"):
        n.gemm_part_rows(h, t, t, 4, 8, 64, 512, 17, t, 512, 4, t, c)
    with pytest.raises(RuntimeError, match="scratch"):
        n.gemm_part_rows(h, t, t, 4, 8, 64, 512, 4, t, 512, 8, t[:16], c)
