"""LLM.int8 kernels (csrc/pc_int8.hip + the *_a8 projection entry points) against oracle/llmint8_oracle.py -- the published
algorithm behind the reference's ``load_in_8bit=True`` (demo.py:27-29).  Integer results are bit-exact."""
import numpy as np
import pytest
import torch

from oracle import int8_oracle as io
from oracle import llmint8_oracle as lo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _n():
    from promptcache_amd import _native
    _native.load()
    return _native


def _acts(T, K, seed, outliers=()):
    rng = np.random.default_rng(seed)
    x = (1.5 * rng.standard_normal((T, K))).astype(np.float32)
    for t, k, v in outliers:
        x[t, k] = v
    return x


@pytest.mark.parametrize("frag", [False, True])
@pytest.mark.parametrize("T,K", [(12, 4096), (1, 64), (37, 11008), (64, 512), (5, 160)])
def test_activation_quantiser_is_bit_exact_with_the_oracle(T, K, frag):
    n = _n()
    x = _acts(T, K, seed=T + K, outliers=[(0, 3, 9.0), (T - 1, K - 1, -6.0), (T // 2, K // 2, 6.0), (0, 17, 5.996)])
    ca, sca, cols, x16 = lo.quantize_activations(x)
    assert {3, K - 1, K // 2} <= set(cols.tolist())
    th = torch.from_numpy(x16).to(DEV)
    if frag:
        src, _ = n.to_act_frags(th.float())                 # hi plane = fp16(x)
    else:
        src = th
    codes = torch.full_like(src, 777.0)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros(K, dtype=torch.uint8, device=DEV)
    nxt = torch.ones(K, dtype=torch.uint8, device=DEV)
    n.quant_act_i8(src, frag, T, K, codes, xs, flags, nxt)
    torch.cuda.synchronize()
    if frag:
        mt = (T + 15) // 16
        got = codes.view(mt, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(mt * 16, K)[:T].float().cpu().numpy()
    else:
        got = codes.float().cpu().numpy()
    want = ca.astype(np.float32).copy()
    # the kernel zeroes outlier ENTRIES; whole outlier columns are cancelled by pc_outlier_corr (see below)
    keep = np.ones(K, bool); keep[cols] = False
    assert np.array_equal(got[:, keep], want[:, keep])
    outl = np.abs(x16.astype(np.float32)) >= 6.0
    assert (got[outl] == 0).all()
    assert np.array_equal(xs.cpu().numpy(), (sca / np.float32(127.0)).astype(np.float32))
    assert np.array_equal(np.flatnonzero(flags.cpu().numpy()), cols)
    assert not nxt.any()                                      # the next slot's flags were cleared


def _linear_skinny(n, x, w, perm=None):
    """LLM.int8 linear through the weight-streaming kernels (fragment planes, T <= 64), EPI_STORE."""
    T, K = x.shape
    N = w.shape[0]
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8 = n.to_weight_frags_i8(q)
    wcodes = q.to(torch.float16)
    hi, _ = n.to_act_frags(torch.from_numpy(x).to(DEV).half().float())
    codes = torch.empty_like(hi)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros(K, dtype=torch.uint8, device=DEV)
    corr = torch.full((T, N), float("nan"), dtype=torch.float32, device=DEV)
    has = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    zero = torch.zeros_like(hi)
    n.quant_act_i8(hi, True, T, K, codes, xs, flags, None)
    n.outlier_corr(flags, K, hi, codes, True, xs, q.t().contiguous(), sc, None, T, N, corr, has)
    y = torch.full((T, N), float("nan"), dtype=torch.float32, device=DEV)
    n.gemm_skinny_a8(wf8, sc, codes, zero, xs, corr, has, T, N, K, n.EPI_STORE, y=y, ldy=N)
    torch.cuda.synchronize()
    return y.cpu().numpy(), int(has[0]), q.cpu().numpy(), sc.cpu().numpy()


def _linear_dense(n, x, w):
    T, K = x.shape
    N = w.shape[0]
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wcodes = q.to(torch.float16)
    x16 = torch.from_numpy(x).to(DEV).half()
    codes = torch.empty_like(x16)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros(K, dtype=torch.uint8, device=DEV)
    corr = torch.full((T, N), float("nan"), dtype=torch.float32, device=DEV)
    has = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    n.quant_act_i8(x16, False, T, K, codes, xs, flags, None)
    n.outlier_corr(flags, K, x16, codes, False, xs, q.t().contiguous(), sc, None, T, N, corr, has)
    y = torch.full((T, N), float("nan"), dtype=torch.float32, device=DEV)
    n.gemm_dense_a8(codes, wcodes, sc, xs, corr, has, T, N, K, n.EPI_STORE, y=y)
    torch.cuda.synchronize()
    return y.cpu().numpy(), int(has[0]), q.cpu().numpy(), sc.cpu().numpy()


@pytest.mark.parametrize("path", ["skinny", "dense"])
@pytest.mark.parametrize("with_outliers", [False, True])
def test_llm_int8_linear_matches_the_oracle(path, with_outliers):
    n = _n()
    T, K, N = (12, 4096, 512) if path == "skinny" else (300, 4096, 768)
    outl = [(0, 5, 8.5), (3, 5, -7.25), (T - 1, 4000, 6.0), (1, 77, 30.0)] if with_outliers else []
    x = _acts(T, K, seed=9, outliers=outl)                    # (column 77: one outlier entry, ordinary entries in the other rows)
    if not with_outliers:
        x = np.clip(x, -5.9, 5.9)
    rng = np.random.default_rng(1)
    w = (0.03 * rng.standard_normal((N, K))).astype(np.float32)
    y, has, q, sc = (_linear_skinny if path == "skinny" else _linear_dense)(n, x, w)
    qo, so = io.quantize_rows_int8(w)
    assert np.array_equal(q, qo) and np.array_equal(sc, so)
    ref = lo.linear(x, qo, so)
    assert has == (1 if with_outliers else 0)
    assert np.isfinite(y).all()
    # fp32 rescaling and the fp32 sums of the outlier part are the only roundings on either side
    assert np.abs(y - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), np.abs(y - ref).max()
    if with_outliers:
        # the decomposition matters: plain vector-wise int8 of the same input is far from it
        y0 = lo.linear(x, qo, so, threshold=0.0)
        assert np.abs(y0 - ref).max() > 50 * np.abs(y - ref).max()


@pytest.mark.parametrize("T,hid", [(12, 4096), (1, 5120), (40, 256), (64, 11008)])
def test_rmsnorm_quant_fused_equals_the_two_launches(T, hid):
    """pc_rmsnorm_quant_i8 == pc_rmsnorm_frag followed by pc_quant_act_i8, bit for bit (hi plane, codes, scales, flags)."""
    n = _n()
    g = torch.Generator().manual_seed(T + hid)
    x = (2.5 * torch.randn((T, hid), generator=g)).to(DEV)
    x[0, 3] = 40.0                                              # an outlier behind the norm as well
    gam = (1.0 + 0.3 * torch.randn(hid, generator=g)).half().to(DEV)
    mt = (T + 15) // 16
    shape = (mt, hid // 32, 64, 8)
    hi_a, lo_a, cd_a = (torch.zeros(shape, dtype=torch.float16, device=DEV) for _ in range(3))
    hi_b, cd_b = (torch.zeros(shape, dtype=torch.float16, device=DEV) for _ in range(2))
    xs_a, xs_b = torch.zeros(T, device=DEV), torch.zeros(T, device=DEV)
    fl_a = torch.zeros((2, 16384), dtype=torch.uint8, device=DEV)
    fl_b = torch.zeros((2, 16384), dtype=torch.uint8, device=DEV)
    fl_a[1].fill_(7); fl_b[1].fill_(7)                          # the "next slot" row must come back cleared
    n.rmsnorm_frag(x.clone(), gam, hi_a, lo_a, T, hid, 1e-5)
    n.quant_act_i8(hi_a, True, T, hid, cd_a, xs_a, fl_a[0], fl_a[1])
    n.rmsnorm_quant_i8(x, gam, 1e-5, T, hid, hi_b, cd_b, xs_b, fl_b[0], fl_b[1])
    torch.cuda.synchronize()
    assert torch.equal(hi_a, hi_b) and torch.equal(cd_a, cd_b) and torch.equal(xs_a, xs_b) and torch.equal(fl_a, fl_b)
    assert int(fl_b[0].sum()) >= 1 and int(fl_b[1].sum()) == 0
