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


@pytest.mark.parametrize("T,K,N,nout", [(12, 4096, 512, 4), (1, 4096, 4096, 0), (40, 1024, 256, 37), (12, 11008, 4096, 460), (16, 512, 64, 512)])
def test_fused_outlier_correction_matches_the_oracle(T, K, N, nout):
    """pc_gemm_skinny_a8c (correction inside the projection launch) against the oracle, and against pc_outlier_corr + pc_gemm_skinny_a8."""
    n = _n()
    rng = np.random.default_rng(T + K + nout)
    x = np.clip(rng.standard_normal((T, K)).astype(np.float32) * 1.5, -5.9, 5.9)
    cols = rng.permutation(K)[:nout]
    x[rng.integers(0, T, size=nout), cols] = rng.choice([7.0, -9.5, 30.0, 6.0], size=nout)
    x = x.astype(np.float16).astype(np.float32)
    w = (0.03 * rng.standard_normal((N, K))).astype(np.float32)
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8 = n.to_weight_frags_i8(q)
    hi, _ = n.to_act_frags(torch.from_numpy(x).to(DEV))
    codes = torch.empty_like(hi)
    zero = torch.zeros_like(hi)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros((2, 16384), dtype=torch.uint8, device=DEV)
    n.quant_act_i8(hi, True, T, K, codes, xs, flags[0], flags[1])
    qt = q.t().contiguous()
    y = torch.full((T, N), float("nan"), dtype=torch.float32, device=DEV)
    n.gemm_skinny_a8c(wf8, sc, codes, zero, xs, flags[0], hi, qt, T, N, K, n.EPI_STORE, y=y, ldy=N)
    corr = torch.zeros((T, N), dtype=torch.float32, device=DEV)
    has = torch.zeros(1, dtype=torch.int32, device=DEV)
    y2 = torch.full((T, N), float("nan"), dtype=torch.float32, device=DEV)
    n.outlier_corr(flags[0], K, hi, codes, True, xs, qt, sc, None, T, N, corr, has)
    n.gemm_skinny_a8(wf8, sc, codes, zero, xs, corr, has, T, N, K, n.EPI_STORE, y=y2, ldy=N)
    torch.cuda.synchronize()
    qo, so = io.quantize_rows_int8(w)
    ref = lo.linear(x, qo, so)
    got, got2 = y.cpu().numpy(), y2.cpu().numpy()
    assert np.isfinite(got).all()
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(got - ref).max() < 2e-5 * scale, np.abs(got - ref).max()
    # the two-launch form agrees up to the fp32 summation order -- and up to one fp16 ulp of a dequantised weight where
    # CB * SCB / 127 is an exact tie (about one weight in 4096; times an outlier value of 30 that is 5e-4)
    assert np.abs(got - got2).max() < 2e-4 * scale
    if nout == 0:
        assert np.array_equal(got, got2)
    # residual add on top, canary column untouched
    base = torch.from_numpy(rng.standard_normal((T, N + 4)).astype(np.float32)).to(DEV)
    y3 = base.clone()
    n.gemm_skinny_a8c(wf8, sc, codes, zero, xs, flags[0], hi, qt, T, N, K, n.EPI_ADD, y=y3, ldy=N + 4)
    torch.cuda.synchronize()
    assert np.abs(y3.cpu().numpy()[:, :N] - (base.cpu().numpy()[:, :N] + ref)).max() < 3e-5 * scale
    assert torch.equal(y3[:, N:], base[:, N:])


def test_fused_outlier_correction_silu_and_qkv_rope_equal_the_two_launch_forms():
    n = _n()
    rng = np.random.default_rng(5)
    T, K, inter = 12, 1024, 704
    x = np.clip(rng.standard_normal((T, K)).astype(np.float32) * 1.5, -5.9, 5.9)
    x[2, 100] = 8.0; x[7, 900] = -11.0; x[0, 3] = 6.5
    x = x.astype(np.float16).astype(np.float32)
    hi, _ = n.to_act_frags(torch.from_numpy(x).to(DEV))
    codes = torch.empty_like(hi); zero = torch.zeros_like(hi)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros((2, 16384), dtype=torch.uint8, device=DEV)
    n.quant_act_i8(hi, True, T, K, codes, xs, flags[0], flags[1])
    has = torch.zeros(1, dtype=torch.int32, device=DEV)
    # gate|up + SiLU
    w = (0.05 * rng.standard_normal((2 * inter, K))).astype(np.float32)
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8, qt = n.to_weight_frags_i8(q), q.t().contiguous()
    shape = (1, inter // 32, 64, 8)
    oh_a, ol_a, oh_b, ol_b = (torch.zeros(shape, dtype=torch.float16, device=DEV) for _ in range(4))
    corr = torch.zeros((T, 2 * inter), dtype=torch.float32, device=DEV)
    n.outlier_corr(flags[0], K, hi, codes, True, xs, qt, sc, None, T, 2 * inter, corr, has)
    n.gemm_skinny_a8(wf8, sc, codes, zero, xs, corr, has, T, 2 * inter, K, n.EPI_SILU, of_hi=oh_a, of_lo=ol_a)
    n.gemm_skinny_a8c(wf8, sc, codes, zero, xs, flags[0], hi, qt, T, 2 * inter, K, n.EPI_SILU, of_hi=oh_b, of_lo=ol_b)
    torch.cuda.synchronize()
    a = oh_a.float() + ol_a.float(); b = oh_b.float() + ol_b.float()
    assert int(has[0]) == 1 and float((a - b).abs().max()) < 2e-5 * max(1.0, float(a.abs().max()))
    # q|k|v + RoPE + append (rotary row permutation)
    B, H, Hkv, D, q_len, past = 1, 4, 4, 128, T, 9
    W = (H + 2 * Hkv) * D
    wq = (0.05 * rng.standard_normal((W, K))).astype(np.float32)
    perm = n.qkv_rope_row_perm(H + 2 * Hkv, D).to(DEV)
    qq, scq = n.quantize_rows_int8(torch.from_numpy(wq).to(DEV))
    wf8p, scp, qtt = n.to_weight_frags_i8(qq[perm].contiguous()), scq[perm].contiguous(), qq.t().contiguous()
    perm32 = perm.to(torch.int32)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    pos = torch.from_numpy(rng.integers(0, 2000, size=T).astype(np.int32)).to(DEV)
    inv = torch.from_numpy((1.0 / (10000.0 ** (np.arange(0, D, 2, dtype=np.float64) / D))).astype(np.float32)).to(DEV)
    n.rope_table(pos, inv, cs, T, D)
    cap = past + q_len + 2
    outs = []
    for fused in (False, True):
        arena = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
        qh = torch.zeros((T, H * D), dtype=torch.float16, device=DEV); ql = torch.zeros_like(qh)
        if fused:
            n.gemm_qkv_rope_a8c(wf8p, scp, codes, zero, xs, flags[0], hi, qtt, perm32, T, K, cs, qh, ql, H * D, arena[:, 0], arena[:, 1],
                                2 * Hkv * cap * D, cap * D, B, H, Hkv, D, q_len, past, cap)
        else:
            corrq = torch.zeros((T, W), dtype=torch.float32, device=DEV)
            n.outlier_corr(flags[0], K, hi, codes, True, xs, qtt, scq, perm32, T, W, corrq, has)
            n.gemm_qkv_rope_a8(wf8p, scp, codes, zero, xs, corrq, has, T, K, cs, qh, ql, H * D, arena[:, 0], arena[:, 1],
                               2 * Hkv * cap * D, cap * D, B, H, Hkv, D, q_len, past, cap)
        torch.cuda.synchronize()
        outs.append((qh.float() + ql.float(), arena.float()))
    for a, b in zip(outs[0], outs[1]):
        assert float((a - b).abs().max()) < 2e-3 * max(1.0, float(a.abs().max()))      # (the arena holds fp16: one ulp)
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 2e-5 * max(1.0, float(outs[0][0].abs().max()))


# ---------------------------------------------------------------------------------------------------------------------------
# the codes as the int8 MFMA's operand image (pc_quant_act_i8 / pc_rmsnorm_quant_i8 codes8 -> pc_gemm x_codes8)
# ---------------------------------------------------------------------------------------------------------------------------

def _image_to_rows(img, T, K):
    """[mt][K/64][64 lanes = g*16 + m][16 bytes = k-step 2s (8), 2s+1 (8)] -> codes [T][K]: k = 64 P + 32 half + 8 g + e."""
    mt = img.shape[0]
    v = img.view(mt, K // 64, 4, 16, 2, 8)                    # P, g, m, half, e
    return v.permute(0, 3, 1, 4, 2, 5).reshape(mt * 16, K)[:T]


@pytest.mark.parametrize("T,K", [(12, 4096), (1, 64), (37, 11008), (64, 512), (17, 5120)])
def test_quantisers_write_the_int8_operand_image_of_the_same_codes(T, K):
    n = _n()
    x = _acts(T, K, seed=3 * T + K, outliers=[(0, 3, 9.0), (T - 1, K - 1, -6.5)])
    mt = (T + 15) // 16
    hi, _ = n.to_act_frags(torch.from_numpy(x).to(DEV).half().float())
    codes = torch.empty_like(hi)
    img = torch.full((mt, K // 64, 64, 16), 99, dtype=torch.int8, device=DEV)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros(K, dtype=torch.uint8, device=DEV)
    n.quant_act_i8(hi, True, T, K, codes, xs, flags, None, codes8=img)
    torch.cuda.synchronize()
    rows = n.from_act_frags(codes, T)
    assert torch.equal(_image_to_rows(img, T, K).to(torch.float16), rows)
    # ... and from the fused RMSNorm + quantiser
    xf = torch.from_numpy(x).to(DEV)
    gam = torch.from_numpy((1.0 + 0.1 * np.random.default_rng(0).standard_normal(K)).astype(np.float16)).to(DEV)
    xh = torch.empty_like(hi); codes2 = torch.empty_like(hi)
    img2 = torch.full_like(img, 99)
    n.rmsnorm_quant_i8(xf, gam, 1e-5, T, K, xh, codes2, xs, flags, None, codes8=img2)
    torch.cuda.synchronize()
    assert torch.equal(_image_to_rows(img2, T, K).to(torch.float16), n.from_act_frags(codes2, T))
    with pytest.raises(RuntimeError, match="operand image"):
        n.quant_act_i8(hi[:, :1].contiguous(), True, T, 32, codes, xs, flags, None, codes8=img)


@pytest.mark.parametrize("T,K,N,epi", [(12, 4096, 512, 0), (12, 11008, 4096, 1), (1, 4096, 4096, 1), (16, 4096, 2 * 1024, 2), (29, 1024, 256, 0),
                                       (50, 4096, 12288, 0), (7, 192, 64, 1), (12, 5120, 5120, 1)])
def test_a8_projection_reads_the_image_and_agrees_with_the_code_plane_bit_for_bit(T, K, N, epi):
    """int32 sums on the int8 MFMA either way: operands packed from the fp16 code plane in the kernel, or loaded from the image."""
    n = _n()
    rng = np.random.default_rng(T + K + N)
    x = _acts(T, K, seed=T + N)
    x = np.clip(x, -5.9, 5.9)
    w = (0.03 * rng.standard_normal((N, K))).astype(np.float32)
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8 = n.to_weight_frags_i8(q)
    mt = (T + 15) // 16
    hi, _ = n.to_act_frags(torch.from_numpy(x).to(DEV).half().float())
    codes = torch.empty_like(hi)
    img = torch.empty((mt, K // 64, 64, 16), dtype=torch.int8, device=DEV)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros(max(K, 16384), dtype=torch.uint8, device=DEV)
    n.quant_act_i8(hi, True, T, K, codes, xs, flags, None, codes8=img)
    corr = torch.zeros((T, N), dtype=torch.float32, device=DEV)
    has = torch.zeros(1, dtype=torch.int32, device=DEV)
    zero = torch.zeros_like(hi)
    out = []
    for c8 in (None, img):
        if epi == 2:
            oh = torch.zeros((mt, N // 2 // 32, 64, 8), dtype=torch.float16, device=DEV); ol = torch.zeros_like(oh)
            n.gemm_skinny_a8(wf8, sc, codes, zero, xs, corr, has, T, N, K, n.EPI_SILU, of_hi=oh, of_lo=ol, codes8=c8)
            out.append(torch.stack([oh, ol]))
        else:
            y = torch.full((T, N), 0.25, dtype=torch.float32, device=DEV)
            n.gemm_skinny_a8(wf8, sc, codes, zero, xs, corr, has, T, N, K, epi, y=y, ldy=N, codes8=c8)
            out.append(y)
    torch.cuda.synchronize()
    assert torch.equal(out[0].view(torch.int32) if out[0].dtype == torch.float32 else out[0].view(torch.int16),
                       out[1].view(torch.int32) if out[1].dtype == torch.float32 else out[1].view(torch.int16))
    if epi != 2:
        # exact integer sums: y = (sum_k cw cx) * w_scale * x_scale (+ 0.25 for the residual add), one fp32 rounding per factor
        cx = n.from_act_frags(codes, T).double().cpu().numpy()
        isum = cx @ q.double().cpu().numpy().T
        ref = isum.astype(np.float32) * (xs.cpu().numpy()[:, None] * sc.cpu().numpy()[None, :]) + (0.25 if epi == 1 else 0.0)
        got = out[1].cpu().numpy()
        assert np.abs(got - ref).max() <= 3e-6 * max(1.0, np.abs(ref).max())
