"""pc_gemm_dense (csrc/pc_gemm_dense.hip): the many-row MFMA projection with split-precision activations and fused
epilogues, against float64 numpy on the same fp16 inputs (the oracle's nn.Linear is ``h @ w.T`` in fp32,
oracle/llama_oracle.py LlamaOracle.forward; float64 here so the comparison sees only the kernel's rounding)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _n():
    from promptcache_amd import _native
    _native.load()
    return _native


def _split(x32):
    hi = x32.astype(np.float16)
    lo = (x32 - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def _inputs(M, N, K, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    x = (scale * rng.standard_normal((M, K))).astype(np.float32)
    w = (0.05 * rng.standard_normal((N, K))).astype(np.float16)
    return x, w


# (M, N, K): ragged M, N not a multiple of the 256/128 panels, K with a partial last 64-step, both tile widths
SHAPES = [(128, 256, 64), (1, 4, 8), (130, 260, 72), (300, 4096, 4096), (1000, 12288, 4096), (77, 512, 344),
          (257, 4672, 4544), (640, 4096, 11008), (2050, 1024, 128)]


@pytest.mark.parametrize("two", [True, False])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_dense_store_and_add(M, N, K, two):
    n = _n()
    x, w = _inputs(M, N, K, seed=M + N + K)
    hi, lo = _split(x)
    xs = hi.astype(np.float64) + (lo.astype(np.float64) if two else 0.0)
    ref = xs @ w.astype(np.float64).T
    th, tl, tw = torch.from_numpy(hi).to(DEV), torch.from_numpy(lo).to(DEV), torch.from_numpy(w).to(DEV)
    y = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    n.gemm_dense(th, tl if two else None, tw, M, N, K, n.EPI_STORE, y=y)
    torch.cuda.synchronize()
    got = y.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    tol = 2e-6 * np.sqrt(K) * np.abs(xs).max() * 0.05 * 8 + 1e-6
    assert np.abs(got - ref).max() < tol, (np.abs(got - ref).max(), tol)
    # residual epilogue: y += acc on top of existing values, and rows past M / cols past N untouched (canary columns)
    base = torch.from_numpy(np.random.default_rng(1).standard_normal((M, N + 4)).astype(np.float32)).to(DEV)
    y2 = base.clone()
    n.gemm_dense(th, tl if two else None, tw, M, N, K, n.EPI_ADD, y=y2, ldy=N + 4)
    torch.cuda.synchronize()
    got2 = y2.cpu().numpy().astype(np.float64)
    assert np.abs(got2[:, :N] - (base.cpu().numpy()[:, :N].astype(np.float64) + ref)).max() < tol
    assert np.array_equal(got2[:, N:], base.cpu().numpy()[:, N:].astype(np.float64))


# few-row launches of the N = hidden projections take the split-K form when a workspace is supplied (pc_gemm_dense_ws);
# (1000, 12288, 4096) and (77, 512, 8) do not split (large grid / too few K-steps) and must equal pc_gemm_dense bit for bit
@pytest.mark.parametrize("two", [True, False])
@pytest.mark.parametrize("M,N,K", [(300, 4096, 4096), (512, 4096, 11008), (130, 260, 1096), (800, 5120, 13824),
                                   (1000, 12288, 4096), (77, 512, 8), (1, 4096, 4096)])
def test_dense_split_k_with_workspace(M, N, K, two):
    n = _n()
    x, w = _inputs(M, N, K, seed=7 * M + N + K)
    hi, lo = _split(x)
    xs = hi.astype(np.float64) + (lo.astype(np.float64) if two else 0.0)
    ref = xs @ w.astype(np.float64).T
    th, tl, tw = torch.from_numpy(hi).to(DEV), torch.from_numpy(lo).to(DEV), torch.from_numpy(w).to(DEV)
    ws = torch.full((34 << 20,), 0xFF, dtype=torch.uint8, device=DEV)          # NaN-filled scratch
    tol = 2e-6 * np.sqrt(K) * np.abs(xs).max() * 0.05 * 8 + 1e-6
    y = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    n.gemm_dense(th, tl if two else None, tw, M, N, K, n.EPI_STORE, y=y, workspace=ws)
    base = torch.from_numpy(np.random.default_rng(1).standard_normal((M, N + 4)).astype(np.float32)).to(DEV)
    y2 = base.clone()
    n.gemm_dense(th, tl if two else None, tw, M, N, K, n.EPI_ADD, y=y2, ldy=N + 4, workspace=ws)
    y3 = torch.empty_like(y)
    n.gemm_dense(th, tl if two else None, tw, M, N, K, n.EPI_STORE, y=y3)
    n.gemm_dense(th, tl if two else None, tw, M, N, K, n.EPI_STORE, y=y, workspace=ws[:1024])   # no room: the plain form
    torch.cuda.synchronize()
    assert torch.equal(y, y3)
    n.gemm_dense(th, tl if two else None, tw, M, N, K, n.EPI_STORE, y=y, workspace=ws)
    torch.cuda.synchronize()
    got = y.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < tol, (np.abs(got - ref).max(), tol)
    got2 = y2.cpu().numpy().astype(np.float64)
    assert np.abs(got2[:, :N] - (base.cpu().numpy()[:, :N].astype(np.float64) + ref)).max() < tol
    assert np.array_equal(got2[:, N:], base.cpu().numpy()[:, N:].astype(np.float64))
    if (M, N, K) in ((1000, 12288, 4096), (77, 512, 8)):
        assert torch.equal(y, y3)
    # deterministic: the slabs are added in slice order
    y4 = torch.empty_like(y)
    n.gemm_dense(th, tl if two else None, tw, M, N, K, n.EPI_STORE, y=y4, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(y, y4)


@pytest.mark.parametrize("M,inter,K", [(128, 128, 64), (300, 11008, 4096), (45, 344, 128), (1030, 1376, 512), (513, 13824, 5120)])
def test_dense_silu_epilogue(M, inter, K):
    n = _n()
    x, w = _inputs(M, 2 * inter, K, seed=inter)
    hi, lo = _split(x)
    xs = hi.astype(np.float64) + lo.astype(np.float64)
    acc = xs @ w.astype(np.float64).T
    g, u = acc[:, :inter], acc[:, inter:]
    ref = g / (1.0 + np.exp(-g)) * u
    th, tl, tw = torch.from_numpy(hi).to(DEV), torch.from_numpy(lo).to(DEV), torch.from_numpy(w).to(DEV)
    oh = torch.full((M, inter), float("nan"), dtype=torch.float16, device=DEV)
    ol = torch.full((M, inter), float("nan"), dtype=torch.float16, device=DEV)
    n.gemm_dense(th, tl, tw, M, 2 * inter, K, n.EPI_SILU, out_hi=oh, out_lo=ol)
    torch.cuda.synchronize()
    got = oh.float().cpu().numpy().astype(np.float64) + ol.float().cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max()
    assert err < 3e-6 * max(1.0, np.abs(ref).max()) * np.sqrt(K) / 8 + 1e-6, err
    # the hi plane alone is the fp16 rounding of the value
    assert np.abs(oh.float().cpu().numpy() - ref.astype(np.float32).astype(np.float16).astype(np.float32)).max() <= \
        np.abs(ref).max() * 2.0 ** -10


def test_dense_gelu_and_wscale():
    n = _n()
    M, N, K = 200, 1792, 448
    x, w = _inputs(M, N, K, seed=9)
    hi, lo = _split(x)
    xs = hi.astype(np.float64) + lo.astype(np.float64)
    sc = (0.5 + np.random.default_rng(2).random(N)).astype(np.float32)
    acc = (xs @ w.astype(np.float64).T) * sc.astype(np.float64)
    from math import erf
    ref = 0.5 * acc * (1.0 + np.vectorize(erf)(acc * 0.7071067811865476))
    th, tl, tw = torch.from_numpy(hi).to(DEV), torch.from_numpy(lo).to(DEV), torch.from_numpy(w).to(DEV)
    ts = torch.from_numpy(sc).to(DEV)
    oh = torch.empty((M, N), dtype=torch.float16, device=DEV)
    ol = torch.empty((M, N), dtype=torch.float16, device=DEV)
    n.gemm_dense(th, tl, tw, M, N, K, n.EPI_GELU, out_hi=oh, out_lo=ol, wscale=ts)
    y = torch.empty((M, N), dtype=torch.float32, device=DEV)
    n.gemm_dense(th, tl, tw, M, N, K, n.EPI_STORE, y=y, wscale=ts)
    torch.cuda.synchronize()
    got = oh.float().cpu().numpy().astype(np.float64) + ol.float().cpu().numpy().astype(np.float64)
    assert np.abs(got - ref).max() < 5e-6 * max(1.0, np.abs(ref).max())
    assert np.abs(y.cpu().numpy() - acc).max() < 5e-6 * max(1.0, np.abs(acc).max())


def test_dense_strided_views_and_argument_errors():
    """Operands are views into wider buffers (row strides larger than K / N), as the layer stack hands them over."""
    n = _n()
    M, N, K = 150, 384, 192
    x, w = _inputs(M, N, K, seed=4)
    hi, lo = _split(x)
    buf = torch.zeros((2, M, K + 64), dtype=torch.float16, device=DEV)
    buf[0, :, :K] = torch.from_numpy(hi).to(DEV)
    buf[1, :, :K] = torch.from_numpy(lo).to(DEV)
    wb = torch.zeros((N, K + 8), dtype=torch.float16, device=DEV)
    wb[:, :K] = torch.from_numpy(w).to(DEV)
    y = torch.empty((M, N), dtype=torch.float32, device=DEV)
    n.gemm_dense(buf[0, :, :K], buf[1, :, :K], wb[:, :K], M, N, K, n.EPI_STORE, y=y)
    torch.cuda.synchronize()
    ref = (hi.astype(np.float64) + lo.astype(np.float64)) @ w.astype(np.float64).T
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-4
    lib = n.load()
    assert lib.pc_gemm_dense(None, None, 8, None, 8, None, 1, 4, 8, 0, None, 4, None, None, 0, None) < 0
    assert lib.pc_gemm_dense(buf.data_ptr(), None, K, wb.data_ptr(), K, None, M, N, 12, 0, y.data_ptr(), N, None, None, 0, None) < 0
    assert b"K%8" in lib.pc_last_error_string()


# ---------------------------------------------------------------------------------------------------------------------------
# round 5: the residual activation plane on the int8 MFMA (pc_quant_rows_i8 + pc_gemm_dense_lo8)
# ---------------------------------------------------------------------------------------------------------------------------

def _q8_rows(a):
    """numpy restatement of the row-wise absmax quantiser: codes = rint(a * (127 / max|a|)) (fp32 arithmetic), scale = max / 127."""
    a = a.astype(np.float32)
    amax = np.abs(a).max(axis=1)
    inv = np.where(amax > 0, np.float32(127.0) / amax, np.float32(0.0)).astype(np.float32)
    codes = np.clip(np.rint(a * inv[:, None]), -127, 127).astype(np.int8)
    return codes, (amax / np.float32(127.0)).astype(np.float32)


def _has_lo8():
    from promptcache_amd import _native
    return _native.has("pc_gemm_dense_lo8")


# (round 6: the int8 residual plane left the product library -- +2 % encode throughput for a second weight image; it is built with
# PC_BUILD_FLAGS=-DPC_DEV_SWEEPS only, csrc/pc_dev.h)
_lo8_only = pytest.mark.skipif(not _has_lo8(), reason="pc_gemm_dense_lo8 exists only in -DPC_DEV_SWEEPS builds (csrc/pc_dev.h)")


@_lo8_only
@pytest.mark.parametrize("M,K", [(1, 64), (130, 4096), (300, 11008), (77, 13824), (5, 5120)])
def test_quant_rows_i8_matches_numpy(M, K):
    n = _n()
    rng = np.random.default_rng(M + K)
    lo = (rng.standard_normal((M, K)) * 2.0 ** -12 * rng.uniform(0.1, 30.0, size=(M, 1))).astype(np.float16)
    lo[M // 2] = 0                                                        # an all-zero row: codes 0, scale 0
    t = torch.from_numpy(lo).to(DEV)
    codes = torch.full((M, K), 99, dtype=torch.int8, device=DEV)
    sc = torch.full((M,), float("nan"), dtype=torch.float32, device=DEV)
    n.quant_rows_i8(t, M, K, codes, sc)
    torch.cuda.synchronize()
    ref_c, ref_s = _q8_rows(lo)
    got_c, got_s = codes.cpu().numpy(), sc.cpu().numpy()
    assert np.array_equal(got_s, ref_s)
    # (the device multiplies by 127 / max formed in fp32 division, numpy the same: identical codes up to ties broken by a last-bit
    # difference of the quotient -- none expected, one code step tolerated)
    assert np.abs(got_c.astype(np.int32) - ref_c.astype(np.int32)).max() <= 1
    assert (got_c != ref_c).mean() < 1e-4


LO8_SHAPES = [(128, 256, 64), (130, 260, 128), (300, 4096, 4096), (1000, 12288, 4096), (640, 4096, 11008), (257, 5120, 13824),
              (2050, 1024, 128), (1, 4096, 4096)]


@_lo8_only
@pytest.mark.parametrize("M,N,K", LO8_SHAPES)
def test_dense_lo8_store_add_and_split_k(M, N, K):
    """(x_hi + x_lo) @ W^T with the residual plane as int8 codes against the int8 weight image: (i) EXACT against the float64
    evaluation of what the kernel is specified to compute -- x_hi . W^T + (codes_x . codes_w^T) * scale_x * scale_w -- up to fp32
    accumulation; (ii) within the residual plane's quantisation of the true split-precision product (x_hi + x_lo) . W^T: an
    error of 2^-8 of the row's largest residual per element, i.e. far below one fp16 ulp of the activations."""
    n = _n()
    x, w = _inputs(M, N, K, seed=3 * M + N + K)
    hi, lo = _split(x)
    cx, sx = _q8_rows(lo)
    tw = torch.from_numpy(w).to(DEV)
    w8, w8s = n.quantize_rows_int8(tw)
    cw, sw = w8.cpu().numpy(), w8s.cpu().numpy()
    spec = hi.astype(np.float64) @ w.astype(np.float64).T + \
        (cx.astype(np.float64) @ cw.astype(np.float64).T) * sx.astype(np.float64)[:, None] * sw.astype(np.float64)[None, :]
    true = (hi.astype(np.float64) + lo.astype(np.float64)) @ w.astype(np.float64).T
    th, tl = torch.from_numpy(hi).to(DEV), torch.from_numpy(lo).to(DEV)
    codes = torch.empty((M, K), dtype=torch.int8, device=DEV)
    sc = torch.empty(M, dtype=torch.float32, device=DEV)
    n.quant_rows_i8(tl, M, K, codes, sc)
    assert np.array_equal(sc.cpu().numpy(), sx)
    cx_dev = codes.cpu().numpy()
    if not np.array_equal(cx_dev, cx):                                  # (a tie broken the other way: follow the device's codes)
        spec = hi.astype(np.float64) @ w.astype(np.float64).T + \
            (cx_dev.astype(np.float64) @ cw.astype(np.float64).T) * sx.astype(np.float64)[:, None] * sw.astype(np.float64)[None, :]
    tol = 2e-6 * np.sqrt(K) * np.abs(hi).max() * 0.05 * 8 + 1e-6          # fp32 accumulation, as for pc_gemm_dense
    ws = torch.full((34 << 20,), 0xFF, dtype=torch.uint8, device=DEV)
    for workspace in (None, ws):
        y = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        n.gemm_dense_lo8(th, codes, sc, tw, w8, w8s, M, N, K, n.EPI_STORE, y=y, workspace=workspace)
        base = torch.from_numpy(np.random.default_rng(1).standard_normal((M, N)).astype(np.float32)).to(DEV)
        y2 = base.clone()
        n.gemm_dense_lo8(th, codes, sc, tw, w8, w8s, M, N, K, n.EPI_ADD, y=y2, workspace=workspace)
        torch.cuda.synchronize()
        got = y.cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all()
        assert np.abs(got - spec).max() < tol, (np.abs(got - spec).max(), tol)
        assert np.abs(y2.cpu().numpy().astype(np.float64) - (base.cpu().numpy().astype(np.float64) + spec)).max() < tol
        # the residual plane's own error: |lo - codes * scale| <= scale / 2 per element and |w - cw * sw| <= sw / 2, summed over K at
        # random sign; bound it by 4 sigma of that sum, and compare with what dropping the plane would cost
        q_err = np.abs(got - true).max()
        drop_err = np.abs(hi.astype(np.float64) @ w.astype(np.float64).T - true).max()
        bound = 4.0 * np.sqrt(K) * (sx.max() / 2 * np.abs(w).max() + np.abs(lo).max() * sw.max() / 2) + tol
        assert q_err < bound, (q_err, bound)
        assert q_err < 0.05 * drop_err + tol, (q_err, drop_err)
        # ... and it agrees with the fp16 residual plane's launch far inside an fp16 ulp of the outputs
        y3 = torch.empty((M, N), dtype=torch.float32, device=DEV)
        n.gemm_dense(th, tl, tw, M, N, K, n.EPI_STORE, y=y3)
        assert float((y3 - y).abs().max()) < 2.0 ** -13 * max(float(y3.abs().max()), 1.0)


@_lo8_only
@pytest.mark.parametrize("M,inter,K", [(300, 11008, 4096), (513, 13824, 5120), (130, 192, 128)])
def test_dense_lo8_silu_epilogue(M, inter, K):
    n = _n()
    x, w = _inputs(M, 2 * inter, K, seed=M + inter)
    hi, lo = _split(x)
    th, tl, tw = torch.from_numpy(hi).to(DEV), torch.from_numpy(lo).to(DEV), torch.from_numpy(w).to(DEV)
    w8, w8s = n.quantize_rows_int8(tw)
    codes = torch.empty((M, K), dtype=torch.int8, device=DEV)
    sc = torch.empty(M, dtype=torch.float32, device=DEV)
    n.quant_rows_i8(tl, M, K, codes, sc)
    oh = torch.empty((M, inter), dtype=torch.float16, device=DEV)
    ol = torch.empty_like(oh)
    n.gemm_dense_lo8(th, codes, sc, tw, w8, w8s, M, 2 * inter, K, n.EPI_SILU, out_hi=oh, out_lo=ol)
    rh, rl = torch.empty_like(oh), torch.empty_like(oh)
    n.gemm_dense(th, tl, tw, M, 2 * inter, K, n.EPI_SILU, out_hi=rh, out_lo=rl)
    torch.cuda.synchronize()
    got = oh.float() + ol.float()
    ref = rh.float() + rl.float()
    xs = hi.astype(np.float64) + lo.astype(np.float64)
    g = xs @ w[:inter].astype(np.float64).T
    u = xs @ w[inter:].astype(np.float64).T
    exact = torch.from_numpy((g / (1.0 + np.exp(-g)) * u)).to(DEV)
    scale = float(exact.abs().max())
    assert float((got.double() - exact).abs().max()) < 2.0 ** -13 * scale
    assert float((got - ref).abs().max()) < 2.0 ** -13 * scale
