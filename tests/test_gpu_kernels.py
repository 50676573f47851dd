"""GPU parity tests for the HIP kernels, called through the C-ABI, checked against the numpy oracle.

Bit-exact for the byte-moving kernels (gather / slice-store / V append); fp16-attention tolerance for
RoPE and attention (tolerances written at each assert).
"""
import numpy as np
import pytest
import torch

from oracle import llama_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _n():
    from promptcache_amd import _native
    return _native


# ---------------------------------------------------------------------------------------------------
# hardware lane maps the attention kernel depends on
# ---------------------------------------------------------------------------------------------------

def test_probe_mfma_cd_layout_and_lds_transpose_read():
    n = _n()
    om = torch.zeros(256, device=DEV, dtype=torch.float32)
    ot = torch.zeros(512, device=DEV, dtype=torch.float32)
    n.probe_layouts(om, ot)
    torch.cuda.synchronize()
    om = om.cpu().numpy().reshape(64, 4)
    ot = ot.cpu().numpy()
    lane = np.arange(64)
    nn, g = lane & 15, lane >> 4
    # MFMA 16x16 C/D: col = lane&15, row = 4*(lane>>4)+reg ; D[m][n] = (m+1)*(n+17)
    exp = np.stack([(4 * g + r + 1) * (nn + 17) for r in range(4)], axis=1).astype(np.float32)
    np.testing.assert_array_equal(om, exp)
    # ds_read_b64_tr_b16, contiguous addressing: lane i of group G receives lds[64G + 16j + i]
    exp_tr = np.stack([64 * g + 16 * j + nn for j in range(4)], axis=1).astype(np.float32)
    np.testing.assert_array_equal(ot[:256].reshape(64, 4), exp_tr)
    # attention-kernel addressing: lane supplies (row 4G + i/4, cols 4(i%4)..) of a 128-wide tile,
    # receives rows 4G + j at column i
    exp_tr2 = np.stack([(4 * g + j) * 128 + nn for j in range(4)], axis=1).astype(np.float32)
    np.testing.assert_array_equal(ot[256:].reshape(64, 4), exp_tr2)


# ---------------------------------------------------------------------------------------------------
# kv_gather / kv_slice_store
# ---------------------------------------------------------------------------------------------------

def _rand_half(shape, rng):
    # arbitrary bit patterns incl. NaN/Inf encodings: the copy must be a pure byte move
    return torch.from_numpy(rng.integers(0, 65536, size=shape, dtype=np.uint16).view(np.float16))


@pytest.mark.parametrize("L,Hkv,D,lens,max_ctx", [
    (2, 4, 32, [5, 1, 1, 17, 1, 64, 65, 3], 200),
    (3, 2, 128, [275, 1, 1, 1, 84, 1, 174, 256, 1], 900),
    (1, 1, 64, [1], 1),
    (2, 2, 128, [1] * 50 + [130], 256),          # > kMaxSeg descriptors -> several launches
    (2, 3, 128, [0, 7, 0, 9], 16),                # empty segments are skipped
])
def test_kv_gather_bit_exact(L, Hkv, D, lens, max_ctx):
    n = _n()
    rng = np.random.default_rng(0)
    segs = [_rand_half((L, 2, Hkv, ln, D), rng).to(DEV) for ln in lens]
    dst = torch.zeros((L, 2, Hkv, max_ctx, D), dtype=torch.float16, device=DEV)
    canary = _rand_half((L, 2, Hkv, max_ctx, D), rng).to(DEV)
    dst.copy_(canary)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(int).tolist()
    n.kv_gather([s.data_ptr() if s.numel() else 0 for s in segs], lens, offs, dst, L, Hkv, D, max_ctx)
    torch.cuda.synchronize()
    exp = canary.clone()
    for s, o, ln in zip(segs, offs, lens):
        exp[:, :, :, o:o + ln, :] = s
    assert torch.equal(dst.view(torch.int16), exp.view(torch.int16))  # rows past S untouched


def test_kv_gather_matches_oracle_rounding_and_layout():
    """Oracle path: fp32 module KV rounded to fp16 by the staging copy (cache_engine.py:148-149)."""
    n = _n()
    rng = np.random.default_rng(1)
    L, H, D, lens, max_ctx = 2, 4, 32, [9, 1, 30], 64
    segs32 = [[(rng.standard_normal((H, ln, D), dtype=np.float32), rng.standard_normal((H, ln, D), dtype=np.float32))
               for _ in range(L)] for ln in lens]
    staged, S = orc.kv_gather(segs32, max_ctx)
    segs_dev = [torch.from_numpy(np.stack([np.stack([kv[0], kv[1]]) for kv in seg]).astype(np.float16)).to(DEV)
                for seg in segs32]
    dst = torch.zeros((L, 2, H, max_ctx, D), dtype=torch.float16, device=DEV)
    n.kv_gather([s.data_ptr() for s in segs_dev], lens, [0, 9, 10], dst, L, H, D, max_ctx)
    torch.cuda.synchronize()
    assert S == 40
    for i in range(L):
        np.testing.assert_array_equal(dst[i, 0, :, :S].cpu().numpy(), staged[i][0])
        np.testing.assert_array_equal(dst[i, 1, :, :S].cpu().numpy(), staged[i][1])


def test_kv_gather_errors():
    n = _n()
    seg = torch.zeros((1, 2, 1, 8, 32), dtype=torch.float16, device=DEV)
    dst = torch.zeros((1, 2, 1, 8, 32), dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="exceeds max_ctx"):
        n.kv_gather([seg.data_ptr()], [8], [1], dst, 1, 1, 32, 8)
    with pytest.raises(RuntimeError, match="multiple of 8"):
        n.kv_gather([seg.data_ptr()], [8], [0], dst, 1, 1, 36, 8)
    n.kv_gather([], [], [], dst, 1, 1, 32, 8)  # empty table is a no-op


def test_slice_store_then_gather_round_trip():
    """encode arena -> segment stores -> staged buffer reproduces the selected rows (any order)."""
    n = _n()
    rng = np.random.default_rng(2)
    L, Hkv, D, cap = 2, 2, 128, 300
    arena = _rand_half((L, 2, Hkv, cap, D), rng).to(DEV)
    src_off = [0, 40, 41, 200, 299]
    lens = [40, 1, 100, 99, 1]
    stores = [torch.empty((L, 2, Hkv, ln, D), dtype=torch.float16, device=DEV) for ln in lens]
    n.kv_slice_store(arena, cap, src_off, lens, [s.data_ptr() for s in stores], L, Hkv, D)
    for s, o, ln in zip(stores, src_off, lens):
        assert torch.equal(s.view(torch.int16), arena[:, :, :, o:o + ln].contiguous().view(torch.int16))
    order = [3, 0, 4, 2, 1]
    dst = torch.zeros((L, 2, Hkv, 256, D), dtype=torch.float16, device=DEV)
    offs, o = [], 0
    for i in order:
        offs.append(o)
        o += lens[i]
    n.kv_gather([stores[i].data_ptr() for i in order], [lens[i] for i in order], offs, dst, L, Hkv, D, 256)
    torch.cuda.synchronize()
    exp = torch.cat([arena[:, :, :, src_off[i]:src_off[i] + lens[i]] for i in order], dim=3)
    assert torch.equal(dst[:, :, :, :o].contiguous().view(torch.int16), exp.contiguous().view(torch.int16))


# ---------------------------------------------------------------------------------------------------
# RoPE table + rotate/append
# ---------------------------------------------------------------------------------------------------

def _inv_freq(D, theta):
    # exactly the reference formula (llama2.py:121), evaluated by torch on the CPU
    return 1.0 / (theta ** (torch.arange(0, D, 2).float() / D))


@pytest.mark.parametrize("D,theta", [(32, 10000.0), (128, 10000.0), (128, 1e6)])
def test_rope_table_matches_oracle(D, theta):
    n = _n()
    pos = np.array([0, 1, 2, 3, 1981, 1992, 4095, 8191, 9185, 16383, 7, 7], dtype=np.int32)
    cs = torch.empty((len(pos), D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(torch.from_numpy(pos).to(DEV), _inv_freq(D, theta).to(DEV), cs, len(pos), D)
    torch.cuda.synchronize()
    cos, sin = orc.rope_cos_sin(pos[None], D, theta, _inv_freq(D, theta).numpy())
    cs = cs.cpu().numpy()
    # fp32 sincos of an fp32 angle: device libm vs numpy agree to a few ulp of 1.0
    np.testing.assert_allclose(cs[..., 0], cos[0, :, :D // 2], atol=2e-6, rtol=0)
    np.testing.assert_allclose(cs[..., 1], sin[0, :, :D // 2], atol=2e-6, rtol=0)


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("B,H,Hkv,D,q_len,past,cap", [
    (1, 4, 4, 32, 12, 40, 64), (2, 4, 2, 128, 5, 0, 8), (1, 32, 32, 128, 14, 100, 128), (1, 2, 2, 64, 1, 7, 8)])
def test_rope_append_matches_oracle(B, H, Hkv, D, q_len, past, cap, f32):
    n = _n()
    rng = np.random.default_rng(3)
    W = (H + 2 * Hkv) * D
    x32 = rng.standard_normal((B, q_len, W), dtype=np.float32)
    qkv = torch.from_numpy(x32 if f32 else x32.astype(np.float16)).to(DEV)
    qkv0 = qkv.clone()
    pos = rng.integers(0, 5000, size=(B, q_len)).astype(np.int32)
    cs = torch.empty((B * q_len, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(torch.from_numpy(pos.reshape(-1)).to(DEV), _inv_freq(D, 10000.0).to(DEV), cs, B * q_len, D)
    arena = _rand_half((B, 2, Hkv, cap, D), np.random.default_rng(4)).to(DEV)
    arena0 = arena.clone()
    k = qkv[:, :, H * D:(H + Hkv) * D]
    v = qkv[:, :, (H + Hkv) * D:]
    if f32:
        q_out = torch.zeros((B, q_len, H * D), dtype=torch.float16, device=DEV)
        qo_bs, qo_ts = q_len * H * D, H * D
    else:
        q_out, qo_bs, qo_ts = qkv, q_len * W, W      # fp16: rotate q in place
    n.rope_append(qkv, q_len * W, W, q_out, qo_bs, qo_ts, k, v, q_len * W, W, arena[:, 0], arena[:, 1],
                  2 * Hkv * cap * D, cap * D, cs, B, H, Hkv, D, q_len, past, cap, f32)
    torch.cuda.synchronize()
    cos, sin = orc.rope_cos_sin(pos, D, 10000.0, _inv_freq(D, 10000.0).numpy())
    x = qkv0.float().cpu().numpy()
    q_ref = orc.apply_rope(x[:, :, :H * D].reshape(B, q_len, H, D).transpose(0, 2, 1, 3), cos, sin)
    k_ref = orc.apply_rope(x[:, :, H * D:(H + Hkv) * D].reshape(B, q_len, Hkv, D).transpose(0, 2, 1, 3), cos, sin)
    q_got = q_out[:, :, :H * D].float().cpu().numpy().reshape(B, q_len, H, D).transpose(0, 2, 1, 3)
    k_got = arena[:, 0, :, past:past + q_len].float().cpu().numpy()
    # one fp16 rounding of an fp32 result: |err| <= 2^-11 * |x| (+ tiny sincos difference)
    np.testing.assert_allclose(q_got, q_ref, atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(k_got, k_ref, atol=1e-3, rtol=1e-3)
    # V is a copy (one fp16 rounding when the projection output is fp32); k/v columns of qkv untouched;
    # arena rows outside [past, past+q) untouched
    v_exp = qkv0[:, :, (H + Hkv) * D:].reshape(B, q_len, Hkv, D).permute(0, 2, 1, 3).half()
    assert torch.equal(arena[:, 1, :, past:past + q_len].contiguous().view(torch.int16), v_exp.contiguous().view(torch.int16))
    assert torch.equal(qkv[:, :, H * D:], qkv0[:, :, H * D:])
    mask = torch.ones(cap, dtype=torch.bool)
    mask[past:past + q_len] = False
    mask = mask.to(DEV)
    assert torch.equal(arena[:, :, :, mask].view(torch.int16), arena0[:, :, :, mask].view(torch.int16))


def test_rope_append_bounds():
    n = _n()
    qkv = torch.zeros((1, 4, 3 * 32), dtype=torch.float16, device=DEV)
    cs = torch.zeros((4, 16, 2), dtype=torch.float32, device=DEV)
    arena = torch.zeros((1, 2, 1, 8, 32), dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="exceeds arena rows"):
        n.rope_append(qkv, 0, 96, qkv, 0, 96, qkv, qkv, 0, 96, arena[:, 0], arena[:, 1], 0, 256, cs, 1, 1, 1, 32, 4, 5, 8, False)


# ---------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------

def _run_attn(q, k, v, past, scale=None):
    """q [B,q,H,D] fp16 cuda; k, v [B,Hkv,cap,D] fp16 cuda with past+q valid rows."""
    n = _n()
    B, ql, H, D = q.shape
    Hkv, cap = k.shape[1], k.shape[2]
    out = torch.full((B, ql, H * D), float("nan"), dtype=torch.float16, device=DEV)
    ws_bytes = n.attn_workspace_bytes(B, H, D, ql, past + ql)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=DEV)
    n.attn_fwd(q, ql * H * D, H * D, k, v, Hkv * cap * D, cap * D, out, ql * H * D, H * D, B, H, Hkv, D, ql, past,
               (1.0 / np.sqrt(D)) if scale is None else scale, ws)
    torch.cuda.synchronize()
    return out, ws_bytes


def _ref_attn(q, k, v, past):
    B, ql, H, D = q.shape
    Hkv = k.shape[1]
    qn = q.float().cpu().numpy().transpose(0, 2, 1, 3)
    kn = k[:, :, :past + ql].float().cpu().numpy()
    vn = v[:, :, :past + ql].float().cpu().numpy()
    o = orc.attention_core(qn, kn, vn, past, H // Hkv)
    if ql == 1:  # oracle skips the mask for q_len==1 exactly as the reference; same thing here
        pass
    return o.transpose(0, 2, 1, 3).reshape(B, ql, H * D)


ATTN_CASES = [
    # B, H, Hkv, D, q_len, past
    (1, 4, 4, 32, 12, 40),
    (1, 4, 4, 32, 1, 0),
    (1, 4, 4, 32, 16, 0),
    (1, 4, 4, 32, 17, 3),
    (2, 4, 2, 128, 12, 300),      # GQA + batch, split-KV
    (1, 32, 32, 128, 12, 1725),   # persona cached prefill shape (one layer)
    (1, 32, 32, 128, 14, 4390),   # game cached prefill shape
    (1, 2, 2, 64, 50, 129),
    (1, 2, 2, 128, 64, 0),
    (1, 2, 2, 128, 65, 0),
    (1, 4, 4, 128, 200, 77),      # several q blocks, causal diagonal inside tiles
    (1, 2, 1, 128, 456, 0),       # encode regime (q = S)
    (1, 8, 8, 128, 1, 1000),      # decode step
    (1, 40, 40, 128, 30, 511),    # 13b head count
    # the 32-rows-per-wave kernel (attn_fwd32_kernel, mfma 32x32x16): dispatched when q_len > 64 and 128-row q-blocks fill
    # the chip (B*H*ceil(q/128) >= 256) or the staged past is long (kv_len >= 8 q_len) -- pc_attn.hip use_rows32()
    (1, 32, 32, 128, 1100, 0),    # 7b heads, encode regime: 288 q-blocks, ragged last block, causal diagonal
    (2, 16, 8, 128, 1030, 5),     # batch + GQA, 288 q-blocks
    (1, 8, 8, 128, 70, 600),      # long staged past (kv >= 8 q): KV splits + merge from the 32-row kernel
    (1, 64, 64, 64, 600, 3),      # D = 64
]


@pytest.mark.parametrize("B,H,Hkv,D,q_len,past", ATTN_CASES)
def test_attn_matches_oracle(B, H, Hkv, D, q_len, past):
    rng = np.random.default_rng(5)
    cap = past + q_len + 3
    q = torch.from_numpy(rng.standard_normal((B, q_len, H, D), dtype=np.float32).astype(np.float16)).to(DEV)
    k = torch.from_numpy(rng.standard_normal((B, Hkv, cap, D), dtype=np.float32).astype(np.float16)).to(DEV)
    v = torch.from_numpy(rng.standard_normal((B, Hkv, cap, D), dtype=np.float32).astype(np.float16)).to(DEV)
    # poison the rows past the valid range: they must never contribute (NaN would propagate)
    k[:, :, past + q_len:] = float("nan")
    v[:, :, past + q_len:] = float("nan")
    out, _ = _run_attn(q, k, v, past)
    ref = _ref_attn(q, k, v, past)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    # fp16 P (2^-11 relative per weight, averaged by the softmax) and fp16 output rounding of O(1) values
    np.testing.assert_allclose(got, ref, atol=4e-3, rtol=1e-2)


FUSED_CASES = [
    # B, H, Hkv, D, q_len, past, tail (split-precision rows of the pass itself: the extra fp32 workgroup)
    (1, 32, 32, 128, 12, 1725, True),     # the persona cached prefill (one layer): 8 stream splits + tail
    (1, 32, 32, 128, 12, 1725, False),
    (2, 4, 2, 128, 12, 300, True),        # batch + GQA: 15 + 1 splits (the 16-partial merge)
    (1, 8, 8, 128, 1, 1000, False),       # decode step
    (1, 40, 40, 128, 16, 511, True),      # 13b head count, a full 16-row tile
    (1, 64, 64, 64, 3, 600, True),        # D = 64: 4 splits
    (1, 4, 4, 32, 12, 400, False),        # D = 32
    (1, 128, 128, 128, 7, 2100, True),    # 2 stream splits + tail, several tiles per wave
]


@pytest.mark.parametrize("B,H,Hkv,D,q_len,past,tail", FUSED_CASES)
@pytest.mark.parametrize("frag", [False, True])
def test_attn_single_launch_merge_equals_two_launch_merge_and_oracle(B, H, Hkv, D, q_len, past, tail, frag):
    """pc_attn with arrival counters (the last-arriving workgroup of a head merges the split-KV partials inside the launch)
    against the two-launch form (attn_combine_kernel) -- same partials, same merge arithmetic in split order: bit-identical --
    and against the oracle; repeated launches over the same counters (they must come back to zero), the consumer's L1 warm."""
    n = _n()
    rng = np.random.default_rng(11)
    cap = past + q_len + 5
    f16 = lambda a: torch.from_numpy(a.astype(np.float16)).to(DEV)             # noqa: E731
    q32 = rng.standard_normal((B, q_len, H, D), dtype=np.float32)
    q = f16(q32)
    ql = f16(q32 - q.float().cpu().numpy())
    k = f16(rng.standard_normal((B, Hkv, cap, D), dtype=np.float32))
    v = f16(rng.standard_normal((B, Hkv, cap, D), dtype=np.float32))
    k[:, :, past + q_len:] = float("nan")
    v[:, :, past + q_len:] = float("nan")
    klo = f16(1e-4 * rng.standard_normal((B, Hkv, q_len, D), dtype=np.float32))
    vlo = f16(1e-4 * rng.standard_normal((B, Hkv, q_len, D), dtype=np.float32))
    kv_lo = (klo, vlo, Hkv * q_len * D, q_len * D, -1) if tail else None
    ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, past + q_len), 4) // 4, dtype=torch.float32, device=DEV)
    counters = torch.zeros(B * H, dtype=torch.int32, device=DEV)
    mt = (B * q_len + 15) // 16
    scale = 1.0 / np.sqrt(D)

    def run(ctr):
        ws.fill_(float("nan"))
        if frag:
            oh = torch.full((mt, H * D // 32, 64, 8), float("nan"), dtype=torch.float16, device=DEV)
            ol = torch.full_like(oh, float("nan"))
            n.attn_fwd(q, q_len * H * D, H * D, k, v, Hkv * cap * D, cap * D, None, 0, 0, B, H, Hkv, D, q_len, past, scale, ws,
                       out_frag=(oh, ol), q_lo=ql, kv_lo=kv_lo, counters=ctr)
        else:
            oh = torch.full((B, q_len, H * D), float("nan"), dtype=torch.float16, device=DEV)
            ol = torch.full_like(oh, float("nan"))
            n.attn_fwd(q, q_len * H * D, H * D, k, v, Hkv * cap * D, cap * D, oh, q_len * H * D, H * D, B, H, Hkv, D, q_len, past,
                       scale, ws, q_lo=ql, out_lo=ol, kv_lo=kv_lo, counters=ctr)
        torch.cuda.synchronize()
        return oh, ol

    two_hi, two_lo = run(None)
    for rep in range(3):
        one_hi, one_lo = run(counters)
        assert int(counters.abs().sum()) == 0, "every launch leaves the counters at zero"
        # pad rows of the fragment planes are never written by either form: compare what both wrote
        m = ~torch.isnan(two_hi.float())
        assert torch.equal(~torch.isnan(one_hi.float()), m)
        one = one_hi.float()[m].double() + one_lo.float()[m].double()
        two = two_hi.float()[m].double() + two_lo.float()[m].double()
        # the same partials through the same merge formula; the two kernels may contract a*b+c differently: fp32 round-off
        d = (one - two).abs()
        assert float(d.max()) <= 4e-7 * float(two.abs().max()) + 1e-9, (float(d.max()), int((d > 0).sum()), d.numel())
        if rep == 0:
            first = (one_hi.clone(), one_lo.clone())
        else:     # run to run: bit-identical whatever the arrival order was
            assert torch.equal(one_hi.view(torch.int16)[m], first[0].view(torch.int16)[m])
            assert torch.equal(one_lo.view(torch.int16)[m], first[1].view(torch.int16)[m])
    if not frag:
        kf = k.float().clone(); vf = v.float().clone()
        if tail:
            kf[:, :, past:past + q_len] += klo.float(); vf[:, :, past:past + q_len] += vlo.float()
        qn = (q.float() + ql.float()).cpu().numpy().transpose(0, 2, 1, 3)
        ref = orc.attention_core(qn, kf[:, :, :past + q_len].cpu().numpy(), vf[:, :, :past + q_len].cpu().numpy(), past, H // Hkv)
        ref = ref.transpose(0, 2, 1, 3).reshape(B, q_len, H * D)
        got = (one_hi.float() + one_lo.float()).cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=2e-4, rtol=2e-3)


@pytest.mark.parametrize("B,H,Hkv,D,q_len,past", [(1, 32, 32, 128, 26, 1725), (2, 8, 2, 128, 17, 300), (1, 16, 16, 64, 32, 777), (1, 4, 4, 32, 21, 400),
                                                   (1, 40, 40, 128, 31, 250)])
@pytest.mark.parametrize("frag", [False, True])
def test_attn_17_to_32_rows_over_a_long_cache_matches_the_oracle(B, H, Hkv, D, q_len, past, frag):
    """17..32 new rows in tail mode over >= 256 keys: attn_small_kernel with two row tiles per wave (+ the 32-row fp32 tail
    workgroup, + attn_combine_kernel) against oracle.attention_core on the split-precision operands."""
    n = _n()
    rng = np.random.default_rng(23)
    cap = past + q_len + 3
    f16 = lambda a: torch.from_numpy(a.astype(np.float16)).to(DEV)             # noqa: E731
    q32 = rng.standard_normal((B, q_len, H, D), dtype=np.float32)
    q = f16(q32)
    ql = f16(q32 - q.float().cpu().numpy())
    k = f16(rng.standard_normal((B, Hkv, cap, D), dtype=np.float32))
    v = f16(rng.standard_normal((B, Hkv, cap, D), dtype=np.float32))
    k[:, :, past + q_len:] = float("nan")
    v[:, :, past + q_len:] = float("nan")
    klo = f16(1e-4 * rng.standard_normal((B, Hkv, q_len, D), dtype=np.float32))
    vlo = f16(1e-4 * rng.standard_normal((B, Hkv, q_len, D), dtype=np.float32))
    kv_lo = (klo, vlo, Hkv * q_len * D, q_len * D, -1)
    ws = torch.full((max(n.attn_workspace_bytes(B, H, D, q_len, past + q_len), 4) // 4,), float("nan"), dtype=torch.float32, device=DEV)
    scale = 1.0 / np.sqrt(D)
    if frag:
        mt = (B * q_len + 15) // 16
        oh = torch.full((mt, H * D // 32, 64, 8), float("nan"), dtype=torch.float16, device=DEV)
        ol = torch.full_like(oh, float("nan"))
        n.attn_fwd(q, q_len * H * D, H * D, k, v, Hkv * cap * D, cap * D, None, 0, 0, B, H, Hkv, D, q_len, past, scale, ws,
                   out_frag=(oh, ol), q_lo=ql, kv_lo=kv_lo)
        got = (n.from_act_frags(oh, B * q_len).float() + n.from_act_frags(ol, B * q_len).float()).view(B, q_len, H * D).cpu().numpy()
    else:
        oh = torch.full((B, q_len, H * D), float("nan"), dtype=torch.float16, device=DEV)
        ol = torch.full_like(oh, float("nan"))
        n.attn_fwd(q, q_len * H * D, H * D, k, v, Hkv * cap * D, cap * D, oh, q_len * H * D, H * D, B, H, Hkv, D, q_len, past, scale, ws,
                   q_lo=ql, out_lo=ol, kv_lo=kv_lo)
        got = (oh.float() + ol.float()).cpu().numpy()
    torch.cuda.synchronize()
    kf = k.float().clone(); vf = v.float().clone()
    kf[:, :, past:past + q_len] += klo.float(); vf[:, :, past:past + q_len] += vlo.float()
    qn = (q.float() + ql.float()).cpu().numpy().transpose(0, 2, 1, 3)
    ref = orc.attention_core(qn, kf[:, :, :past + q_len].cpu().numpy(), vf[:, :, :past + q_len].cpu().numpy(), past, H // Hkv)
    ref = ref.transpose(0, 2, 1, 3).reshape(B, q_len, H * D)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=2e-4)


def test_attn_softmax_rescale_with_key_spike():
    """Force the running max to jump late in the KV stream (online-softmax rescale path) and early
    (later tiles contribute ~0): a spike row in K aligned with one query."""
    rng = np.random.default_rng(6)
    B, H, D, q_len, past = 1, 2, 128, 12, 700
    cap = past + q_len
    q = torch.from_numpy(rng.standard_normal((B, q_len, H, D), dtype=np.float32).astype(np.float16)).to(DEV)
    k = torch.from_numpy((0.3 * rng.standard_normal((B, H, cap, D), dtype=np.float32)).astype(np.float16)).to(DEV)
    v = torch.from_numpy(rng.standard_normal((B, H, cap, D), dtype=np.float32).astype(np.float16)).to(DEV)
    for spike_at in (650, 5):
        k2 = k.clone()
        k2[0, 0, spike_at] = q[0, 3, 0] * 2.0      # q.k ~ 2*|q|^2 = 256 -> scaled ~ 22 above the rest
        out, _ = _run_attn(q, k2, v, past)
        ref = _ref_attn(q, k2, v, past)
        np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=4e-3, rtol=1e-2)


def test_attn_index_order_mask_not_position_order():
    """Permuting the *past* keys (with their values) must not change the output: every new token sees
    all staged KV regardless of order (llama2.py:62-76 gives zeros over all past columns)."""
    rng = np.random.default_rng(7)
    B, H, D, q_len, past = 1, 4, 128, 9, 333
    cap = past + q_len
    q = torch.from_numpy(rng.standard_normal((B, q_len, H, D), dtype=np.float32).astype(np.float16)).to(DEV)
    k = torch.from_numpy(rng.standard_normal((B, H, cap, D), dtype=np.float32).astype(np.float16)).to(DEV)
    v = torch.from_numpy(rng.standard_normal((B, H, cap, D), dtype=np.float32).astype(np.float16)).to(DEV)
    out1, _ = _run_attn(q, k, v, past)
    perm = torch.from_numpy(rng.permutation(past)).to(DEV)
    k2, v2 = k.clone(), v.clone()
    k2[:, :, :past] = k[:, :, perm]
    v2[:, :, :past] = v[:, :, perm]
    out2, _ = _run_attn(q, k2, v2, past)
    np.testing.assert_allclose(out1.float().cpu().numpy(), out2.float().cpu().numpy(), atol=2e-3, rtol=0)


def test_attn_full_size_property_uniform_values():
    """BASELINE-size property check (13b heads, 8k staged keys): with all V rows equal to one vector the
    output equals that vector for every query, whatever the scores are."""
    rng = np.random.default_rng(8)
    B, H, D, q_len, past = 1, 40, 128, 260, 8000
    cap = past + q_len
    q = torch.from_numpy(rng.standard_normal((B, q_len, H, D), dtype=np.float32).astype(np.float16)).to(DEV)
    k = torch.from_numpy(rng.standard_normal((B, H, cap, D), dtype=np.float32).astype(np.float16)).to(DEV)
    vec = torch.from_numpy(rng.standard_normal((H, D), dtype=np.float32).astype(np.float16)).to(DEV)
    v = vec[None, :, None, :].expand(B, H, cap, D).contiguous()
    out, _ = _run_attn(q, k, v, past)
    exp = vec.reshape(1, 1, H * D).expand(B, q_len, H * D)
    np.testing.assert_allclose(out.float().cpu().numpy(), exp.float().cpu().numpy(), atol=2e-3, rtol=2e-3)


def test_attn_errors():
    n = _n()
    t = torch.zeros(4096, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="unsupported"):
        n.attn_fwd(t, 0, 96, t, t, 0, 96, t, 0, 96, 1, 1, 1, 96, 1, 0, 1.0)
    with pytest.raises(RuntimeError, match="workspace"):
        # long KV + tiny q -> split-KV needs a workspace
        big = torch.zeros(2 * 2048 * 128, dtype=torch.float16, device=DEV)
        n.attn_fwd(big, 0, 256, big, big, 0, 2048 * 128, big, 0, 256, 1, 2, 2, 128, 1, 2000, 1.0)


# ---------------------------------------------------------------------------------------------------
# elementwise pieces
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("rows,hidden,f32", [(12, 4096, True), (3, 128, False), (14, 5120, True), (1, 256, False)])
def test_rmsnorm_matches_oracle(rows, hidden, f32):
    n = _n()
    rng = np.random.default_rng(9)
    x = rng.standard_normal((rows, hidden), dtype=np.float32) * 3
    w = (1 + 0.1 * rng.standard_normal(hidden, dtype=np.float32)).astype(np.float16)
    xd = torch.from_numpy(x if f32 else x.astype(np.float16)).to(DEV)
    out = torch.empty((rows, hidden), dtype=torch.float16, device=DEV)
    n.rmsnorm(xd, torch.from_numpy(w).to(DEV), out, rows, hidden, 1e-5, f32)
    torch.cuda.synchronize()
    ref = orc.rmsnorm(xd.float().cpu().numpy(), w.astype(np.float32), 1e-5)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=4e-3, rtol=4e-3)  # two fp16 roundings


def test_silu_mul_and_embed():
    n = _n()
    rng = np.random.default_rng(10)
    rows, inter = 5, 11008
    g32 = rng.standard_normal((rows, 2 * inter), dtype=np.float32)
    for f32 in (False, True):
        gu = torch.from_numpy(g32 if f32 else g32.astype(np.float16)).to(DEV)
        out = torch.empty((rows, inter), dtype=torch.float16, device=DEV)
        n.silu_mul(gu, out, rows, inter, f32)
        g = gu.float().cpu().numpy()
        ref = orc.silu(g[:, :inter]) * g[:, inter:]
        np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=2e-3, rtol=2e-3)
    table = torch.from_numpy(rng.standard_normal((100, 128), dtype=np.float32).astype(np.float16)).to(DEV)
    ids = torch.tensor([0, 99, 5, 5, 42], dtype=torch.int64, device=DEV)
    e = torch.empty((5, 128), dtype=torch.float16, device=DEV)
    n.embed_gather(table, ids, e, 5, 128, 100)
    torch.cuda.synchronize()
    assert torch.equal(e, table[ids])


# ---------------------------------------------------------------------------------------------------
# weight-streaming projections (pc_gemm.hip)
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("M,N,K", [(12, 12288, 4096), (1, 4096, 4096), (16, 4096, 11008), (17, 256, 128), (33, 1024, 512),
                                   (64, 4096, 4096), (12, 32000, 4096), (5, 48, 32), (14, 15360, 5120),
                                   # wide workgroups: more tiles than one reduction round holds (MT*T > 8)
                                   (64, 12288, 4096), (24, 32000, 4096), (48, 16384, 512)])
@pytest.mark.parametrize("two_pass", [True, False])
def test_gemm_skinny_store_and_add(M, N, K, two_pass):
    n = _n()
    rng = np.random.default_rng(11)
    w = torch.from_numpy((0.05 * rng.standard_normal((N, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    wf = n.to_weight_frags(w)
    hi, lo = n.to_act_frags(x)
    # poison the pad rows of the planes: they may only reach output columns that are never stored
    pad = n.from_act_frags(hi, hi.shape[0] * 16)
    y = torch.full((M, N), 7.0, dtype=torch.float32, device=DEV)
    n.gemm_skinny(wf, hi, lo if two_pass else None, M, N, K, n.EPI_STORE, y=y, ldy=N)
    xin = x if two_pass else x.half().float()
    ref = (xin.double() @ w.double().t()).float()
    # split-precision activations keep ~22 bits: fp32-accumulation-level agreement with an fp64 reference
    tol = 2e-4 * float(ref.abs().max()) + 1e-5
    assert (y - ref).abs().max().item() < tol
    y2 = torch.full((M, N), 7.0, dtype=torch.float32, device=DEV)
    n.gemm_skinny(wf, hi, lo if two_pass else None, M, N, K, n.EPI_ADD, y=y2, ldy=N)
    assert (y2 - 7.0 - ref).abs().max().item() < tol + 1e-5
    # deterministic: fixed-order split-K reduction
    y3 = torch.empty_like(y)
    n.gemm_skinny(wf, hi, lo if two_pass else None, M, N, K, n.EPI_STORE, y=y3, ldy=N)
    assert torch.equal(y, y3)


@pytest.mark.parametrize("M,inter,K", [(12, 11008, 4096), (3, 64, 32), (20, 1376, 512), (50, 13824, 5120),
                                       (64, 11008, 4096), (24, 13824, 5120), (31, 11008, 4096)])
def test_gemm_skinny_silu_epilogue(M, inter, K):
    n = _n()
    rng = np.random.default_rng(12)
    w = torch.from_numpy((0.05 * rng.standard_normal((2 * inter, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    hi, lo = n.to_act_frags(x)
    mt = (M + 15) // 16
    oh = torch.zeros((mt, inter // 32, 64, 8), dtype=torch.float16, device=DEV)
    ol = torch.zeros_like(oh)
    n.gemm_skinny(n.to_weight_frags(w), hi, lo, M, 2 * inter, K, n.EPI_SILU, of_hi=oh, of_lo=ol)
    gu = x.double() @ w.double().t()
    g, u = gu[:, :inter], gu[:, inter:]
    ref = (g / (1 + torch.exp(-g)) * u).float()
    got = n.from_act_frags(oh, M).float() + n.from_act_frags(ol, M).float()
    assert (got - ref).abs().max().item() < 2e-4 * float(ref.abs().max()) + 1e-5


@pytest.mark.parametrize("nslabs", [0, 4])
@pytest.mark.parametrize("rows,hidden", [(12, 4096), (1, 128), (33, 5120)])
def test_rmsnorm_frag_matches_oracle(rows, hidden, nslabs):
    n = _n()
    rng = np.random.default_rng(13)
    x = rng.standard_normal((rows, hidden), dtype=np.float32) * 3
    sl = rng.standard_normal((max(nslabs, 1), rows, hidden), dtype=np.float32)
    w = (1 + 0.1 * rng.standard_normal(hidden, dtype=np.float32)).astype(np.float16)
    mt = (rows + 15) // 16
    hi = torch.zeros((mt, hidden // 32, 64, 8), dtype=torch.float16, device=DEV)
    lo = torch.zeros_like(hi)
    xd = torch.from_numpy(x).to(DEV)
    n.rmsnorm_frag(xd, torch.from_numpy(w).to(DEV), hi, lo, rows, hidden, 1e-5, torch.from_numpy(sl).to(DEV), nslabs)
    xe = x.copy()
    for s_ in range(nslabs):
        xe = xe + sl[s_]                       # fixed order, fp32
    np.testing.assert_array_equal(xd.cpu().numpy(), xe)          # residual updated in place, bit-exact
    got = (n.from_act_frags(hi, rows).float() + n.from_act_frags(lo, rows).float()).cpu().numpy()
    ref = orc.rmsnorm(xe, w.astype(np.float32), 1e-5)
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("M,N,K,kq", [(12, 4096, 4096, 4), (12, 4096, 11008, 4), (3, 64, 96, 2), (20, 512, 1376, 3),
                                      (40, 4096, 4096, 4), (64, 4096, 11008, 4), (30, 5120, 13824, 4)])
def test_gemm_skinny_k_slices(M, N, K, kq):
    """K-sliced launch: the slabs add up to the full product."""
    n = _n()
    rng = np.random.default_rng(15)
    w = torch.from_numpy((0.05 * rng.standard_normal((N, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    hi, lo = n.to_act_frags(x)
    slabs = torch.full((kq, M, N), 3.0, dtype=torch.float32, device=DEV)
    n.gemm_skinny(n.to_weight_frags(w), hi, lo, M, N, K, n.EPI_STORE, y=slabs, ldy=N, kslices=kq)
    ref = (x.double() @ w.double().t()).float()
    assert (slabs.sum(0) - ref).abs().max().item() < 2e-4 * float(ref.abs().max()) + 1e-5


@pytest.mark.parametrize("M,N,K,S,T", [(12, 4096, 4096, 4, 4), (12, 4096, 11008, 8, 8), (12, 4096, 11008, 2, 1), (1, 4096, 11008, 4, 2),
                                       (16, 5120, 13824, 8, 4), (3, 64, 96, 2, 1), (7, 80, 1376, 3, 2), (12, 4096, 4096, 1, 1),
                                       # (17..32 rows: two row tiles per workgroup -- the questions of BASELINE config 3)
                                       (26, 4096, 11008, 2, 2), (32, 4096, 4096, 4, 4), (17, 4096, 4096, 1, 1), (22, 5120, 13824, 4, 2),
                                       (31, 80, 1376, 3, 1)])
def test_gemm_skinny_ks_in_launch_reduction(M, N, K, S, T):
    """pc_gemm_skinny_ks: K split across workgroups, partials added inside the launch by the last arriver -- against an fp64
    reference, bit-reproducible over repeated launches (slice-order sum whoever arrives last), counters back at zero."""
    n = _n()
    rng = np.random.default_rng(19)
    w = torch.from_numpy((0.05 * rng.standard_normal((N, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    wf = n.to_weight_frags(w)
    hi, lo = n.to_act_frags(x)
    scratch = torch.full(((2 if M > 16 else 1) * n.gemm_skinny_ks_scratch_bytes(N, S) // 4,), float("nan"), dtype=torch.float32, device=DEV)
    counters = torch.zeros((N // 16 + T - 1) // T, dtype=torch.int32, device=DEV)
    ref = (x.double() @ w.double().t()).float()
    tol = 2e-4 * float(ref.abs().max()) + 1e-5
    outs = []
    for rep in range(4):
        y = torch.full((M, N), 7.0, dtype=torch.float32, device=DEV)
        n.gemm_skinny_ks(wf, hi, lo, M, N, K, y, N, S, T, scratch, counters)
        torch.cuda.synchronize()
        assert int(counters.abs().sum()) == 0
        assert (y - 7.0 - ref).abs().max().item() < tol + 1e-5
        outs.append(y)
    assert all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("B,H,Hkv,D,q_len,past", [(1, 32, 32, 128, 12, 1725), (1, 4, 4, 32, 17, 3), (2, 4, 2, 128, 12, 300),
                                                  (1, 2, 2, 64, 50, 129)])
def test_attn_fragment_plane_output(B, H, Hkv, D, q_len, past):
    """Same attention, result delivered as split-precision fragment planes (what o_proj consumes)."""
    n = _n()
    rng = np.random.default_rng(14)
    cap = past + q_len
    q = torch.from_numpy(rng.standard_normal((B, q_len, H, D), dtype=np.float32).astype(np.float16)).to(DEV)
    k = torch.from_numpy(rng.standard_normal((B, Hkv, cap, D), dtype=np.float32).astype(np.float16)).to(DEV)
    v = torch.from_numpy(rng.standard_normal((B, Hkv, cap, D), dtype=np.float32).astype(np.float16)).to(DEV)
    T = B * q_len
    mt = (T + 15) // 16
    fh = torch.zeros((mt, H * D // 32, 64, 8), dtype=torch.float16, device=DEV)
    fl = torch.zeros_like(fh)
    ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, cap), 4) // 4, dtype=torch.float32, device=DEV)
    n.attn_fwd(q, q_len * H * D, H * D, k, v, Hkv * cap * D, cap * D, None, 0, 0, B, H, Hkv, D, q_len, past,
               1.0 / np.sqrt(D), ws, out_frag=(fh, fl))
    torch.cuda.synchronize()
    got = (n.from_act_frags(fh, T).float() + n.from_act_frags(fl, T).float()).cpu().numpy().reshape(B, q_len, H * D)
    ref = _ref_attn(q, k, v, past)
    np.testing.assert_allclose(got, ref, atol=3e-3, rtol=1e-2)   # fp16 P inside the kernel; fp32-ish output


@pytest.mark.parametrize("B,H,Hkv,D,q_len,past,hid", [(1, 32, 32, 128, 12, 100, 4096), (2, 4, 2, 128, 5, 7, 512), (1, 4, 4, 32, 17, 0, 128),
                                                       (1, 2, 2, 64, 1, 30, 256), (1, 32, 32, 128, 40, 100, 4096),
                                                       (2, 32, 32, 128, 12, 9, 4096), (1, 40, 40, 128, 64, 0, 5120)])
def test_gemm_qkv_rope_fused_equals_unfused(B, H, Hkv, D, q_len, past, hid):
    """Fused projection + RoPE + KV-append epilogue == plain-store GEMM followed by pc_rope_append: same dot
    products (only the weight rows are permuted inside the tiles), same rotation up to how the compiler contracts
    a*c -/+ b*s into FMAs, so the fp16 results agree to one ulp and almost always exactly."""
    n = _n()
    rng = np.random.default_rng(16)
    T = B * q_len
    W = (H + 2 * Hkv) * D
    cap = past + q_len + 2
    w = torch.from_numpy((0.05 * rng.standard_normal((W, hid), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((T, hid), dtype=np.float32)).to(DEV)
    hi, lo = n.to_act_frags(x)
    pos = rng.integers(0, 3000, size=T).astype(np.int32)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(torch.from_numpy(pos).to(DEV), _inv_freq(D, 10000.0).to(DEV), cs, T, D)
    # unfused reference path
    qkv = torch.empty((T, W), dtype=torch.float32, device=DEV)
    n.gemm_skinny(n.to_weight_frags(w), hi, lo, T, W, hid, n.EPI_STORE, y=qkv, ldy=W)
    arena_a = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    qa = torch.zeros((T, H * D), dtype=torch.float16, device=DEV)
    qal = torch.zeros_like(qa)
    n.rope_append(qkv, q_len * W, W, qa, q_len * H * D, H * D, qkv[:, H * D:], qkv[:, (H + Hkv) * D:], q_len * W, W,
                  arena_a[:, 0], arena_a[:, 1], 2 * Hkv * cap * D, cap * D, cs, B, H, Hkv, D, q_len, past, cap, True, q_out_lo=qal)
    # fused path
    perm = n.qkv_rope_row_perm(H + 2 * Hkv, D).to(DEV)
    arena_b = torch.zeros_like(arena_a)
    qb = torch.zeros_like(qa)
    qbl = torch.zeros_like(qa)
    n.gemm_qkv_rope(n.to_weight_frags(w[perm].contiguous()), hi, lo, T, hid, cs, qb, qbl, H * D, arena_b[:, 0], arena_b[:, 1],
                    2 * Hkv * cap * D, cap * D, B, H, Hkv, D, q_len, past, cap)
    torch.cuda.synchronize()
    def close(a, b):
        d = (a.float() - b.float()).abs()
        assert float(d.max()) <= 2e-3 * max(1.0, float(a.float().abs().max())), float(d.max())     # <= 1 fp16 ulp
        assert float((a != b).float().mean()) < 0.02
    close(qa, qb)
    close(arena_a, arena_b)
    assert torch.equal(arena_a[:, 1], arena_b[:, 1])                       # V is a pure fp32 -> fp16 conversion
    rec_a, rec_b = qa.float() + qal.float(), qb.float() + qbl.float()      # split-precision q: ~fp32 agreement
    assert float((rec_a - rec_b).abs().max()) < 1e-5 * max(1.0, float(rec_a.abs().max()))
    assert float(arena_b[:, :, :, past:past + q_len].abs().sum()) > 0 and float(arena_b[:, :, :, :past].abs().sum()) == 0


# ---------------------------------------------------------------------------------------------------
# 65..512 rows: the row-split weight-streaming kernel (hi + lo activation planes; hi only when lo is None)
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("M,N,K,kq", [(65, 4096, 4096, 1), (100, 12288, 4096, 1), (259, 15360, 5120, 1), (512, 4096, 4096, 1),
                                      (130, 48, 32, 1), (70, 64, 96, 2), (259, 5120, 13824, 4), (200, 4096, 11008, 4),
                                      (96, 32000, 4096, 1), (70, 32, 64, 4),      # (70, ..., 4): two of the four K slices are empty
                                      # two row blocks whose SECOND block is a compute wave shorter than the first (21 / 25 / 29 row
                                      # tiles: M = 321..336, 385..400, 449..464) -- the launch and the kernel must agree on the wave split
                                      (330, 4096, 4096, 2), (450, 5120, 13824, 4), (336, 4096, 11008, 3), (390, 2048, 1024, 1),
                                      # an odd number of row tiles in one block: the last tile is multiplied by the staging waves (round 6, XT);
                                      # K ranges that end inside a stage, more slices than stages
                                      (225, 5120, 5120, 2), (257, 4096, 11008, 6), (273 - 16, 128, 224, 3), (81 + 16, 256, 96, 4)])
def test_gemm_rows_store_add_slices(M, N, K, kq):
    n = _n()
    rng = np.random.default_rng(21)
    w = torch.from_numpy((0.05 * rng.standard_normal((N, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    wf = n.to_weight_frags(w)
    hi, lo = n.to_act_frags(x)
    ref = (x.double() @ w.double().t()).float()                 # hi + lo planes: ~22-bit activations
    ref_hi = (x.half().double() @ w.double().t()).float()       # hi plane only: fp16(x)
    tol = 2e-4 * float(ref.abs().max()) + 1e-5
    y = torch.full((kq, M, N), 7.0, dtype=torch.float32, device=DEV)
    n.gemm_skinny(wf, hi, lo, M, N, K, n.EPI_STORE, y=y, ldy=N, kslices=kq)
    assert (y.sum(dim=0) - ref).abs().max().item() < tol * kq
    y3 = torch.empty_like(y)
    n.gemm_skinny(wf, hi, lo, M, N, K, n.EPI_STORE, y=y3, ldy=N, kslices=kq)
    assert torch.equal(y, y3)
    if kq == 1:
        y2 = torch.full((M, N), 7.0, dtype=torch.float32, device=DEV)
        n.gemm_skinny(wf, hi, None, M, N, K, n.EPI_ADD, y=y2, ldy=N)
        assert (y2 - 7.0 - ref_hi).abs().max().item() < tol + 1e-5


@pytest.mark.parametrize("M,inter,K", [(65, 11008, 4096), (259, 13824, 5120), (300, 64, 32), (128, 1376, 512), (512, 11008, 4096),
                                       (225, 1376, 544), (97, 13824, 5120)])
def test_gemm_rows_silu_epilogue(M, inter, K):
    n = _n()
    rng = np.random.default_rng(22)
    w = torch.from_numpy((0.05 * rng.standard_normal((2 * inter, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    hi, lo = n.to_act_frags(x)
    mt = (M + 15) // 16
    oh = torch.zeros((mt, inter // 32, 64, 8), dtype=torch.float16, device=DEV)
    ol = torch.zeros_like(oh)
    n.gemm_skinny(n.to_weight_frags(w), hi, lo, M, 2 * inter, K, n.EPI_SILU, of_hi=oh, of_lo=ol)
    gu = x.double() @ w.double().t()
    g, u = gu[:, :inter], gu[:, inter:]
    ref = (g / (1 + torch.exp(-g)) * u).float()
    got = n.from_act_frags(oh, M).float() + n.from_act_frags(ol, M).float()
    assert (got - ref).abs().max().item() < 2e-4 * float(ref.abs().max()) + 1e-5


@pytest.mark.parametrize("B,H,Hkv,D,q_len,past,hid", [(1, 32, 32, 128, 100, 50, 4096), (1, 40, 40, 128, 259, 0, 5120),
                                                       (2, 4, 2, 128, 40, 7, 512), (1, 4, 4, 32, 65, 3, 128),
                                                       (1, 8, 8, 128, 400, 5, 1024), (2, 4, 4, 128, 150, 0, 512)])   # 289..512 rows: two row blocks
def test_gemm_rows_qkv_rope(B, H, Hkv, D, q_len, past, hid):
    """Row-split kernel with the fused RoPE / KV-append epilogue against the plain-store launch + pc_rope_append."""
    n = _n()
    rng = np.random.default_rng(23)
    T = B * q_len
    W = (H + 2 * Hkv) * D
    cap = past + q_len + 2
    w = torch.from_numpy((0.05 * rng.standard_normal((W, hid), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((T, hid), dtype=np.float32)).to(DEV)
    hi, lo = n.to_act_frags(x)
    pos = rng.integers(0, 3000, size=T).astype(np.int32)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(torch.from_numpy(pos).to(DEV), _inv_freq(D, 10000.0).to(DEV), cs, T, D)
    qkv = torch.empty((T, W), dtype=torch.float32, device=DEV)
    n.gemm_skinny(n.to_weight_frags(w), hi, lo, T, W, hid, n.EPI_STORE, y=qkv, ldy=W)
    arena_a = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    qa = torch.zeros((T, H * D), dtype=torch.float16, device=DEV)
    qal = torch.zeros_like(qa)
    n.rope_append(qkv, q_len * W, W, qa, q_len * H * D, H * D, qkv[:, H * D:], qkv[:, (H + Hkv) * D:], q_len * W, W,
                  arena_a[:, 0], arena_a[:, 1], 2 * Hkv * cap * D, cap * D, cs, B, H, Hkv, D, q_len, past, cap, True, q_out_lo=qal)
    perm = n.qkv_rope_row_perm(H + 2 * Hkv, D).to(DEV)
    arena_b = torch.zeros_like(arena_a)
    qb = torch.zeros_like(qa)
    qbl = torch.zeros_like(qa)
    n.gemm_qkv_rope(n.to_weight_frags(w[perm].contiguous()), hi, lo, T, hid, cs, qb, qbl, H * D, arena_b[:, 0], arena_b[:, 1],
                    2 * Hkv * cap * D, cap * D, B, H, Hkv, D, q_len, past, cap)
    torch.cuda.synchronize()
    ref = x.double() @ w.double().t()
    assert (qkv.double() - ref).abs().max().item() < 2e-4 * float(ref.abs().max())
    for a, b in ((qa, qb), (arena_a, arena_b)):
        d = (a.float() - b.float()).abs()
        assert float(d.max()) <= 2e-3 * max(1.0, float(a.float().abs().max()))
        assert float((a != b).float().mean()) < 0.02
    assert torch.equal(arena_a[:, 1], arena_b[:, 1])
    assert float(arena_b[:, :, :, past:past + q_len].abs().sum()) > 0 and float(arena_b[:, :, :, :past].abs().sum()) == 0


@pytest.mark.parametrize("B,H,Hkv,q_len,past,hid,ks", [(1, 40, 40, 259, 30, 5120, 2), (2, 4, 2, 70, 7, 512, 2), (1, 8, 8, 100, 0, 1024, 3)])
def test_gemm_rows_qkv_rope_with_k_slices_equals_the_single_launch(B, H, Hkv, q_len, past, hid, ks):
    """pc_gemm's q|k|v epilogue with K slices (65..288 rows: wide panels, partial slabs, qkv_rope_slabs_kernel) against the same
    call without slices: same q planes, same arena rows and residual rows up to the fp32 order of the K sum."""
    n = _n()
    rng = np.random.default_rng(29)
    D = 128
    T, W = B * q_len, (H + 2 * Hkv) * D
    cap = past + q_len + 2
    w = torch.from_numpy((0.05 * rng.standard_normal((W, hid), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((T, hid), dtype=np.float32)).to(DEV)
    hi, lo = n.to_act_frags(x)
    pos = rng.integers(0, 3000, size=T).astype(np.int32)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(torch.from_numpy(pos).to(DEV), _inv_freq(D, 10000.0).to(DEV), cs, T, D)
    wf = n.to_weight_frags(w[n.qkv_rope_row_perm(H + 2 * Hkv, D).to(DEV)].contiguous())
    outs = []
    for slices in (1, ks):
        arena = torch.full((B, 2, Hkv, cap, D), 5.0, dtype=torch.float16, device=DEV)
        klo = torch.full((B, 2, Hkv, q_len, D), 5.0, dtype=torch.float16, device=DEV)
        q, ql = torch.zeros((T, H * D), dtype=torch.float16, device=DEV), torch.zeros((T, H * D), dtype=torch.float16, device=DEV)
        scratch = torch.empty((slices, T, W), dtype=torch.float32, device=DEV) if slices > 1 else None
        n.gemm_qkv_rope(wf, hi, lo, T, hid, cs, q, ql, H * D, arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D, B, H, Hkv, D,
                        q_len, past, cap, kv_lo=(klo[:, 0], klo[:, 1], 2 * Hkv * q_len * D, q_len * D), lo_base=-1,
                        kslices=slices, scratch=scratch)
        torch.cuda.synchronize()
        outs.append((q.double() + ql.double(), arena[:, :, :, past:past + q_len].double() + klo.double(), arena.clone()))
    (q1, kv1, a1), (q2, kv2, a2) = outs
    scale = max(1.0, float(q1.abs().max()), float(kv1.abs().max()))
    assert (q1 - q2).abs().max().item() < 2e-6 * scale and (kv1 - kv2).abs().max().item() < 2e-6 * scale
    assert torch.equal(a1[:, :, :, :past], a2[:, :, :, :past]) and torch.equal(a1[:, :, :, past + q_len:], a2[:, :, :, past + q_len:])
    with pytest.raises(RuntimeError, match="ks_scratch"):
        n.gemm_qkv_rope(wf, hi, lo, T, hid, cs, q, ql, H * D, arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D, B, H, Hkv, D,
                        q_len, past, cap, kslices=ks)


# ---------------------------------------------------------------------------------------------------
# RMSNorm folded into the projection (M <= 32: one or two row tiles)
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("M,N,K", [(12, 12288, 4096), (1, 4096, 4096), (16, 32000, 4096), (5, 48, 32), (14, 15360, 5120), (3, 64, 96),
                                   (17, 4096, 4096), (26, 32000, 4096), (32, 12288, 4096), (20, 48, 32), (29, 15360, 5120), (31, 64, 96)])
def test_gemm_skinny_norm_store(M, N, K):
    n = _n()
    rng = np.random.default_rng(31)
    w = torch.from_numpy((0.05 * rng.standard_normal((N, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy((3.0 * rng.standard_normal((M, K), dtype=np.float32))).to(DEV)
    gam = torch.from_numpy((1.0 + 0.2 * rng.standard_normal(K, dtype=np.float32)).astype(np.float16)).to(DEV)
    eps = 1e-5
    y = torch.full((M, N), 7.0, dtype=torch.float32, device=DEV)
    n.gemm_skinny_norm(n.to_weight_frags(w), x, gam, eps, M, N, K, n.EPI_STORE, y=y, ldy=N)
    xd = x.double()
    xn = xd * torch.rsqrt((xd * xd).mean(dim=1, keepdim=True) + eps) * gam.double()
    ref = (xn @ w.double().t()).float()
    assert (y - ref).abs().max().item() < 2e-4 * float(ref.abs().max()) + 1e-5
    # and against the two-launch path (pc_rmsnorm_frag + pc_gemm_skinny): same math, fp32-level agreement
    mt = (M + 15) // 16
    hi = torch.zeros((mt, K // 32, 64, 8), dtype=torch.float16, device=DEV)
    lo = torch.zeros_like(hi)
    n.rmsnorm_frag(x.clone(), gam, hi, lo, M, K, eps)
    y2 = torch.empty_like(y)
    n.gemm_skinny(n.to_weight_frags(w), hi, lo, M, N, K, n.EPI_STORE, y=y2, ldy=N)
    assert (y - y2).abs().max().item() < 1e-4 * float(ref.abs().max()) + 1e-5
    y3 = torch.empty_like(y)
    n.gemm_skinny_norm(n.to_weight_frags(w), x, gam, eps, M, N, K, n.EPI_STORE, y=y3, ldy=N)
    assert torch.equal(y, y3)


@pytest.mark.parametrize("M,inter,K", [(12, 11008, 4096), (3, 64, 32), (16, 13824, 5120), (1, 1376, 512),
                                       (26, 11008, 4096), (17, 64, 32), (32, 13824, 5120), (21, 1376, 512)])
def test_gemm_skinny_norm_silu(M, inter, K):
    n = _n()
    rng = np.random.default_rng(32)
    w = torch.from_numpy((0.05 * rng.standard_normal((2 * inter, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy((2.0 * rng.standard_normal((M, K), dtype=np.float32))).to(DEV)
    gam = torch.from_numpy((1.0 + 0.2 * rng.standard_normal(K, dtype=np.float32)).astype(np.float16)).to(DEV)
    eps = 1e-6
    oh = torch.zeros(((M + 15) // 16, inter // 32, 64, 8), dtype=torch.float16, device=DEV)
    ol = torch.zeros_like(oh)
    n.gemm_skinny_norm(n.to_weight_frags(w), x, gam, eps, M, 2 * inter, K, n.EPI_SILU, of_hi=oh, of_lo=ol)
    xd = x.double()
    xn = xd * torch.rsqrt((xd * xd).mean(dim=1, keepdim=True) + eps) * gam.double()
    gu = xn @ w.double().t()
    g, u = gu[:, :inter], gu[:, inter:]
    ref = (g / (1 + torch.exp(-g)) * u).float()
    got = n.from_act_frags(oh, M).float() + n.from_act_frags(ol, M).float()
    assert (got - ref).abs().max().item() < 2e-4 * float(ref.abs().max()) + 1e-5


@pytest.mark.parametrize("B,H,Hkv,D,q_len,past,hid", [(1, 32, 32, 128, 12, 100, 4096), (2, 4, 2, 128, 5, 7, 512), (1, 2, 2, 64, 1, 30, 256),
                                                       (1, 32, 32, 128, 26, 100, 4096), (2, 4, 2, 128, 11, 7, 512), (1, 2, 2, 64, 32, 30, 256),
                                                       (1, 40, 40, 128, 17, 9, 5120)])
def test_gemm_qkv_rope_norm_equals_two_launches(B, H, Hkv, D, q_len, past, hid):
    n = _n()
    rng = np.random.default_rng(33)
    T = B * q_len
    W = (H + 2 * Hkv) * D
    cap = past + q_len + 2
    w = torch.from_numpy((0.05 * rng.standard_normal((W, hid), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy((2.0 * rng.standard_normal((T, hid), dtype=np.float32))).to(DEV)
    gam = torch.from_numpy((1.0 + 0.2 * rng.standard_normal(hid, dtype=np.float32)).astype(np.float16)).to(DEV)
    eps = 1e-5
    pos = rng.integers(0, 3000, size=T).astype(np.int32)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(torch.from_numpy(pos).to(DEV), _inv_freq(D, 10000.0).to(DEV), cs, T, D)
    perm = n.qkv_rope_row_perm(H + 2 * Hkv, D).to(DEV)
    wf = n.to_weight_frags(w[perm].contiguous())
    mt = (T + 15) // 16
    hi = torch.zeros((mt, hid // 32, 64, 8), dtype=torch.float16, device=DEV)
    lo = torch.zeros_like(hi)
    n.rmsnorm_frag(x.clone(), gam, hi, lo, T, hid, eps)
    arena_a = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    qa = torch.zeros((T, H * D), dtype=torch.float16, device=DEV); qal = torch.zeros_like(qa)
    n.gemm_qkv_rope(wf, hi, lo, T, hid, cs, qa, qal, H * D, arena_a[:, 0], arena_a[:, 1], 2 * Hkv * cap * D, cap * D,
                    B, H, Hkv, D, q_len, past, cap)
    arena_b = torch.zeros_like(arena_a)
    qb = torch.zeros_like(qa); qbl = torch.zeros_like(qa)
    n.gemm_qkv_rope_norm(wf, x, gam, eps, T, hid, cs, qb, qbl, H * D, arena_b[:, 0], arena_b[:, 1], 2 * Hkv * cap * D, cap * D,
                         B, H, Hkv, D, q_len, past, cap)
    torch.cuda.synchronize()
    for a, b in ((qa, qb), (arena_a, arena_b)):
        d = (a.float() - b.float()).abs()
        assert float(d.max()) <= 2e-3 * max(1.0, float(a.float().abs().max()))      # <= 1 fp16 ulp
        assert float((a != b).float().mean()) < 0.02
    rec_a, rec_b = qa.float() + qal.float(), qb.float() + qbl.float()
    assert float((rec_a - rec_b).abs().max()) < 2e-5 * max(1.0, float(rec_a.abs().max()))


# ---------------------------------------------------------------------------------------------------
# Falcon ops: LayerNorm (dense and fragment planes with slab folding), GELU (dense and GEMM epilogue)
# ---------------------------------------------------------------------------------------------------

def _ln_ref(x, w, b, eps):
    xd = x.double()
    mu = xd.mean(dim=1, keepdim=True)
    var = ((xd - mu) ** 2).mean(dim=1, keepdim=True)
    return (xd - mu) * torch.rsqrt(var + eps) * w.double() + b.double()


@pytest.mark.parametrize("rows,hidden", [(12, 4544), (1, 128), (70, 448), (33, 5120)])
@pytest.mark.parametrize("nslabs", [0, 8])
def test_layernorm_dense_and_frag(rows, hidden, nslabs):
    n = _n()
    rng = np.random.default_rng(41)
    x = torch.from_numpy((2.0 * rng.standard_normal((rows, hidden), dtype=np.float32) + 0.7)).to(DEV)
    w = torch.from_numpy((1.0 + 0.2 * rng.standard_normal(hidden, dtype=np.float32)).astype(np.float16)).to(DEV)
    b = torch.from_numpy((0.3 * rng.standard_normal(hidden, dtype=np.float32)).astype(np.float16)).to(DEV)
    eps = 1e-5
    slabs = torch.from_numpy(rng.standard_normal((max(nslabs, 1), rows, hidden), dtype=np.float32)).to(DEV)
    xsum = x + (slabs[:nslabs].sum(dim=0) if nslabs else 0)
    ref = _ln_ref(xsum, w.float(), b.float(), eps).float()
    mt = (rows + 15) // 16
    hi = torch.zeros((mt, hidden // 32, 64, 8), dtype=torch.float16, device=DEV)
    lo = torch.zeros_like(hi)
    xw = x.clone()
    n.layernorm_frag(xw, w, b, hi, lo, rows, hidden, eps, slabs if nslabs else None, nslabs)
    got = n.from_act_frags(hi, rows).float() + n.from_act_frags(lo, rows).float()
    assert (got - ref).abs().max().item() < 2e-5 * max(1.0, float(ref.abs().max()))
    if nslabs:
        assert (xw - xsum).abs().max().item() < 1e-5 * max(1.0, float(xsum.abs().max()))    # folded in place
    out = torch.empty((rows, hidden), dtype=torch.float16, device=DEV)
    n.layernorm(xsum.contiguous(), w, b, out, rows, hidden, eps)
    assert (out.float() - ref).abs().max().item() < 2e-3 * max(1.0, float(ref.abs().max()))     # one fp16 rounding


def test_gelu_dense():
    n = _n()
    x = torch.linspace(-8, 8, 4096 * 8, device=DEV).view(8, -1).contiguous()
    out = torch.empty_like(x, dtype=torch.float16)
    n.gelu(x, out, x.numel())
    ref = torch.nn.functional.gelu(x.double()).float()
    assert (out.float() - ref).abs().max().item() < 4e-3


@pytest.mark.parametrize("M,N,K", [(12, 18176, 4544), (3, 64, 32), (40, 1792, 448), (100, 18176, 4544), (259, 512, 128)])
def test_gemm_gelu_epilogue(M, N, K):
    n = _n()
    rng = np.random.default_rng(42)
    w = torch.from_numpy((0.05 * rng.standard_normal((N, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    hi, lo = n.to_act_frags(x)
    mt = (M + 15) // 16
    oh = torch.zeros((mt, N // 32, 64, 8), dtype=torch.float16, device=DEV)
    ol = torch.zeros_like(oh)
    n.gemm_skinny(n.to_weight_frags(w), hi, lo, M, N, K, n.EPI_GELU, of_hi=oh, of_lo=ol)
    ref = torch.nn.functional.gelu(x.double() @ w.double().t()).float()
    got = n.from_act_frags(oh, M).float() + n.from_act_frags(ol, M).float()
    assert (got - ref).abs().max().item() < 2e-4 * float(ref.abs().max()) + 1e-5


# ---------------------------------------------------------------------------------------------------
# ALiBi attention (MPT): score += slope[h] * position_id[key]
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("B,H,D,q_len,past", [(1, 4, 32, 12, 100), (1, 8, 64, 1, 77), (2, 4, 64, 30, 5), (1, 32, 128, 12, 1725),
                                              (1, 8, 64, 200, 300), (1, 4, 128, 70, 0)])
def test_attn_alibi_matches_reference_math(B, H, D, q_len, past):
    n = _n()
    rng = np.random.default_rng(51)
    kv = past + q_len
    cap = kv + 3
    q = torch.from_numpy(rng.standard_normal((B, q_len, H, D), dtype=np.float32)).to(DEV)
    k = torch.from_numpy(rng.standard_normal((B, H, cap, D), dtype=np.float32)).to(DEV).half()
    v = torch.from_numpy(rng.standard_normal((B, H, cap, D), dtype=np.float32)).to(DEV).half()
    pos = torch.from_numpy(np.sort(rng.choice(4000, size=(B, kv), replace=True), axis=1).astype(np.int64)).to(DEV)
    n2 = 2 ** int(np.ceil(np.log2(H)))
    slopes = 1.0 / torch.pow(2, torch.arange(1, n2 + 1, dtype=torch.float32) * (8.0 / n2))
    slopes = slopes[:H].to(DEV)
    cols = (cap + 63) // 64 * 64 + 64
    kpos = torch.zeros((B, cols), dtype=torch.float32, device=DEV)
    kpos[:, :kv] = pos.float()
    q16 = q.half().contiguous()
    out = torch.empty((B, q_len, H * D), dtype=torch.float16, device=DEV)
    ws_bytes = n.attn_workspace_bytes(B, H, D, q_len, kv)
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=DEV)
    n.attn_fwd(q16, q_len * H * D, H * D, k, v, H * cap * D, cap * D, out, q_len * H * D, H * D, B, H, H, D, q_len, past,
               1.0 / np.sqrt(D), ws, alibi=(kpos, slopes * 1.4426950408889634))
    # reference math (mpt.py:157-184) in fp64 on the fp16-rounded operands
    qd, kd, vd = q16.double().permute(0, 2, 1, 3), k[:, :, :kv].double(), v[:, :, :kv].double()
    s = qd @ kd.transpose(2, 3) / np.sqrt(D)
    s = s + slopes.double()[None, :, None, None] * (pos.double()[:, None, None, :] - pos.max().double())
    idx = torch.arange(q_len, device=DEV)
    mask = torch.ones((q_len, kv), dtype=torch.bool, device=DEV)
    mask[:, past:] = idx[None, :] <= idx[:, None]
    s = s.masked_fill(~mask[None, None], float("-inf"))
    ref = (torch.softmax(s, dim=-1) @ vd).permute(0, 2, 1, 3).reshape(B, q_len, H * D)
    err = (out.double() - ref).abs().max().item()
    assert err < (3e-3 if q_len > 64 else 2e-3), err


# ---------------------------------------------------------------------------------------------------
# split-precision many-row path: elementwise ops with (hi, lo) outputs, K/V residual planes in the attention
# ---------------------------------------------------------------------------------------------------

def test_split_elementwise_ops():
    n = _n()
    rng = np.random.default_rng(61)
    T, hid, inter = 37, 512, 1376
    x = torch.from_numpy((2.0 * rng.standard_normal((T, hid), dtype=np.float32) + 0.3)).to(DEV)
    w = torch.from_numpy((1.0 + 0.2 * rng.standard_normal(hid, dtype=np.float32)).astype(np.float16)).to(DEV)
    b = torch.from_numpy((0.3 * rng.standard_normal(hid, dtype=np.float32)).astype(np.float16)).to(DEV)
    hi = torch.empty((T, hid), dtype=torch.float16, device=DEV); lo = torch.empty_like(hi)
    n.rmsnorm_split(x, w, hi, lo, T, hid, 1e-5)
    xd = x.double()
    ref = xd * torch.rsqrt((xd * xd).mean(1, keepdim=True) + 1e-5) * w.double()
    assert ((hi.double() + lo.double()) - ref).abs().max().item() < 2e-6 * float(ref.abs().max())
    assert (hi.double() - ref).abs().max().item() > 1e-4           # the hi plane alone is only fp16-accurate
    n.layernorm_split(x, w, b, hi, lo, T, hid, 1e-5)
    assert ((hi.double() + lo.double()) - _ln_ref(x, w.float(), b.float(), 1e-5)).abs().max().item() < 5e-6 * 5
    # silu(g)*u of the sum of two addends
    a1 = torch.from_numpy(rng.standard_normal((T, 2 * inter), dtype=np.float32)).to(DEV)
    a2 = torch.from_numpy((1e-3 * rng.standard_normal((T, 2 * inter), dtype=np.float32))).to(DEV)
    oh = torch.empty((T, inter), dtype=torch.float16, device=DEV); ol = torch.empty_like(oh)
    n.silu_mul_split(a1, a2, oh, ol, T, inter)
    gsum = (a1 + a2).double()
    ref = torch.nn.functional.silu(gsum[:, :inter]) * gsum[:, inter:]
    assert ((oh.double() + ol.double()) - ref).abs().max().item() < 3e-6 * max(1.0, float(ref.abs().max()))
    gh = torch.empty((T, 2 * inter), dtype=torch.float16, device=DEV); gl = torch.empty_like(gh)
    n.gelu_split(a1, a2, gh, gl, T * 2 * inter)
    ref = torch.nn.functional.gelu(gsum)
    assert ((gh.double() + gl.double()) - ref).abs().max().item() < 3e-6 * max(1.0, float(ref.abs().max()))
    y = x.clone()
    p1, p2 = torch.randn((T, hid), device=DEV), torch.randn((T, hid), device=DEV)
    n.add3(y, p1, p2, T * hid)
    assert torch.allclose(y, x + p1 + p2, atol=1e-6)


@pytest.mark.parametrize("B,H,Hkv,D,q_len,past", [(1, 4, 4, 128, 300, 0), (1, 4, 2, 128, 100, 70), (2, 2, 2, 64, 70, 33), (1, 8, 8, 128, 16, 200)])
def test_attn_split_precision_qkv(B, H, Hkv, D, q_len, past):
    """pc_rope_append_ex + pc_attn_fwd_ex: Q, P and the pass's own K/V rows in split precision.  Against fp64 attention
    over the un-rounded new K/V (staged rows are fp16 as the reference stages them) the error must be far below what
    fp16 K/V of the new rows costs."""
    n = _n()
    rng = np.random.default_rng(62)
    T, cap, Wd = B * q_len, past + q_len + 5, (H + 2 * Hkv) * D
    qkv = torch.from_numpy(rng.standard_normal((T, Wd), dtype=np.float32)).to(DEV)
    arena = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    arena[:, :, :, :past] = torch.from_numpy(rng.standard_normal((B, 2, Hkv, past, D), dtype=np.float32)).to(DEV).half()
    pos = torch.from_numpy(np.tile(np.arange(past, past + q_len, dtype=np.int32), B)).to(DEV)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(pos, _inv_freq(D, 10000.0).to(DEV), cs, T, D)
    q16 = torch.empty((T, H * D), dtype=torch.float16, device=DEV); q16l = torch.empty_like(q16)
    klo = torch.empty((B, Hkv, q_len, D), dtype=torch.float16, device=DEV); vlo = torch.empty_like(klo)
    lo = (klo, vlo, Hkv * q_len * D, q_len * D, past)
    n.rope_append(qkv, q_len * Wd, Wd, q16, q_len * H * D, H * D, qkv[:, H * D:], qkv[:, (H + Hkv) * D:], q_len * Wd, Wd,
                  arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D, cs, B, H, Hkv, D, q_len, past, cap, True,
                  q_out_lo=q16l, kv_lo=lo)
    out = torch.empty((B, q_len, H * D), dtype=torch.float16, device=DEV); out_lo = torch.empty_like(out)
    ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, past + q_len), 4) // 4, dtype=torch.float32, device=DEV)
    n.attn_fwd(q16, q_len * H * D, H * D, arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D, out, q_len * H * D, H * D,
               B, H, Hkv, D, q_len, past, 1.0 / np.sqrt(D), ws, q_lo=q16l, out_lo=out_lo, kv_lo=lo)
    # fp64 reference on the exact rotated q / new k / new v
    def rot(t):   # [T, heads, D]
        c, s_ = cs[..., 0].double(), cs[..., 1].double()
        c, s_ = torch.cat([c, c], 1)[:, None], torch.cat([s_, s_], 1)[:, None]
        return t * c + torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1) * s_
    qd = rot(qkv[:, :H * D].double().view(T, H, D)).view(B, q_len, H, D).permute(0, 2, 1, 3)
    kn = rot(qkv[:, H * D:(H + Hkv) * D].double().view(T, Hkv, D)).view(B, q_len, Hkv, D).permute(0, 2, 1, 3)
    vn = qkv[:, (H + Hkv) * D:].double().view(B, q_len, Hkv, D).permute(0, 2, 1, 3)
    K = torch.cat([arena[:, 0, :, :past].double(), kn], 2).repeat_interleave(H // Hkv, 1)
    V = torch.cat([arena[:, 1, :, :past].double(), vn], 2).repeat_interleave(H // Hkv, 1)
    s = qd @ K.transpose(2, 3) / np.sqrt(D)
    idx = torch.arange(q_len, device=DEV)
    mask = torch.ones((q_len, past + q_len), dtype=torch.bool, device=DEV)
    mask[:, past:] = idx[None, :] <= idx[:, None]
    ref = (torch.softmax(s.masked_fill(~mask, float("-inf")), -1) @ V).permute(0, 2, 1, 3).reshape(B, q_len, H * D)
    err = ((out.double() + out_lo.double()) - ref).abs().max().item()
    # what rounding the new K/V to fp16 alone would cost
    Kh = torch.cat([arena[:, 0, :, :past].double(), kn.half().double()], 2).repeat_interleave(H // Hkv, 1)
    Vh = torch.cat([arena[:, 1, :, :past].double(), vn.half().double()], 2).repeat_interleave(H // Hkv, 1)
    s2 = qd @ Kh.transpose(2, 3) / np.sqrt(D)
    ref16 = (torch.softmax(s2.masked_fill(~mask, float("-inf")), -1) @ Vh).permute(0, 2, 1, 3).reshape(B, q_len, H * D)
    floor16 = (ref16 - ref).abs().max().item()
    assert err < 3e-5 and err < 0.25 * floor16, (err, floor16)
    # the arena itself holds the fp16 value (what gets stored / staged)
    assert torch.equal(arena[:, 0, :, past:past + q_len], kn.half()[:, :, :, :].to(arena.dtype)) or \
        (arena[:, 0, :, past:past + q_len].double() - kn).abs().max().item() < 4e-3


@pytest.mark.parametrize("B,H,Hkv,D,q_len,past,hid", [(1, 32, 32, 128, 12, 100, 4096), (2, 4, 2, 128, 5, 70, 512), (1, 8, 1, 64, 40, 130, 512),
                                                       (1, 4, 4, 128, 104, 90, 512), (1, 32, 32, 128, 26, 300, 4096), (2, 8, 1, 64, 31, 5, 512),
                                                       (1, 4, 4, 32, 32, 0, 128), (1, 4, 2, 128, 17, 1000, 512), (1, 4, 4, 128, 16, 64, 512)])
def test_weight_streaming_prefill_keeps_new_kv_residuals(B, H, Hkv, D, q_len, past, hid):
    """pc_gemm_qkv_rope(k_lo, v_lo) + pc_attn_fwd_ex(lo_row0 = -1, device past_len, fragment output): the K / V rows the
    pass appends reach its own attention in split precision.  (a) arena + lo planes reproduce the fp32 projection;
    (b) the attention output is far closer to fp64 attention over the un-rounded new rows than fp16 new rows allow."""
    n = _n()
    rng = np.random.default_rng(77)
    T, W, cap = B * q_len, (H + 2 * Hkv) * D, past + q_len + 3
    w = torch.from_numpy((0.05 * rng.standard_normal((W, hid), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((T, hid), dtype=np.float32)).to(DEV)
    hi, lo = n.to_act_frags(x)
    pos = torch.from_numpy(np.tile(np.arange(past, past + q_len, dtype=np.int32), B)).to(DEV)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(pos, _inv_freq(D, 10000.0).to(DEV), cs, T, D)
    arena = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    arena[:, :, :, :past] = torch.from_numpy(rng.standard_normal((B, 2, Hkv, past, D), dtype=np.float32)).to(DEV).half()
    perm = n.qkv_rope_row_perm(H + 2 * Hkv, D).to(DEV)
    q16 = torch.empty((T, H * D), dtype=torch.float16, device=DEV); q16l = torch.empty_like(q16)
    klo = torch.full((B, Hkv, q_len, D), 7.0, dtype=torch.float16, device=DEV); vlo = torch.full_like(klo, 7.0)
    past_dev = torch.tensor([past], dtype=torch.int32, device=DEV)
    # host past_len deliberately wrong (0): the kernels must take it from the device word
    n.gemm_qkv_rope(n.to_weight_frags(w[perm].contiguous()), hi, lo, T, hid, cs, q16, q16l, H * D, arena[:, 0], arena[:, 1],
                    2 * Hkv * cap * D, cap * D, B, H, Hkv, D, q_len, 0, cap, past_dev,
                    kv_lo=(klo, vlo, Hkv * q_len * D, q_len * D))
    # fp64 projection + rotation of the same operands
    xd = (n.from_act_frags(hi, T).double() + n.from_act_frags(lo, T).double())
    qkv = xd @ w.double().T
    def rot(t):
        c, s_ = cs[..., 0].double(), cs[..., 1].double()
        c, s_ = torch.cat([c, c], 1)[:, None], torch.cat([s_, s_], 1)[:, None]
        return t * c + torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1) * s_
    kn = rot(qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)).view(B, q_len, Hkv, D).permute(0, 2, 1, 3)
    vn = qkv[:, (H + Hkv) * D:].view(B, q_len, Hkv, D).permute(0, 2, 1, 3)
    k_rec = arena[:, 0, :, past:past + q_len].double() + klo.double()
    v_rec = arena[:, 1, :, past:past + q_len].double() + vlo.double()
    scale_k, scale_v = float(kn.abs().max()), float(vn.abs().max())
    assert (k_rec - kn).abs().max().item() < 2e-6 * max(1.0, scale_k)
    assert (v_rec - vn).abs().max().item() < 2e-6 * max(1.0, scale_v)
    assert (arena[:, 0, :, past:past + q_len].double() - kn).abs().max().item() < 1e-3 * max(1.0, scale_k)   # arena = fp16(value)
    # attention: fragment-plane output, lo_row0 = -1
    mt = (T + 15) // 16
    fh = torch.zeros((mt, H * D // 32, 64, 8), dtype=torch.float16, device=DEV); fl = torch.zeros_like(fh)
    ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, past + q_len), 4) // 4, dtype=torch.float32, device=DEV)
    n.attn_fwd(q16, q_len * H * D, H * D, arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D, None, 0, 0,
               B, H, Hkv, D, q_len, past, 1.0 / np.sqrt(D), ws, past_len_dev=past_dev, out_frag=(fh, fl), q_lo=q16l,
               kv_lo=(klo, vlo, Hkv * q_len * D, q_len * D, -1))
    got = (n.from_act_frags(fh, T).double() + n.from_act_frags(fl, T).double()).view(B, q_len, H * D)
    qd = rot(qkv[:, :H * D].view(T, H, D)).view(B, q_len, H, D).permute(0, 2, 1, 3)
    idx = torch.arange(q_len, device=DEV)
    mask = torch.ones((q_len, past + q_len), dtype=torch.bool, device=DEV)
    mask[:, past:] = idx[None, :] <= idx[:, None]
    def attn(kx, vx):
        K = torch.cat([arena[:, 0, :, :past].double(), kx], 2).repeat_interleave(H // Hkv, 1)
        V = torch.cat([arena[:, 1, :, :past].double(), vx], 2).repeat_interleave(H // Hkv, 1)
        s = qd @ K.transpose(2, 3) / np.sqrt(D)
        return (torch.softmax(s.masked_fill(~mask, float("-inf")), -1) @ V).permute(0, 2, 1, 3).reshape(B, q_len, H * D)
    ref = attn(kn, vn)
    floor16 = (attn(kn.half().double(), vn.half().double()) - ref).abs().max().item()
    err = (got - ref).abs().max().item()
    assert err < 3e-5 * max(1.0, scale_v) and err < 0.25 * floor16, (err, floor16)


def test_rope_append_sums_the_two_halves_of_a_stacked_projection():
    """pc_rope_append_ex(in2_offset): inputs are x[i] + x[i + in2_offset] -- identical to adding the halves first."""
    n = _n()
    rng = np.random.default_rng(91)
    B, H, Hkv, D, q_len, past = 2, 4, 2, 64, 37, 11
    T, W, cap = B * q_len, (H + 2 * Hkv) * D, past + q_len + 3
    x2 = torch.from_numpy(rng.standard_normal((2 * T, W), dtype=np.float32)).to(DEV)
    xs = (x2[:T] + x2[T:]).contiguous()
    pos = torch.from_numpy(rng.integers(0, 4000, size=T).astype(np.int32)).to(DEV)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(pos, _inv_freq(D, 10000.0).to(DEV), cs, T, D)
    outs = []
    for src, off in ((xs, 0), (x2, T * W)):
        arena = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
        q16 = torch.zeros((T, H * D), dtype=torch.float16, device=DEV); q16l = torch.zeros_like(q16)
        klo = torch.zeros((B, Hkv, q_len, D), dtype=torch.float16, device=DEV); vlo = torch.zeros_like(klo)
        n.rope_append(src, q_len * W, W, q16, q_len * H * D, H * D, src[:, H * D:], src[:, (H + Hkv) * D:], q_len * W, W,
                      arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D, cs, B, H, Hkv, D, q_len, past, cap, True,
                      q_out_lo=q16l, kv_lo=(klo, vlo, Hkv * q_len * D, q_len * D, past), in2_offset=off)
        outs.append((arena, q16, q16l, klo, vlo))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------
# int8 weights (weight-only; pc_gemm_*_w8)
# ---------------------------------------------------------------------------------------------------

def test_int8_row_quantizer_is_bit_exact_with_the_oracle():
    from oracle import int8_oracle as io
    n = _n()
    rng = np.random.default_rng(5)
    w = (0.05 * rng.standard_normal((96, 192), dtype=np.float32)).astype(np.float16)
    w[7] = 0                                                    # an all-zero row keeps scale 1
    w[11, 3] = np.float16(0.31)                                 # an outlier sets its row's scale
    for r in range(20, 60):                                     # exact ties: w = amax / 2 -> 63.5 before rounding; a 1-ulp
        w[r, 5] = np.float16(-0.5) * np.abs(w[r]).max()         # error in 127 / amax flips the result (seen on real weights)
    q, s = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    qo, so = io.quantize_rows_int8(w.astype(np.float32))
    assert np.array_equal(q.cpu().numpy(), qo) and np.array_equal(s.cpu().numpy(), so)
    assert int(np.abs(qo).max()) == 127 and so[7] == 1.0
    img = n.to_weight_frags_i8(q).cpu().numpy()
    assert img.dtype == np.uint8 and img.shape == (6, 3, 4, 16, 2, 8)    # [tile][k-step pair][g][m][half][8]: lane = 16 g + m
    # lane g*16+m of tile t, k-step s holds row 16t+m, features 32s+8g .. +7, offset binary
    assert np.array_equal(img[2, 1, 1, 5, 1].astype(np.int16) - 128, qo[2 * 16 + 5, 3 * 32 + 8: 3 * 32 + 16].astype(np.int16))


@pytest.mark.parametrize("M,N,K,epi,kq", [(12, 4096, 4096, 0, 1), (12, 4096, 11008, 0, 4), (1, 12288, 4096, 0, 1), (40, 4096, 4096, 1, 1),
                                          (64, 512, 384, 0, 2), (16, 22016, 4096, 2, 1), (33, 1376 * 2, 512, 2, 1), (5, 2048, 512, 4, 1),
                                          (3, 64, 64, 0, 1), (20, 4096, 1408, 0, 4), (1, 4096, 11008, 0, 4)])
def test_gemm_skinny_w8_matches_fp64_on_dequantised_weights(M, N, K, epi, kq):
    """y = scale[n] * sum_k q[n][k] * (x_hi + x_lo)[k]: exact integers times split-precision activations, fp32 sums."""
    n = _n()
    rng = np.random.default_rng(17)
    w = torch.from_numpy((0.05 * rng.standard_normal((N, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    q, sc = n.quantize_rows_int8(w)
    wf8 = n.to_weight_frags_i8(q)
    wd = q.double() * sc.double()[:, None]
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    hi, lo = n.to_act_frags(x)
    xd = n.from_act_frags(hi, M).double() + n.from_act_frags(lo, M).double()
    ref = xd @ wd.T
    mt = (M + 15) // 16
    if epi in (0, 1):
        y = torch.full((kq, M, N), 0.5, dtype=torch.float32, device=DEV) if epi == 1 else torch.empty((kq, M, N), dtype=torch.float32, device=DEV)
        n.gemm_skinny(wf8, hi, lo, M, N, K, epi, y=y, ldy=N, kslices=kq, wscale=sc)
        got = y.double().sum(0)
        if epi == 1:
            ref = ref + 0.5
    else:
        inter = N // 2 if epi == 2 else N
        oh = torch.zeros((mt, inter // 32, 64, 8), dtype=torch.float16, device=DEV); ol = torch.zeros_like(oh)
        n.gemm_skinny(wf8, hi, lo, M, N, K, epi, of_hi=oh, of_lo=ol, wscale=sc)
        got = n.from_act_frags(oh, M).double() + n.from_act_frags(ol, M).double()
        ref = torch.nn.functional.silu(ref[:, :inter]) * ref[:, inter:] if epi == 2 else torch.nn.functional.gelu(ref)
    err = (got - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K,epi", [(12, 22016, 4096, 2), (7, 32000, 4096, 0), (16, 1024, 512, 0)])
def test_gemm_skinny_norm_w8(M, N, K, epi):
    n = _n()
    rng = np.random.default_rng(23)
    w = torch.from_numpy((0.05 * rng.standard_normal((N, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    q, sc = n.quantize_rows_int8(w)
    wd = q.double() * sc.double()[:, None]
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    gam = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(K, dtype=np.float32)).astype(np.float16)).to(DEV)
    xn = x.double() * torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-5) * gam.double()
    ref = xn @ wd.T
    if epi == 0:
        y = torch.empty((M, N), dtype=torch.float32, device=DEV)
        n.gemm_skinny_norm(n.to_weight_frags_i8(q), x, gam, 1e-5, M, N, K, 0, y=y, ldy=N, wscale=sc)
        got = y.double()
    else:
        inter = N // 2
        oh = torch.zeros((1, inter // 32, 64, 8), dtype=torch.float16, device=DEV); ol = torch.zeros_like(oh)
        n.gemm_skinny_norm(n.to_weight_frags_i8(q), x, gam, 1e-5, M, N, K, 2, of_hi=oh, of_lo=ol, wscale=sc)
        got = n.from_act_frags(oh, M).double() + n.from_act_frags(ol, M).double()
        ref = torch.nn.functional.silu(ref[:, :inter]) * ref[:, inter:]
    err = (got - ref).abs().max().item()
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("B,H,Hkv,D,q_len,past,hid,norm", [(1, 32, 32, 128, 12, 100, 4096, True), (1, 32, 32, 128, 40, 9, 4096, False),
                                                            (2, 4, 2, 64, 5, 7, 256, True)])
def test_gemm_qkv_rope_w8(B, H, Hkv, D, q_len, past, hid, norm):
    """The int8 fused q|k|v projection against the fp16 fused projection run on the DEQUANTISED weights -- those are
    exactly representable here only approximately (q * scale rounds in fp16), so the comparison is against fp64 math."""
    n = _n()
    rng = np.random.default_rng(29)
    T, W, cap = B * q_len, (H + 2 * Hkv) * D, past + q_len + 2
    w = torch.from_numpy((0.05 * rng.standard_normal((W, hid), dtype=np.float32)).astype(np.float16)).to(DEV)
    q8, sc = n.quantize_rows_int8(w)
    wd = q8.double() * sc.double()[:, None]
    perm = n.qkv_rope_row_perm(H + 2 * Hkv, D).to(DEV)
    x = torch.from_numpy(rng.standard_normal((T, hid), dtype=np.float32)).to(DEV)
    gam = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(hid, dtype=np.float32)).astype(np.float16)).to(DEV)
    pos = torch.from_numpy(rng.integers(0, 3000, size=T).astype(np.int32)).to(DEV)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(pos, _inv_freq(D, 10000.0).to(DEV), cs, T, D)
    arena = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    q16 = torch.zeros((T, H * D), dtype=torch.float16, device=DEV); q16l = torch.zeros_like(q16)
    klo = torch.zeros((B, Hkv, q_len, D), dtype=torch.float16, device=DEV); vlo = torch.zeros_like(klo)
    kv_lo = (klo, vlo, Hkv * q_len * D, q_len * D)
    wf8, scp = n.to_weight_frags_i8(q8[perm].contiguous()), sc[perm].contiguous()
    if norm:
        n.gemm_qkv_rope_norm(wf8, x, gam, 1e-5, T, hid, cs, q16, q16l, H * D, arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D,
                             B, H, Hkv, D, q_len, past, cap, kv_lo=kv_lo, wscale=scp)
        xd = x.double() * torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-5) * gam.double()
    else:
        hi, lo = n.to_act_frags(x)
        n.gemm_qkv_rope(wf8, hi, lo, T, hid, cs, q16, q16l, H * D, arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D,
                        B, H, Hkv, D, q_len, past, cap, kv_lo=kv_lo, wscale=scp)
        xd = n.from_act_frags(hi, T).double() + n.from_act_frags(lo, T).double()
    qkv = xd @ wd.T
    def rot(t):
        c, s_ = cs[..., 0].double(), cs[..., 1].double()
        c, s_ = torch.cat([c, c], 1)[:, None], torch.cat([s_, s_], 1)[:, None]
        return t * c + torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1) * s_
    qd = rot(qkv[:, :H * D].view(T, H, D)).reshape(T, H * D)
    kn = rot(qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)).view(B, q_len, Hkv, D).permute(0, 2, 1, 3)
    vn = qkv[:, (H + Hkv) * D:].view(B, q_len, Hkv, D).permute(0, 2, 1, 3)
    tol = 3e-5 * max(1.0, float(qkv.abs().max()))
    assert ((q16.double() + q16l.double()) - qd).abs().max().item() < tol
    assert ((arena[:, 0, :, past:past + q_len].double() + klo.double()) - kn).abs().max().item() < tol
    assert ((arena[:, 1, :, past:past + q_len].double() + vlo.double()) - vn).abs().max().item() < tol


def test_residual_tail_outlives_the_pass_for_decode():
    """pc_gemm_qkv_rope_ex(lo_base) + pc_attn_fwd_ex(lo_row0 = -2): a prefill writes tail rows [0, q), decode steps append
    one row each at past_len - base (read from past_len_dev[1]), and the decode attention sees every row since `base` in
    split precision: the result must beat what fp16 rows since `base` allow, against fp64 attention."""
    n = _n()
    rng = np.random.default_rng(101)
    B, H, Hkv, D, hid, base, q0, steps = 1, 8, 4, 128, 512, 70, 9, 5
    W, cap, tcap = (H + 2 * Hkv) * D, 128, 32
    w = torch.from_numpy((0.05 * rng.standard_normal((W, hid), dtype=np.float32)).astype(np.float16)).to(DEV)
    wf = n.to_weight_frags(w[n.qkv_rope_row_perm(H + 2 * Hkv, D).to(DEV)].contiguous())
    arena = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    arena[:, :, :, :base] = torch.from_numpy(rng.standard_normal((B, 2, Hkv, base, D), dtype=np.float32)).to(DEV).half()
    tail = torch.zeros((B, 2, Hkv, tcap, D), dtype=torch.float16, device=DEV)
    kv_lo = (tail[:, 0], tail[:, 1], 2 * Hkv * tcap * D, tcap * D)
    inv = _inv_freq(D, 10000.0).to(DEV)
    Kd = [arena[0, 0, :, :base].double()]; Vd = [arena[0, 1, :, :base].double()]          # fp64 truth per row block

    def rot(t, cs):
        c, s_ = cs[..., 0].double(), cs[..., 1].double()
        c, s_ = torch.cat([c, c], 1)[:, None], torch.cat([s_, s_], 1)[:, None]
        return t * c + torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1) * s_

    past = base
    for step in range(steps + 1):
        q_len = q0 if step == 0 else 1
        x = torch.from_numpy(rng.standard_normal((q_len, hid), dtype=np.float32)).to(DEV)
        hi, lo = n.to_act_frags(x)
        pos = torch.arange(past, past + q_len, dtype=torch.int32, device=DEV)
        cs = torch.empty((q_len, D // 2, 2), dtype=torch.float32, device=DEV)
        n.rope_table(pos, inv, cs, q_len, D)
        q16 = torch.zeros((q_len, H * D), dtype=torch.float16, device=DEV); q16l = torch.zeros_like(q16)
        pdev = torch.tensor([past, base], dtype=torch.int32, device=DEV)
        n.gemm_qkv_rope(wf, hi, lo, q_len, hid, cs, q16, q16l, H * D, arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D,
                        B, H, Hkv, D, q_len, 0, cap, pdev, kv_lo=kv_lo, lo_base=-1 if step == 0 else -2)
        xd = n.from_act_frags(hi, q_len).double() + n.from_act_frags(lo, q_len).double()
        qkv = xd @ w.double().T
        Kd.append(rot(qkv[:, H * D:(H + Hkv) * D].view(q_len, Hkv, D), cs).permute(1, 0, 2))
        Vd.append(qkv[:, (H + Hkv) * D:].view(q_len, Hkv, D).permute(1, 0, 2))
        qd = rot(qkv[:, :H * D].view(q_len, H, D), cs).permute(1, 0, 2)                    # [H, q, D]
        out = torch.zeros((1, q_len, H * D), dtype=torch.float16, device=DEV); out_lo = torch.zeros_like(out)
        ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, past + q_len), 4) // 4, dtype=torch.float32, device=DEV)
        n.attn_fwd(q16, q_len * H * D, H * D, arena[:, 0], arena[:, 1], 2 * Hkv * cap * D, cap * D, out, q_len * H * D, H * D,
                   B, H, Hkv, D, q_len, 0, 1.0 / np.sqrt(D), ws, past_len_dev=pdev, q_lo=q16l, out_lo=out_lo,
                   kv_lo=kv_lo + (-1 if step == 0 else -2,))
        K = torch.cat(Kd, 1).repeat_interleave(H // Hkv, 0); V = torch.cat(Vd, 1).repeat_interleave(H // Hkv, 0)
        idx = torch.arange(q_len, device=DEV)
        mask = torch.ones((q_len, past + q_len), dtype=torch.bool, device=DEV)
        mask[:, past:] = idx[None, :] <= idx[:, None]
        def attn(K_, V_):
            s_ = (qd @ K_.transpose(1, 2)) / np.sqrt(D)
            return (torch.softmax(s_.masked_fill(~mask, float("-inf")), -1) @ V_).permute(1, 0, 2).reshape(q_len, H * D)
        ref = attn(K, V)
        K16 = torch.cat([Kd[0]] + [k.half().double() for k in Kd[1:]], 1).repeat_interleave(H // Hkv, 0)
        V16 = torch.cat([Vd[0]] + [v.half().double() for v in Vd[1:]], 1).repeat_interleave(H // Hkv, 0)
        floor16 = (attn(K16, V16) - ref).abs().max().item()
        err = ((out[0].double() + out_lo[0].double()) - ref).abs().max().item()
        assert err < 3e-5 and err < 0.25 * floor16, (step, err, floor16)
        past += q_len
    # the tail holds one residual row per appended key, in key order
    rec = arena[0, 0, :, base:past].double() + tail[0, 0, :, :past - base].double()
    assert (rec - torch.cat(Kd[1:], 1)).abs().max().item() < 1e-5


@pytest.mark.parametrize("two", [True, False])
def test_ragged_past_rope_append_and_attention_match_oracle_row_by_row(two):
    """pc_rope_append_var + pc_attn_fwd_var: one past length per batch row (schema-encode suffix batches over trunk prefixes
    of different lengths).  Every batch row must equal the single-row call / the oracle at its own past length."""
    n = _n()
    rng = np.random.default_rng(21)
    B, H, Hkv, D, q_len = 3, 4, 2, 128, 70
    pasts = [5, 130, 64]
    cap = max(pasts) + q_len + 2
    W = (H + 2 * Hkv) * D
    qkv = torch.from_numpy(rng.standard_normal((B, q_len, W), dtype=np.float32)).to(DEV)
    arena = torch.from_numpy(rng.standard_normal((B, 2, Hkv, cap, D), dtype=np.float32).astype(np.float16)).to(DEV)
    lo = torch.zeros_like(arena)
    inv = _inv_freq(D, 10000.0).to(DEV)
    pos = torch.from_numpy(np.stack([np.arange(p, p + q_len) for p in pasts]).astype(np.int32)).to(DEV)
    cs = torch.empty((B * q_len, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(pos.reshape(-1), inv, cs, B * q_len, D)
    pl = torch.tensor(pasts, dtype=torch.int32, device=DEV)

    def run(arena_t, lo_t, rows, past_arg, past_lens):
        Bn = len(rows)
        q16 = torch.empty((Bn, q_len, H * D), dtype=torch.float16, device=DEV)
        q16l = torch.empty_like(q16) if two else None
        kp, vp = arena_t[:, 0], arena_t[:, 1]
        bs, hs = 2 * Hkv * cap * D, cap * D
        kvlo = (lo_t[:, 0], lo_t[:, 1], bs, hs, 0) if two else None
        x = qkv[rows].contiguous()
        c = cs.view(B, q_len, D // 2, 2)[rows].contiguous().view(Bn * q_len, D // 2, 2)
        n.rope_append(x, q_len * W, W, q16, q_len * H * D, H * D, x[:, :, H * D:], x[:, :, (H + Hkv) * D:], q_len * W, W,
                      kp, vp, bs, hs, c, Bn, H, Hkv, D, q_len, past_arg, cap, True, q_out_lo=q16l, kv_lo=kvlo, past_lens=past_lens)
        out = torch.full((Bn, q_len, H * D), float("nan"), dtype=torch.float16, device=DEV)
        out_lo = torch.empty_like(out) if two else None
        ws = torch.empty(max(n.attn_workspace_bytes(Bn, H, D, q_len, past_arg + q_len), 4) // 4, dtype=torch.float32, device=DEV)
        n.attn_fwd(q16, q_len * H * D, H * D, kp, vp, bs, hs, out, q_len * H * D, H * D, Bn, H, Hkv, D, q_len, past_arg,
                   1.0 / np.sqrt(D), ws, q_lo=q16l, out_lo=out_lo, kv_lo=kvlo, past_lens=past_lens)
        torch.cuda.synchronize()
        return out.float() + (out_lo.float() if two else 0.0), q16

    a1, l1 = arena.clone(), lo.clone()
    got, _ = run(a1, l1, [0, 1, 2], max(pasts), pl)
    assert torch.isfinite(got).all()
    for b, p in enumerate(pasts):
        a2, l2 = arena[b:b + 1].clone(), lo[b:b + 1].clone()
        one, q16 = run(a2, l2, [b], p, None)
        # same kernels, same data, a scalar past length: identical bits
        assert torch.equal(a1[b, :, :, :p + q_len], a2[0, :, :, :p + q_len])
        assert torch.equal(got[b], one[0])
        # rows past the row's own end were not touched
        assert torch.equal(a1[b, :, :, p + q_len:], arena[b, :, :, p + q_len:])
        # and the oracle on that row (K/V as appended, Q as rotated)
        kk = (a2[0, 0].float() + l2[0, 0].float())[None, :, :p + q_len]
        vv = (a2[0, 1].float() + l2[0, 1].float())[None, :, :p + q_len]
        qq = q16.float().view(1, q_len, H, D)
        ref = orc.attention_core(qq.cpu().numpy().transpose(0, 2, 1, 3), kk.cpu().numpy(), vv.cpu().numpy(), p, H // Hkv)
        ref = ref.transpose(0, 2, 1, 3).reshape(q_len, H * D)
        np.testing.assert_allclose(one[0].cpu().numpy(), ref, atol=4e-3 if not two else 2.5e-3, rtol=1e-2)


@pytest.mark.parametrize("prefix_lo,q_len", [(True, 70), (False, 70), (True, 9), (True, 200)])
def test_shared_prefix_attention_equals_the_prefix_copied_into_every_row(prefix_lo, q_len):
    """pc_attn with prefix_k / prefix_v (the suffix batches of a schema encode read the trunk's K/V in place): batch row b sees
    rows [0, past_lens[b]) of the SHARED planes and then its own rows, which its arena holds from row 0 on.  Must equal the
    launch over arenas that carry a copy of the prefix in every batch row (same keys, same order; the tiles are cut at the
    prefix end instead of every 64 keys, so the online softmax may round differently: fp32 round-off)."""
    n = _n()
    rng = np.random.default_rng(33)
    B, H, Hkv, D = 3, 4, 2, 128
    pres = [37, 300, 128]
    n_trunk = 320
    scale = 1.0 / np.sqrt(D)

    def f16(*shape):
        return torch.from_numpy((0.5 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float16)).to(DEV)

    trunk, trunk_lo = f16(2, Hkv, n_trunk, D), f16(2, Hkv, n_trunk, D) * 2.0 ** -11
    own, own_lo = f16(B, 2, Hkv, q_len, D), f16(B, 2, Hkv, q_len, D) * 2.0 ** -11
    q, q_lo = f16(B, q_len, H * D), f16(B, q_len, H * D) * 2.0 ** -11
    pl = torch.tensor(pres, dtype=torch.int32, device=DEV)
    ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, max(pres) + q_len), 4) // 4, dtype=torch.float32, device=DEV)

    def launch(k, v, bs, hs, kvlo, prefix):
        out = torch.full((B, q_len, H * D), float("nan"), dtype=torch.float16, device=DEV)
        out_lo = torch.empty_like(out)
        n.attn_fwd(q, q_len * H * D, H * D, k, v, bs, hs, out, q_len * H * D, H * D, B, H, Hkv, D, q_len, max(pres), scale, ws,
                   q_lo=q_lo, out_lo=out_lo, kv_lo=kvlo, past_lens=pl, prefix=prefix)
        torch.cuda.synchronize()
        return out.double() + out_lo.double()

    # in place: own arenas hold the pass's rows only
    got = launch(own[:, 0], own[:, 1], 2 * Hkv * q_len * D, q_len * D, (own_lo[:, 0], own_lo[:, 1], 2 * Hkv * q_len * D, q_len * D, 0),
                 (trunk[0], trunk[1], trunk_lo[0] if prefix_lo else None, trunk_lo[1] if prefix_lo else None, n_trunk * D))
    # copies: row b = trunk[:pre_b] ++ own[b]
    cap = max(pres) + q_len
    full, full_lo = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV), torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    for b, pre in enumerate(pres):
        full[b, :, :, :pre] = trunk[:, :, :pre]
        full[b, :, :, pre:pre + q_len] = own[b]
        if prefix_lo:
            full_lo[b, :, :, :pre] = trunk_lo[:, :, :pre]
        full_lo[b, :, :, pre:pre + q_len] = own_lo[b]
    want = launch(full[:, 0], full[:, 1], 2 * Hkv * cap * D, cap * D, (full_lo[:, 0], full_lo[:, 1], 2 * Hkv * cap * D, cap * D, 0), None)
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err <= 2e-6 * max(1.0, want.abs().max().item()), err
    # and the oracle, row by row, on the very keys
    for b, pre in enumerate(pres):
        kk = (full[b, 0].double() + full_lo[b, 0].double())[None, :, :pre + q_len].cpu().numpy()
        vv = (full[b, 1].double() + full_lo[b, 1].double())[None, :, :pre + q_len].cpu().numpy()
        qq = (q[b].double() + q_lo[b].double()).view(1, q_len, H, D).cpu().numpy().transpose(0, 2, 1, 3)
        ref = orc.attention_core(qq, kk, vv, pre, H // Hkv).transpose(0, 2, 1, 3).reshape(q_len, H * D)
        np.testing.assert_allclose(got[b].cpu().numpy(), ref, atol=3e-5, rtol=0)


@pytest.mark.parametrize("B,q_len,H,Hkv,two,ragged", [(1, 300, 4, 4, True, False), (3, 70, 4, 2, True, True), (2, 129, 3, 1, False, False),
                                                      (1, 5, 2, 2, True, False)])
def test_dense_qkv_rope_epilogue_equals_projection_then_rope_append(B, q_len, H, Hkv, two, ragged):
    """pc_gemm_dense_qkv_rope (RoPE + KV append in the epilogue of the many-row q|k|v projection, head_dim 128) against the two
    launches it replaces: pc_gemm_dense (fp32 [M][W]) + pc_rope_append.  Same fp32 accumulators, same rotation formula:
    identical bits for V and for every row the arena does not touch, fp32 round-off (an fma contraction) at most elsewhere."""
    n = _n()
    rng = np.random.default_rng(17 + q_len)
    D, K = 128, 256
    W, T = (H + 2 * Hkv) * D, B * q_len
    pasts = [9, 40, 0][:B] if ragged else [11] * B
    cap = max(pasts) + q_len + 3

    def f16(*shape, scale=1.0):
        return torch.from_numpy((scale * rng.standard_normal(shape, dtype=np.float32)).astype(np.float16)).to(DEV)

    x_hi, x_lo = f16(T, K), (f16(T, K, scale=2.0 ** -11) if two else None)
    w = f16(W, K, scale=0.08)
    pos = torch.from_numpy(np.concatenate([np.arange(p, p + q_len) * 3 + 1 for p in pasts]).astype(np.int32)).to(DEV)
    cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=DEV)
    n.rope_table(pos, _inv_freq(D, 10000.0).to(DEV), cs, T, D)
    pl = torch.tensor(pasts, dtype=torch.int32, device=DEV) if ragged else None
    bs, hs = 2 * Hkv * cap * D, cap * D

    def fresh():
        arena = torch.full((B, 2, Hkv, cap, D), 7.0, dtype=torch.float16, device=DEV)
        lo = torch.full((B, 2, Hkv, cap, D), 7.0, dtype=torch.float16, device=DEV) if two else None
        q = torch.full((T, H * D), float("nan"), dtype=torch.float16, device=DEV)
        ql = torch.full((T, H * D), float("nan"), dtype=torch.float16, device=DEV) if two else None
        return arena, lo, q, ql

    a1, l1, q1, ql1 = fresh()
    kvlo1 = (l1[:, 0], l1[:, 1], bs, hs, 0) if two else None
    n.gemm_dense_qkv_rope(x_hi, x_lo, w, K, cs, q1, ql1, H * D, a1[:, 0], a1[:, 1], bs, hs, B, H, Hkv, D, q_len, max(pasts), cap,
                          kv_lo=kvlo1, past_lens=pl)
    a2, l2, q2, ql2 = fresh()
    kvlo2 = (l2[:, 0], l2[:, 1], bs, hs, 0) if two else None
    qkv = torch.empty((T, W), dtype=torch.float32, device=DEV)
    n.gemm_dense(x_hi, x_lo, w, T, W, K, n.EPI_STORE, y=qkv)
    n.rope_append(qkv, q_len * W, W, q2, q_len * H * D, H * D, qkv[:, H * D:], qkv[:, (H + Hkv) * D:], q_len * W, W,
                  a2[:, 0], a2[:, 1], bs, hs, cs, B, H, Hkv, D, q_len, max(pasts), cap, True, q_out_lo=ql2, kv_lo=kvlo2, past_lens=pl)
    torch.cuda.synchronize()
    assert torch.equal(a1[:, 1], a2[:, 1])                                     # V: no arithmetic between the accumulator and the split
    for b, p in enumerate(pasts):                                              # untouched rows stay untouched
        assert torch.equal(a1[b, :, :, :p], a2[b, :, :, :p]) and torch.equal(a1[b, :, :, p + q_len:], a2[b, :, :, p + q_len:])
    full = lambda hi, lo_: hi.double() + (lo_.double() if lo_ is not None else 0.0)   # noqa: E731
    tol = 1e-6 if two else 2e-3
    assert (full(q1, ql1) - full(q2, ql2)).abs().max().item() <= tol * max(1.0, full(q2, ql2).abs().max().item())
    for b, p in enumerate(pasts):
        k1 = full(a1[b, 0, :, p:p + q_len], l1[b, 0, :, p:p + q_len] if two else None)
        k2 = full(a2[b, 0, :, p:p + q_len], l2[b, 0, :, p:p + q_len] if two else None)
        assert (k1 - k2).abs().max().item() <= tol * max(1.0, k2.abs().max().item())
        if two:
            assert torch.equal(l1[b, 1, :, p:p + q_len], l2[b, 1, :, p:p + q_len])


def _attn_ref64(q, k, v, past):
    """float64 attention of q [T,H,D] over k / v [Hkv,S,D] under the index-order causal mask (row i sees keys < past + i + 1)."""
    T, H, D = q.shape
    rep = H // k.shape[0]
    kk, vv = k.repeat_interleave(rep, 0), v.repeat_interleave(rep, 0)
    sc = torch.einsum("thd,hsd->hts", q, kk) / np.sqrt(D)
    S = kk.shape[1]
    mask = torch.arange(S, device=q.device)[None, :] <= (past + torch.arange(T, device=q.device))[:, None]
    sc = sc.masked_fill(~mask[None], float("-inf"))
    return torch.einsum("hts,hsd->thd", torch.softmax(sc, -1), vv).reshape(T, H * D)


@pytest.mark.parametrize("q_len,past,lo_mode,frag", [
    (259, 1000, "own", True),      # config-4 shape class: staged plain keys, the pass's rows with residuals, fragment output, KV splits
    (130, 77, "own", False),       # plain region shorter than one stage, not tile-aligned
    (200, 0, "all", False),        # schema-encode pass: every key has a residual row, no past
    (300, 513, "none", False),     # split-precision Q / P only
    (65, 3000, "own", False),      # one q-block with one full wave-tile row block + 1 row, many splits
    (128, 128, "all", False),
])
def test_ring_attention_matches_float64_and_the_64_row_kernel(q_len, past, lo_mode, frag, monkeypatch):
    """attn_ring_kernel (pc_attn_ring.hip: > 64 split-precision rows at head_dim 128, 128 rows per workgroup, K / V tiles by
    LDS-DMA) against a float64 reference on the very operands (hi + lo planes), and against attn_fwd_kernel on the same call
    (PC_ATTN_NO_RING=1): same keys in the same order, only the tile boundaries differ -> fp32 round-off."""
    n = _n()
    rng = np.random.default_rng(q_len + past)
    B, H, Hkv, D = 1, 8, 4, 128
    S = past + q_len
    cap = S + 7

    def f16(*shape, scale=0.5):
        return torch.from_numpy((scale * rng.standard_normal(shape, dtype=np.float32)).astype(np.float16)).to(DEV)

    arena = f16(2, Hkv, cap, D)
    q, q_lo = f16(q_len, H * D), f16(q_len, H * D, scale=2.0 ** -12)
    kvlo, k_eff, v_eff = None, arena[0, :, :S].double(), arena[1, :, :S].double()
    if lo_mode == "own":          # residual rows for the rows this pass appended (compact planes, lo_row0 = -1 -> past_len)
        lo = f16(2, Hkv, q_len + 5, D, scale=2.0 ** -12)
        kvlo = (lo[0], lo[1], Hkv * (q_len + 5) * D, (q_len + 5) * D, -1)
        k_eff = k_eff.clone(); v_eff = v_eff.clone()
        k_eff[:, past:] += lo[0, :, :q_len].double(); v_eff[:, past:] += lo[1, :, :q_len].double()
    elif lo_mode == "all":        # arena-shaped residual planes (encode): every key
        lo = f16(2, Hkv, cap, D, scale=2.0 ** -12)
        kvlo = (lo[0], lo[1], Hkv * cap * D, cap * D, 0)
        k_eff = k_eff + lo[0, :, :S].double(); v_eff = v_eff + lo[1, :, :S].double()
    ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, S), 4) // 4, dtype=torch.float32, device=DEV)
    mt = (q_len + 15) // 16

    def run():
        if frag:
            ah = torch.zeros((mt, H * D // 32, 64, 8), dtype=torch.float16, device=DEV)
            al = torch.zeros_like(ah)
            n.attn_fwd(q, q_len * H * D, H * D, arena[0], arena[1], 2 * Hkv * cap * D, cap * D, None, 0, 0, B, H, Hkv, D, q_len, past,
                       1.0 / np.sqrt(D), ws, out_frag=(ah, al), q_lo=q_lo, kv_lo=kvlo)
            torch.cuda.synchronize()
            full = (ah.double() + al.double()).view(mt, H * D // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(mt * 16, H * D)
            return full[:q_len]
        out = torch.full((q_len, H * D), float("nan"), dtype=torch.float16, device=DEV)
        out_lo = torch.empty_like(out)
        n.attn_fwd(q, q_len * H * D, H * D, arena[0], arena[1], 2 * Hkv * cap * D, cap * D, out, q_len * H * D, H * D, B, H, Hkv, D,
                   q_len, past, 1.0 / np.sqrt(D), ws, q_lo=q_lo, out_lo=out_lo, kv_lo=kvlo)
        torch.cuda.synchronize()
        return out.double() + out_lo.double()

    monkeypatch.delenv("PC_ATTN_NO_RING", raising=False)
    got = run()
    monkeypatch.setenv("PC_ATTN_NO_RING", "1")
    old = run()
    monkeypatch.delenv("PC_ATTN_NO_RING", raising=False)
    ref = _attn_ref64((q.double() + q_lo.double()).view(q_len, H, D), k_eff, v_eff, past)
    assert torch.isfinite(got).all()
    err, err_old = (got - ref).abs().max().item(), (old - ref).abs().max().item()
    assert err < 2e-5 and err <= 2.0 * err_old + 1e-6, (err, err_old)
    assert (got - old).abs().max().item() < 4e-6
    assert torch.equal(got, run())                      # run to run: same bits


@pytest.mark.parametrize("prefix_lo", [True, False])
def test_ring_attention_with_a_shared_prefix_and_ragged_batch_rows(prefix_lo, monkeypatch):
    """The ring kernel's region walk [prefix plain | prefix with residuals | own plain | own with residuals] on a batch whose
    rows sit behind prefixes of different lengths (the suffix batches of a schema encode), against the 64-row kernel."""
    n = _n()
    rng = np.random.default_rng(8)
    B, H, Hkv, D, q_len = 3, 4, 2, 128, 150
    pres, n_trunk = [37, 300, 128], 320

    def f16(*shape, scale=0.5):
        return torch.from_numpy((scale * rng.standard_normal(shape, dtype=np.float32)).astype(np.float16)).to(DEV)

    trunk, trunk_lo = f16(2, Hkv, n_trunk, D), f16(2, Hkv, n_trunk, D, scale=2.0 ** -12)
    own, own_lo = f16(B, 2, Hkv, q_len, D), f16(B, 2, Hkv, q_len, D, scale=2.0 ** -12)
    q, q_lo = f16(B, q_len, H * D), f16(B, q_len, H * D, scale=2.0 ** -12)
    pl = torch.tensor(pres, dtype=torch.int32, device=DEV)
    ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, max(pres) + q_len), 4) // 4, dtype=torch.float32, device=DEV)

    def run():
        out = torch.full((B, q_len, H * D), float("nan"), dtype=torch.float16, device=DEV)
        out_lo = torch.empty_like(out)
        n.attn_fwd(q, q_len * H * D, H * D, own[:, 0], own[:, 1], 2 * Hkv * q_len * D, q_len * D, out, q_len * H * D, H * D, B, H, Hkv, D,
                   q_len, max(pres), 1.0 / np.sqrt(D), ws, q_lo=q_lo, out_lo=out_lo, past_lens=pl,
                   kv_lo=(own_lo[:, 0], own_lo[:, 1], 2 * Hkv * q_len * D, q_len * D, 0),
                   prefix=(trunk[0], trunk[1], trunk_lo[0] if prefix_lo else None, trunk_lo[1] if prefix_lo else None, n_trunk * D))
        torch.cuda.synchronize()
        return out.double() + out_lo.double()

    monkeypatch.delenv("PC_ATTN_NO_RING", raising=False)
    got = run()
    monkeypatch.setenv("PC_ATTN_NO_RING", "1")
    old = run()
    monkeypatch.delenv("PC_ATTN_NO_RING", raising=False)
    assert torch.isfinite(got).all()
    assert (got - old).abs().max().item() < 4e-6
    for b, pre in enumerate(pres):
        kk = torch.cat([trunk[0, :, :pre].double() + (trunk_lo[0, :, :pre].double() if prefix_lo else 0), own[b, 0].double() + own_lo[b, 0].double()], 1)
        vv = torch.cat([trunk[1, :, :pre].double() + (trunk_lo[1, :, :pre].double() if prefix_lo else 0), own[b, 1].double() + own_lo[b, 1].double()], 1)
        ref = _attn_ref64((q[b].double() + q_lo[b].double()).view(q_len, H, D), kk, vv, pre)
        assert (got[b] - ref).abs().max().item() < 2e-5


def test_shared_prefix_argument_checks():
    n = _n()
    t = torch.zeros((2, 2, 64, 128), dtype=torch.float16, device=DEV)
    q = torch.zeros((1, 32, 256), dtype=torch.float16, device=DEV)
    out = torch.empty_like(q)
    pl = torch.tensor([8], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="shared prefix"):      # no past_lens
        n.attn_fwd(q, 32 * 256, 256, t[0], t[1], 2 * 64 * 128, 64 * 128, out, 32 * 256, 256, 1, 2, 2, 128, 32, 8, 0.1, None,
                   prefix=(t[0], t[1], None, None, 64 * 128))
    with pytest.raises(RuntimeError, match="go together"):        # prefix residuals without k_lo / v_lo
        n.attn_fwd(q, 32 * 256, 256, t[0], t[1], 2 * 64 * 128, 64 * 128, out, 32 * 256, 256, 1, 2, 2, 128, 32, 8, 0.1, None,
                   past_lens=pl, prefix=(t[0], t[1], t[0], t[1], 64 * 128))


def test_greedy_advance_argmax_ties_and_state_words():
    """pc_greedy_advance: argmax with the lowest index among equal maxima (torch.argmax on a contiguous row), written to
    the loop's device words; position and past length advance by one; the ring records the tokens in order."""
    n = _n()
    rng = np.random.default_rng(3)
    ids = torch.zeros(1, dtype=torch.int64, device=DEV)
    pos = torch.tensor([41], dtype=torch.int32, device=DEV)
    past = torch.tensor([1700, 7], dtype=torch.int32, device=DEV)
    ring = torch.full((8,), -1, dtype=torch.int32, device=DEV)
    ctr = torch.zeros(1, dtype=torch.int32, device=DEV)
    want = []
    for V in (32000, 32016, 50432, 1027, 5):
        x = rng.standard_normal(V).astype(np.float32)
        if V > 100:
            top = x.max() + 1.0
            where = sorted(rng.choice(V, size=3, replace=False).tolist())
            x[where] = top                                     # three equal maxima: the first one wins
        t = torch.zeros(V + 4, dtype=torch.float32, device=DEV)[:V]
        t.copy_(torch.from_numpy(x))
        n.greedy_advance(t, V, ids, pos, past, ring, ctr)
        torch.cuda.synchronize()
        want.append(int(np.argmax(x)))
        assert int(ids[0]) == want[-1] == int(torch.argmax(t))
    assert int(pos[0]) == 41 + 5 and past.tolist() == [1705, 7] and int(ctr[0]) == 5
    assert ring[:5].tolist() == want and ring[5:].tolist() == [-1, -1, -1]
