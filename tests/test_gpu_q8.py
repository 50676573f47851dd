"""pc_gemm_q8 (csrc/pc_gemm_q8.h): LLM.int8 projections of <= 16 rows with the activation quantiser inside the launch, against
the stand-alone quantisers + pc_gemm (bit for bit where the summation order is the same) and against oracle/llmint8_oracle.py --
the published algorithm behind the reference's ``load_in_8bit=True`` (demo.py:27-29)."""
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


def _image_to_rows(img, T, K):
    """[K/64][64 lanes = g*16 + m][16 bytes = k-step 2s (8), 2s+1 (8)] -> codes [T][K]: k = 64 P + 32 half + 8 g + e."""
    a = img.reshape(K // 64, 4, 16, 2, 8)                  # P, g, m, half, e
    return np.transpose(a, (2, 0, 3, 1, 4)).reshape(16, K)[:T]


def _acts(rng, T, K, nout, scale=1.5):
    x = np.clip(rng.standard_normal((T, K)).astype(np.float32) * scale, -5.9, 5.9)
    if nout:
        cols = rng.permutation(K)[:nout]
        x[rng.integers(0, T, size=nout), cols] = rng.choice([7.0, -9.5, 30.0, 6.0], size=nout)
    return x.astype(np.float16).astype(np.float32)


def _dbg(K, T):
    return (torch.zeros((K // 64, 64, 16), dtype=torch.int8, device=DEV), torch.zeros(16, dtype=torch.float32, device=DEV),
            torch.zeros(K, dtype=torch.uint8, device=DEV))


@pytest.mark.parametrize("T,K,N,nout", [(12, 4096, 4096, 0), (12, 4096, 512, 5), (1, 4096, 4096, 0), (1, 4096, 256, 2), (16, 512, 64, 40),
                                        (7, 5120, 5120, 3), (3, 1024, 48, 0)])
def test_o_proj_form_equals_the_quantiser_launch_plus_pc_gemm_bit_for_bit(T, K, N, nout):
    """fp16 plane source, residual add / plain store: codes, scales, flags and the result equal pc_quant_act_i8 + pc_gemm (a8c)."""
    n = _n()
    rng = np.random.default_rng(T * 7 + K + nout)
    x = _acts(rng, T, K, nout)
    w = (0.03 * rng.standard_normal((N, K))).astype(np.float32)
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8, qt = n.to_weight_frags_i8(q), q.t().contiguous()
    hi, _ = n.to_act_frags(torch.from_numpy(x).to(DEV))
    codes, zero = torch.empty_like(hi), torch.zeros_like(hi)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros((2, 16384), dtype=torch.uint8, device=DEV)
    c8 = torch.zeros(((T + 15) // 16, K // 64, 64, 16), dtype=torch.int8, device=DEV)
    n.quant_act_i8(hi, True, T, K, codes, xs, flags[0], flags[1], codes8=c8)
    base = torch.from_numpy(rng.standard_normal((T, N + 4)).astype(np.float32)).to(DEV)
    for epi in (n.EPI_ADD, n.EPI_STORE):
        y_old, y_new = base.clone(), base.clone()
        n.gemm_skinny_a8c(wf8, sc, codes, zero, xs, flags[0], hi, qt, T, N, K, epi, y=y_old, ldy=N + 4, codes8=c8)
        dc, ds, df = _dbg(K, T)
        clr = torch.full((4096,), 7, dtype=torch.uint8, device=DEV)
        n.gemm_q8(epilogue=epi, wf=wf8, w_scale=sc, w_codes_t=qt, xf_hi=hi, M=T, N=N, K=K, y=y_new, ldy=N + 4, dbg_codes=dc, dbg_scale=ds,
                  dbg_flags=df, flags_clear=clr, clear_bytes=4096 - 16)
        torch.cuda.synchronize()
        assert torch.equal(y_old, y_new), float((y_old - y_new).abs().max())
        assert torch.equal(y_new[:, N:], base[:, N:])
        assert torch.equal(ds[:T], xs)
        assert torch.equal(df, flags[0, :K])
        assert np.array_equal(_image_to_rows(dc.cpu().numpy(), T, K), _image_to_rows(c8[0].cpu().numpy(), T, K))
        assert int(clr[:4096 - 16].max()) == 0 and int(clr[4096 - 16:].min()) == 7
    # ... and the oracle (published algorithm)
    qo, so = io.quantize_rows_int8(w)
    ref = lo.linear(x, qo, so)
    y = torch.zeros((T, N), dtype=torch.float32, device=DEV)
    n.gemm_q8(epilogue=n.EPI_STORE, wf=wf8, w_scale=sc, w_codes_t=qt, xf_hi=hi, M=T, N=N, K=K, y=y, ldy=N)
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("T,hid,inter,nout", [(12, 4096, 1024, 0), (12, 4096, 11008, 6), (1, 4096, 1024, 1), (16, 1024, 704, 9), (5, 5120, 512, 2)])
def test_gate_up_form_equals_rmsnorm_quant_plus_pc_gemm_bit_for_bit(T, hid, inter, nout):
    """fp32 residual-stream source (RMSNorm folded in), SiLU epilogue: equals pc_rmsnorm_quant_i8 + pc_gemm (a8c) bit for bit; the
    per-tile row maxima and outlier flags it leaves for down_proj equal what a pass over its output gives."""
    n = _n()
    rng = np.random.default_rng(T + hid + inter)
    x = rng.standard_normal((T, hid)).astype(np.float32) * 1.3
    gam = (1.0 + 0.1 * rng.standard_normal(hid)).astype(np.float16)
    if nout:
        cols = rng.permutation(hid)[:nout]
        x[rng.integers(0, T, size=nout), cols] *= 9.0
    w = (0.2 * rng.standard_normal((2 * inter, hid))).astype(np.float32)       # large enough for |silu(g) * u| to pass 6 here and there
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8, qt = n.to_weight_frags_i8(q), q.t().contiguous()
    xd, gd = torch.from_numpy(x).to(DEV), torch.from_numpy(gam).to(DEV)
    eps = 1e-5
    shape = (1, hid // 32, 64, 8)
    hi, codes = torch.zeros(shape, dtype=torch.float16, device=DEV), torch.zeros(shape, dtype=torch.float16, device=DEV)
    zero = torch.zeros_like(hi)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros((2, 16384), dtype=torch.uint8, device=DEV)
    c8 = torch.zeros((1, hid // 64, 64, 16), dtype=torch.int8, device=DEV)
    n.rmsnorm_quant_i8(xd, gd, eps, T, hid, hi, codes, xs, flags[0], flags[1], codes8=c8)
    oshape = (1, inter // 32, 64, 8)
    oh_a, ol_a, oh_b, ol_b = (torch.zeros(oshape, dtype=torch.float16, device=DEV) for _ in range(4))
    n.gemm_skinny_a8c(wf8, sc, codes, zero, xs, flags[0], hi, qt, T, 2 * inter, hid, n.EPI_SILU, of_hi=oh_a, of_lo=ol_a, codes8=c8)
    dc, ds, df = _dbg(hid, T)
    pm = torch.full((inter // 16, 16), -1.0, dtype=torch.float32, device=DEV)
    fo = torch.zeros(16384, dtype=torch.uint8, device=DEV)
    n.gemm_q8(epilogue=n.EPI_SILU, wf=wf8, w_scale=sc, w_codes_t=qt, x=xd, norm_weight=gd, eps=eps, M=T, N=2 * inter, K=hid, of_hi=oh_b,
              of_lo=ol_b, row_max_out=pm, flags_out=fo, dbg_codes=dc, dbg_scale=ds, dbg_flags=df)
    torch.cuda.synchronize()
    assert torch.equal(ds[:T], xs) and torch.equal(df, flags[0, :hid])
    assert np.array_equal(_image_to_rows(dc.cpu().numpy(), T, hid), _image_to_rows(c8[0].cpu().numpy(), T, hid))
    assert torch.equal(oh_a, oh_b) and torch.equal(ol_a, ol_b)
    # what the epilogue leaves for the consumer
    act = n.from_act_frags(oh_b, T).float().cpu().numpy()
    outl = np.abs(act) >= 6.0
    want_flags = outl.any(axis=0).astype(np.uint8)
    assert np.array_equal(fo.cpu().numpy()[:inter], want_flags) and int(fo[inter:].max()) == 0
    masked = np.where(outl, 0.0, np.abs(act)).reshape(T, inter // 16, 16).max(axis=2)       # [T][units]
    assert np.array_equal(pm.cpu().numpy()[:, :T], masked.T)
    # without the optional outputs: same planes, no lo plane needed
    oh_c = torch.zeros(oshape, dtype=torch.float16, device=DEV)
    n.gemm_q8(epilogue=n.EPI_SILU, wf=wf8, w_scale=sc, w_codes_t=qt, x=xd, norm_weight=gd, eps=eps, M=T, N=2 * inter, K=hid, of_hi=oh_c)
    torch.cuda.synchronize()
    assert torch.equal(oh_c, oh_b)


@pytest.mark.parametrize("T,nout", [(12, 0), (12, 3), (1, 0), (16, 2)])
def test_qkv_form_equals_rmsnorm_quant_plus_pc_gemm_bit_for_bit(T, nout):
    n = _n()
    rng = np.random.default_rng(11 + T + nout)
    B, H, Hkv, D, q_len, past, K = 1, 8, 4, 128, T, 9, 2048
    W = (H + 2 * Hkv) * D
    x = rng.standard_normal((T, K)).astype(np.float32) * 1.3
    if nout:
        cols = rng.permutation(K)[:nout]
        x[rng.integers(0, T, size=nout), cols] *= 9.0
    gam = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float16)
    xd, gd = torch.from_numpy(x).to(DEV), torch.from_numpy(gam).to(DEV)
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
    shape = (1, K // 32, 64, 8)
    hi, codes = torch.zeros(shape, dtype=torch.float16, device=DEV), torch.zeros(shape, dtype=torch.float16, device=DEV)
    zero = torch.zeros_like(hi)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros((2, 16384), dtype=torch.uint8, device=DEV)
    c8 = torch.zeros((1, K // 64, 64, 16), dtype=torch.int8, device=DEV)
    n.rmsnorm_quant_i8(xd, gd, 1e-5, T, K, hi, codes, xs, flags[0], flags[1], codes8=c8)
    outs = []
    for new in (False, True):
        arena = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
        lo_k, lo_v = torch.zeros((B, Hkv, q_len, D), dtype=torch.float16, device=DEV), torch.zeros((B, Hkv, q_len, D), dtype=torch.float16, device=DEV)
        qh = torch.zeros((T, H * D), dtype=torch.float16, device=DEV); ql = torch.zeros_like(qh)
        if new:
            n.gemm_q8(epilogue=n.EPI_QKV_ROPE, wf=wf8p, w_scale=scp, w_codes_t=qtt, row_perm=perm32, x=xd, norm_weight=gd, eps=1e-5, M=T, K=K,
                      cs=cs, q_hi=qh, q_lo=ql, q_token_stride=H * D, k_arena=arena[:, 0], v_arena=arena[:, 1], arena_batch_stride=2 * Hkv * cap * D,
                      arena_head_stride=cap * D, B=B, H=H, Hkv=Hkv, D=D, q_len=q_len, past_len=past, cap=cap, k_lo=lo_k, v_lo=lo_v,
                      lo_batch_stride=Hkv * q_len * D, lo_head_stride=q_len * D, lo_base=-1)
        else:
            n.gemm_qkv_rope_a8c(wf8p, scp, codes, zero, xs, flags[0], hi, qtt, perm32, T, K, cs, qh, ql, H * D, arena[:, 0], arena[:, 1],
                                2 * Hkv * cap * D, cap * D, B, H, Hkv, D, q_len, past, cap, kv_lo=(lo_k, lo_v, Hkv * q_len * D, q_len * D),
                                lo_base=-1, codes8=c8)
        torch.cuda.synchronize()
        outs.append((qh, ql, arena, lo_k, lo_v))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert float(outs[1][2][:, :, :, past:past + q_len].abs().max()) > 0


@pytest.mark.parametrize("T,K,N,nout,tiles,slices", [(12, 11008, 4096, 0, 4, 4), (12, 11008, 4096, 30, 4, 4), (1, 11008, 4096, 2, 4, 4),
                                                     (12, 11008, 4096, 7, 2, 8), (16, 13824, 5120, 5, 8, 3), (9, 1024, 256, 64, 4, 2),
                                                     (12, 11008, 4096, 460, 8, 1), (3, 4096, 64, 3, 2, 1),
                                                     # (<= 4 rows in one slice: the compact-image form of decode)
                                                     (1, 11008, 4096, 0, 1, 1), (1, 11008, 4096, 3, 1, 1), (4, 11008, 4096, 9, 1, 1),
                                                     (2, 13824, 5120, 4, 2, 1), (3, 1024, 256, 64, 1, 1), (4, 16384, 128, 700, 2, 1)])
def test_down_proj_form_matches_the_quantiser_launch_plus_pc_gemm_and_the_oracle(T, K, N, nout, tiles, slices):
    """fp16 plane + the producer's per-tile row maxima and flag bytes, K sliced across workgroups, reduction inside the launch."""
    n = _n()
    rng = np.random.default_rng(T + K + nout + tiles)
    x = _acts(rng, T, K, nout)
    w = (0.03 * rng.standard_normal((N, K))).astype(np.float32)
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8, qt = n.to_weight_frags_i8(q), q.t().contiguous()
    hi, _ = n.to_act_frags(torch.from_numpy(x).to(DEV))
    codes, zero = torch.empty_like(hi), torch.zeros_like(hi)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros((2, 16384), dtype=torch.uint8, device=DEV)
    n.quant_act_i8(hi, True, T, K, codes, xs, flags[0], flags[1])
    # what the producer's epilogue would have left
    outl = np.abs(x) >= 6.0
    pm = np.zeros((K // 16, 16), dtype=np.float32)
    pm[:, :T] = np.where(outl, 0.0, np.abs(x)).reshape(T, K // 16, 16).max(axis=2).T
    pm[:, T:] = 1e30                                       # rows behind M are never read into a result
    pmd = torch.from_numpy(pm).to(DEV)
    fl = torch.zeros(16384, dtype=torch.uint8, device=DEV)
    fl[:K] = torch.from_numpy(outl.any(axis=0).astype(np.uint8)).to(DEV)
    assert torch.equal(fl, flags[0])
    base = torch.from_numpy(rng.standard_normal((T, N + 4)).astype(np.float32)).to(DEV)
    y_old, y_new = base.clone(), base.clone()
    n.gemm_skinny_a8c(wf8, sc, codes, zero, xs, flags[0], hi, qt, T, N, K, n.EPI_ADD, y=y_old, ldy=N + 4)
    scr = torch.empty(n.gemm_skinny_ks_scratch_bytes(N, 8) // 4, dtype=torch.float32, device=DEV)
    ctr = torch.zeros(N // 16, dtype=torch.int32, device=DEV)
    ds = torch.zeros(16, dtype=torch.float32, device=DEV)
    for rep in range(3):                                   # the counters come back to zero
        y_new = base.clone()
        n.gemm_q8(epilogue=n.EPI_ADD, wf=wf8, w_scale=sc, w_codes_t=qt, xf_hi=hi, row_max=pmd, row_max_units=K // 16, flags_in=fl, M=T, N=N, K=K,
                  y=y_new, ldy=N + 4, ks_tiles=tiles, kslices=slices, ks_scratch=scr, ks_scratch_bytes=scr.numel() * 4, ks_counters=ctr, dbg_scale=ds)
        torch.cuda.synchronize()
        assert int(ctr.abs().max()) == 0
        assert torch.equal(ds[:T], xs)
        assert torch.equal(y_new[:, N:], base[:, N:])
        qo, so = io.quantize_rows_int8(w)
        ref = base.cpu().numpy()[:, :N] + lo.linear(x, qo, so)
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(y_new.cpu().numpy()[:, :N] - ref).max() < 3e-5 * scale
        # the summation order differs from the one-workgroup-per-tile launch (slices, then the correction per slice): fp32 rounding only
        assert float((y_new - y_old).abs().max()) < 2e-4 * scale


@pytest.mark.parametrize("S,nout", [(1800, 0), (1800, 3), (300, 1), (5000, 0)])
def test_o_proj_merges_the_attention_partials_itself_bit_for_bit(S, nout):
    """A decode step: pc_attn(defer_merge) leaves the split-KV partials, pc_gemm_q8(part_o) merges them in its prologue -- the same bits
    as attn_combine_kernel + the fp16-plane source."""
    n = _n()
    rng = np.random.default_rng(S + nout)
    H, Hkv, D, K, N = 32, 32, 128, 4096, 4096
    q = torch.from_numpy((rng.standard_normal((1, H * D)) * (3.0 if nout else 1.0)).astype(np.float16)).to(DEV)
    ql = torch.zeros_like(q)
    kv = torch.from_numpy(rng.standard_normal((1, 2, Hkv, S + 8, D)).astype(np.float16)).to(DEV)
    if nout:                                               # a few large value entries: |merged output| >= 6 in some columns
        kv[0, 1, :, :, :nout] *= 40.0
    w = (0.03 * rng.standard_normal((N, K))).astype(np.float32)
    qw, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8, qt = n.to_weight_frags_i8(qw), qw.t().contiguous()
    ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, 1, S + 1), 4) // 4, dtype=torch.float32, device=DEV)
    scale = 1.0 / np.sqrt(D)
    cap = S + 8
    args = (q, H * D, H * D, kv[:, 0], kv[:, 1], 2 * Hkv * cap * D, cap * D, None, 0, 0, 1, H, Hkv, D, 1, S, scale, ws)
    ah, al = (torch.zeros((1, K // 32, 64, 8), dtype=torch.float16, device=DEV) for _ in range(2))
    assert n.attn_fwd(*args, out_frag=(ah, al), q_lo=ql) == 1
    base = torch.from_numpy(rng.standard_normal((1, N)).astype(np.float32)).to(DEV)
    y_a = base.clone()
    dsa = torch.zeros(16, dtype=torch.float32, device=DEV)
    dca, dfa = torch.zeros((K // 64, 64, 16), dtype=torch.int8, device=DEV), torch.zeros(K, dtype=torch.uint8, device=DEV)
    n.gemm_q8(epilogue=n.EPI_ADD, wf=wf8, w_scale=sc, w_codes_t=qt, xf_hi=ah, M=1, N=N, K=K, y=y_a, ldy=N, dbg_codes=dca, dbg_scale=dsa, dbg_flags=dfa)
    ah2, al2 = torch.zeros_like(ah), torch.zeros_like(al)
    ns = n.attn_fwd(*args, out_frag=(ah2, al2), q_lo=ql, defer_merge=True)
    assert 2 <= ns <= 8 and float(ah2.abs().max()) == 0.0      # (no merge ran)
    y_b = base.clone()
    dsb = torch.zeros(16, dtype=torch.float32, device=DEV)
    dcb, dfb = torch.zeros_like(dca), torch.zeros_like(dfa)
    n.gemm_q8(epilogue=n.EPI_ADD, wf=wf8, w_scale=sc, w_codes_t=qt, part_o=ws, part_ml=ws[H * ns * D:], part_nsplit=ns, part_head_dim=D,
              M=1, N=N, K=K, y=y_b, ldy=N, dbg_codes=dcb, dbg_scale=dsb, dbg_flags=dfb)
    torch.cuda.synchronize()
    assert int(dfa.sum()) >= (1 if nout else 0)
    assert torch.equal(dfa, dfb) and torch.equal(dsa, dsb)
    assert np.array_equal(_image_to_rows(dca.cpu().numpy(), 1, K), _image_to_rows(dcb.cpu().numpy(), 1, K))
    assert torch.equal(y_a, y_b)


@pytest.mark.parametrize("shape_name,layers", [("llama2-7b", 3), ("llama2-13b", 2)])
def test_7b_shape_int8_stack_inlaunch_quantisers_equal_the_quantiser_launches(shape_name, layers):
    """Three layers of the 7b shape, load_in_8bit: the cached step (12 rows: quantiser launches + the F form for down_proj), a 3-row
    step and decode steps (1 row: every quantiser inside its projection, the attention's partials merged by o_proj) against round 4's
    path (PC_INT8_INLAUNCH=0: a quantiser launch in front of every projection).  The P forms are bit-identical; the K-sliced / compact
    down_proj forms add their slices in another order (fp32 rounding)."""
    import dataclasses
    from promptcache_amd.model import Llama2
    from promptcache_amd.model.config import SHAPES
    from promptcache_amd.model.weights import random_weights_device
    # (13b: hidden 5120 -- three chunks per thread in the P form, K = 13824 in the F / C forms, H * D > 4096: o_proj reads the merged plane)
    shape = dataclasses.replace(SHAPES[shape_name], num_hidden_layers=layers, vocab_size=4096)
    w = random_weights_device(shape, DEV, torch.float16, seed=5)
    w["embed"][:, [7, 300, 2049]] *= 40.0                 # outlier feature channels: flagged columns on every layer's q|k|v and gate|up input
    lm = Llama2(name="q8-7b", shape=shape, weights=w, device=DEV, load_in_8bit=True)
    m = lm.hf_model
    assert m.llm_int8 and m.i8_inlaunch
    g = torch.Generator().manual_seed(3)
    S = 700
    ids = torch.randint(0, shape.vocab_size, (1, S), generator=g).to(DEV)
    out0 = lm(input_ids=ids, position_ids=torch.arange(S, device=DEV)[None], use_cache=True)     # many-row path: fills the arena
    results = {}
    for mode in (True, False):
        m.i8_inlaunch = mode
        m._graphs.clear()
        past = out0.past_key_values
        logs = []
        pos = S
        for rows in (12, 3, 1, 1):
            step = torch.randint(0, shape.vocab_size, (1, rows), generator=torch.Generator().manual_seed(rows + pos)).to(DEV)
            o = lm(input_ids=step, position_ids=torch.arange(pos, pos + rows, device=DEV)[None], past_key_values=past, use_cache=True)
            logs.append(o.logits[0].float().cpu().numpy())
            past = o.past_key_values
            pos += rows
        results[mode] = logs                               # (the next mode starts from out0's views again: rows S.. are rewritten)
    for a, b in zip(results[True], results[False]):
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() < 2e-3 * max(1.0, np.abs(b).max()), np.abs(a - b).max()


@pytest.mark.parametrize("S,H,D", [(1800, 32, 128), (300, 32, 128), (5000, 16, 128), (900, 8, 64)])
def test_fp16_o_proj_merges_the_attention_partials_itself_bit_for_bit(S, H, D):
    """pc_gemm_part (a decode step of the fp16 model): pc_attn(defer_merge) + o_proj on the partials = pc_attn + merge launch + pc_gemm."""
    n = _n()
    rng = np.random.default_rng(S + H)
    Hkv, K, N = H, H * D, 4096 if H * D >= 2048 else 512
    q = torch.from_numpy(rng.standard_normal((1, K)).astype(np.float16)).to(DEV)
    ql = torch.from_numpy((rng.standard_normal((1, K)) * 1e-3).astype(np.float16)).to(DEV)
    cap = S + 8
    kv = torch.from_numpy(rng.standard_normal((1, 2, Hkv, cap, D)).astype(np.float16)).to(DEV)
    w = torch.from_numpy((0.03 * rng.standard_normal((N, K))).astype(np.float16)).to(DEV)
    wf = n.to_weight_frags(w)
    ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, 1, S + 1), 4) // 4, dtype=torch.float32, device=DEV)
    args = (q, K, K, kv[:, 0], kv[:, 1], 2 * Hkv * cap * D, cap * D, None, 0, 0, 1, H, Hkv, D, 1, S, 1.0 / np.sqrt(D), ws)
    ah, al = (torch.zeros((1, K // 32, 64, 8), dtype=torch.float16, device=DEV) for _ in range(2))
    assert n.attn_fwd(*args, out_frag=(ah, al), q_lo=ql) == 1
    base = torch.from_numpy(rng.standard_normal((1, N)).astype(np.float32)).to(DEV)
    y_a = base.clone()
    n.gemm_skinny(wf, ah, al, 1, N, K, n.EPI_ADD, y=y_a, ldy=N)
    ah2, al2 = torch.zeros_like(ah), torch.zeros_like(al)
    ns = n.attn_fwd(*args, out_frag=(ah2, al2), q_lo=ql, defer_merge=True)
    if ns == 1:                                            # (a launch shape without key splits merges nothing: planes final)
        assert torch.equal(ah, ah2)
        return
    assert 2 <= ns <= 8 and float(ah2.abs().max()) == 0.0
    y_b = base.clone()
    n.gemm_part(wf, ws, ws[H * ns * D:], ns, H, D, N, y_b)
    torch.cuda.synchronize()
    assert torch.equal(y_a, y_b), float((y_a - y_b).abs().max())
    ref = base.cpu().numpy() + (n.from_act_frags(ah, 1).float().cpu().numpy() + n.from_act_frags(al, 1).float().cpu().numpy()) @ w.float().cpu().numpy().T
    assert np.abs(y_b.cpu().numpy() - ref).max() < 1e-3


@pytest.mark.parametrize("T,nout", [(12, 0), (12, 4), (16, 30), (5, 1)])
def test_image_source_equals_pc_gemm_on_the_quantiser_outputs_bit_for_bit(T, nout):
    """5..16 rows: the quantiser stays a launch, pc_gemm_q8 takes its operand image / scales / flags (x_codes8) -- residual add, SiLU and
    q|k|v epilogues against pc_gemm with x_scale + flags."""
    n = _n()
    rng = np.random.default_rng(77 + T + nout)
    K, inter = 2048, 1408
    x = _acts(rng, T, K, nout)
    hi, _ = n.to_act_frags(torch.from_numpy(x).to(DEV))
    codes, zero = torch.empty_like(hi), torch.zeros_like(hi)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros((2, 16384), dtype=torch.uint8, device=DEV)
    c8 = torch.zeros((1, K // 64, 64, 16), dtype=torch.int8, device=DEV)
    n.quant_act_i8(hi, True, T, K, codes, xs, flags[0], flags[1], codes8=c8)
    # residual add
    N = 1024
    w = (0.03 * rng.standard_normal((N, K))).astype(np.float32)
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8, qt = n.to_weight_frags_i8(q), q.t().contiguous()
    base = torch.from_numpy(rng.standard_normal((T, N)).astype(np.float32)).to(DEV)
    y_a, y_b = base.clone(), base.clone()
    n.gemm_skinny_a8c(wf8, sc, codes, zero, xs, flags[0], hi, qt, T, N, K, n.EPI_ADD, y=y_a, ldy=N, codes8=c8)
    n.gemm_q8(epilogue=n.EPI_ADD, wf=wf8, w_scale=sc, w_codes_t=qt, xf_hi=hi, x_codes8=c8, x_scale=xs, x_flags=flags[0], M=T, N=N, K=K, y=y_b, ldy=N)
    torch.cuda.synchronize()
    assert torch.equal(y_a, y_b)
    # gate|up + SiLU (+ what it leaves for down_proj)
    w = (0.2 * rng.standard_normal((2 * inter, K))).astype(np.float32)
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8, qt = n.to_weight_frags_i8(q), q.t().contiguous()
    oshape = (1, inter // 32, 64, 8)
    oh_a, ol_a, oh_b, ol_b = (torch.zeros(oshape, dtype=torch.float16, device=DEV) for _ in range(4))
    pm_a, pm_b = (torch.zeros((inter // 16, 16), dtype=torch.float32, device=DEV) for _ in range(2))
    fo_a, fo_b = (torch.zeros(16384, dtype=torch.uint8, device=DEV) for _ in range(2))
    n.gemm_skinny_a8c(wf8, sc, codes, zero, xs, flags[0], hi, qt, T, 2 * inter, K, n.EPI_SILU, of_hi=oh_a, of_lo=ol_a, codes8=c8, row_max_out=pm_a,
                      flags_out=fo_a)
    n.gemm_q8(epilogue=n.EPI_SILU, wf=wf8, w_scale=sc, w_codes_t=qt, xf_hi=hi, x_codes8=c8, x_scale=xs, x_flags=flags[0], M=T, N=2 * inter, K=K,
              of_hi=oh_b, of_lo=ol_b, row_max_out=pm_b, flags_out=fo_b)
    torch.cuda.synchronize()
    assert torch.equal(oh_a, oh_b) and torch.equal(ol_a, ol_b) and torch.equal(pm_a[:, :T], pm_b[:, :T]) and torch.equal(fo_a, fo_b)
    # q|k|v + RoPE + append
    B, H, Hkv, D, q_len, past = 1, 8, 4, 128, T, 9
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
    for new in (False, True):
        arena = torch.zeros((B, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
        qh = torch.zeros((T, H * D), dtype=torch.float16, device=DEV); ql = torch.zeros_like(qh)
        if new:
            n.gemm_q8(epilogue=n.EPI_QKV_ROPE, wf=wf8p, w_scale=scp, w_codes_t=qtt, row_perm=perm32, xf_hi=hi, x_codes8=c8, x_scale=xs,
                      x_flags=flags[0], M=T, K=K, cs=cs, q_hi=qh, q_lo=ql, q_token_stride=H * D, k_arena=arena[:, 0], v_arena=arena[:, 1],
                      arena_batch_stride=2 * Hkv * cap * D, arena_head_stride=cap * D, B=B, H=H, Hkv=Hkv, D=D, q_len=q_len, past_len=past, cap=cap)
        else:
            n.gemm_qkv_rope_a8c(wf8p, scp, codes, zero, xs, flags[0], hi, qtt, perm32, T, K, cs, qh, ql, H * D, arena[:, 0], arena[:, 1],
                                2 * Hkv * cap * D, cap * D, B, H, Hkv, D, q_len, past, cap, codes8=c8)
        torch.cuda.synchronize()
        outs.append((qh, ql, arena))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
