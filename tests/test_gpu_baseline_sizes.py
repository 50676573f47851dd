"""Parity against the PINNED numpy oracle at the sizes BASELINE.json quotes: the attention of one layer at
config 4's shape (13b heads, 259 new rows over 8 258 staged keys: the ring kernel on its XCD-aware split grid) and config 2's
(7b heads, 14 new rows over 4 390 staged keys: the staging / streaming kernel), called the way the forward calls it, against
``oracle.llama_oracle.attention_core`` (llama2.py:368-388 restated); and a 4-layer stack at the 13b layer shape over 8 258 staged
rows + 259 new ones against ``LlamaOracle.forward`` on oracle-staged random K/V.  The oracle only pays for the new rows, so these
sizes cost it seconds."""
import dataclasses
import os
import time

import numpy as np
import pytest
import torch

from oracle import llama_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _n():
    from promptcache_amd import _native
    return _native


def _f16(a):
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.float16)).to(DEV)


@pytest.mark.parametrize("H,Hkv,D,q_len,past,stage_while_reading", [
    (40, 40, 128, 259, 8258, False),      # BASELINE config 4: the wide kernel over the staged keys (17 row groups per workgroup, 6 key slices) + the ring kernel over the pass's own rows (residual tiles)
    (40, 40, 128, 259, 8258, True),       # ... the same launch reading the staged rows from one module store and staging them (config 4's first forward)
    (8, 2, 128, 130, 6200, True),         # wide kernel, grouped-query heads (4 per kv head: one writer each), 9 row groups = 2 + 1 + ... + 1, last tile partial
    (4, 4, 128, 300, 6200, False),        # wide kernel, two q-blocks (19 row groups = 10 + 9)
    (32, 32, 128, 14, 4390, False),       # BASELINE config 2: <= 16 rows over the game prompt's staged keys (fp32 tail workgroup)
    (32, 32, 128, 14, 4390, True),        # ... the same launch reading the staged rows from module stores and staging them
    (32, 32, 128, 12, 1725, True),        # the headline step's attention
])
def test_attention_at_baseline_sizes_vs_pinned_oracle(H, Hkv, D, q_len, past, stage_while_reading):
    n = _n()
    rng = np.random.default_rng(100 + q_len)
    cap = past + q_len + 7
    q32 = rng.standard_normal((1, q_len, H, D), dtype=np.float32)
    k_st = (0.7 * rng.standard_normal((Hkv, past, D), dtype=np.float32)).astype(np.float16)        # staged rows: fp16, as the reference stages them
    v_st = rng.standard_normal((Hkv, past, D), dtype=np.float32).astype(np.float16)
    k_new = 0.7 * rng.standard_normal((Hkv, q_len, D), dtype=np.float32)                            # the pass's own rows: fp32 in the reference
    v_new = rng.standard_normal((Hkv, q_len, D), dtype=np.float32)
    q_hi = _f16(q32)
    q_lo = _f16(q32 - q_hi.float().cpu().numpy())
    arena = torch.full((2, Hkv, cap, D), float("nan"), dtype=torch.float16, device=DEV)
    arena[0, :, past:past + q_len] = _f16(k_new)
    arena[1, :, past:past + q_len] = _f16(v_new)
    k_lo = _f16(k_new - arena[0, :, past:past + q_len].float().cpu().numpy()).unsqueeze(0)
    v_lo = _f16(v_new - arena[1, :, past:past + q_len].float().cpu().numpy()).unsqueeze(0)
    gather = None
    if stage_while_reading:
        lens = [306, 2, 2, 2, 2, 76] + [800] * 5 if past == 4390 else [275, 1, 1, 1, 1, 1, 84, 1, 1, 174, 1, 1, 256, 1, 1, 155, 1, 1, 267, 1, 1, 265, 1, 1, 232]
        if past > 5000:
            lens = [258, past - 258 - 3001, 3000, 1]           # (config 4's schema: system blurb | one long context module | ...)
        assert sum(lens) == past
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(int)
        stores = [torch.stack([_f16(k_st[:, o:o + ln]), _f16(v_st[:, o:o + ln])]).unsqueeze(0).contiguous() for o, ln in zip(offs, lens)]   # [1][2][Hkv][len][D]
        arr = np.array([(s.data_ptr(), int(o), int(ln)) for s, o, ln in zip(stores, offs, lens)],
                       dtype=np.dtype([("src", "<u8"), ("dst_row", "<i4"), ("len", "<i4")]))
        segs = torch.from_numpy(arr.view(np.uint8).copy()).to(DEV)
        words = torch.tensor([len(lens), past + q_len], dtype=torch.int32, device=DEV)
        rows = torch.zeros(cap * 16, dtype=torch.uint8, device=DEV)
        n.kv_row_table(segs, words[0:1], 64, words[1:2], arena, Hkv, D, cap, rows)
        gather = (rows, 0, Hkv)
    else:
        arena[0, :, :past] = _f16(k_st)
        arena[1, :, :past] = _f16(v_st)
    ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q_len, past + q_len), 4) // 4, dtype=torch.float32, device=DEV)
    mt = (q_len + 15) // 16
    oh = torch.full((mt, H * D // 32, 64, 8), float("nan"), dtype=torch.float16, device=DEV)
    ol = torch.full_like(oh, float("nan"))
    n.attn_fwd(q_hi, q_len * H * D, H * D, arena[0].unsqueeze(0), arena[1].unsqueeze(0), 2 * Hkv * cap * D, cap * D, None, 0, 0,
               1, H, Hkv, D, q_len, past, 1.0 / np.sqrt(D), ws, out_frag=(oh, ol), q_lo=q_lo,
               kv_lo=(k_lo, v_lo, Hkv * q_len * D, q_len * D, -1), gather=gather)
    torch.cuda.synchronize()
    got = (n.from_act_frags(oh, q_len).float() + n.from_act_frags(ol, q_len).float()).cpu().numpy()
    t0 = time.perf_counter()
    kk = np.concatenate([k_st.astype(np.float32), k_new], axis=1)[None]
    vv = np.concatenate([v_st.astype(np.float32), v_new], axis=1)[None]
    ref = orc.attention_core(q32.transpose(0, 2, 1, 3), kk, vv, past, H // Hkv).transpose(0, 2, 1, 3).reshape(q_len, H * D)
    err = np.abs(got - ref).max()
    print(f"[attention H={H} q={q_len} past={past} staging={stage_while_reading}] max|d| vs attention_core = {err:.2e} "
          f"(max|out| {np.abs(ref).max():.2f}; oracle {time.perf_counter() - t0:.1f} s)")
    assert np.isfinite(got).all()
    # split-precision Q, P and new K/V rows against an fp32 reference: what is left is the fp32 summation order over 8 k keys
    assert err < 2e-4
    if stage_while_reading:
        assert torch.equal(arena[0, :, :past].view(torch.int16), _f16(k_st).view(torch.int16))
        assert torch.equal(arena[1, :, :past].view(torch.int16), _f16(v_st).view(torch.int16))


def test_13b_stack_at_config4_size_vs_llama_oracle():
    """4 layers at the llama2-13b layer shape (small vocabulary), 259 new rows over 8 258 STAGED rows of random fp16 K/V: the
    65..512-row forward (row-split projections with wide panels and K slices, q|k|v slabs, ring attention with KV splits) against
    LlamaOracle.forward over the same staged K/V."""
    from oracle.llama_oracle import LlamaOracle, OracleConfig
    from promptcache_amd.model import Llama2
    from promptcache_amd.model.config import SHAPES
    from promptcache_amd.model.weights import random_weights_device
    L, S, q = 4, 8258, 259
    shape = dataclasses.replace(SHAPES["llama2-13b"], num_hidden_layers=L, vocab_size=8192)
    w = random_weights_device(shape, "cuda:0", torch.float16, seed=17)
    lm = Llama2(name="c4", shape=shape, weights=w, device="cuda:0")
    m = lm.hf_model
    g = torch.Generator().manual_seed(3)
    arena = m.new_arena(1, 9186)
    kv = (torch.randn((L, 2, m.Hkv, S, m.D), generator=g) * torch.tensor([0.5, 1.0]).view(1, 2, 1, 1, 1)).half()
    arena.buf[0, :, :, :, :S] = kv.to("cuda")
    arena.length = S
    ids = torch.randint(3, shape.vocab_size, (1, q), generator=g)
    pos = torch.arange(8300, 8300 + q).unsqueeze(0)
    out = m(input_ids=ids.cuda(), position_ids=pos.cuda(), past_key_values=arena.views(S), use_cache=True)
    got = out.logits[0].cpu().numpy()
    t0 = time.perf_counter()
    cfg = OracleConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, intermediate_size=shape.intermediate_size,
                       num_hidden_layers=L, num_attention_heads=shape.num_attention_heads, num_key_value_heads=shape.num_key_value_heads,
                       rms_norm_eps=shape.rms_norm_eps, rope_theta=shape.rope_theta, inv_freq=m.inv_freq_cpu.numpy())
    oracle = LlamaOracle(cfg, {k: v.float().cpu().numpy() for k, v in w.items()})
    past = [(kv[i, 0].numpy()[None], kv[i, 1].numpy()[None]) for i in range(L)]
    logits, present = oracle.forward(ids.numpy(), pos.numpy(), past=past)
    err = np.abs(got - logits[0]).max()
    # the rows the pass appended: fp16(oracle's fp32 K/V) up to an ulp
    k_new = out.past_key_values[L - 1][0][0, :, S:S + q].float().cpu().numpy()
    kerr = np.abs(k_new - present[L - 1][0][0, :, S:S + q]).max()
    print(f"[13b x {L} layers, S={S} q={q}] max|dlogit| vs LlamaOracle = {err:.2e}, appended K rows {kerr:.2e} "
          f"(max|logit| {np.abs(logits).max():.2f}; oracle {time.perf_counter() - t0:.0f} s)")
    assert err < 1e-2 and kerr < 4e-3


def test_wide_and_ring_paths_agree_at_config4_size():
    """The same launch through the wide staged-key kernel (+ own rows + merge) and through the ring kernel alone (PC_ATTN_NO_WIDE=1):
    two summation orders of the same split-precision arithmetic -- they agree to fp32 rounding, and both leave identical K / V."""
    n = _n()
    H = Hkv = 40; D = 128; q_len, past = 259, 8258
    g = torch.Generator(device=DEV).manual_seed(7)
    cap = past + q_len + 9
    arena = torch.stack([0.7 * torch.randn((Hkv, cap, D), device=DEV, generator=g), torch.randn((Hkv, cap, D), device=DEV, generator=g)]).half()
    q = torch.randn((q_len, H * D), device=DEV, generator=g)
    qh = q.half(); ql = (q - qh.float()).half()
    lo = (torch.randn((2, 1, Hkv, q_len, D), device=DEV, generator=g) * 2 ** -12).half()
    ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q_len, past + q_len), 4) // 4, dtype=torch.float32, device=DEV)
    outs = []
    for no_wide in ("0", "1"):
        os.environ["PC_ATTN_NO_WIDE"] = no_wide
        try:
            mt = (q_len + 15) // 16
            oh = torch.full((mt, H * D // 32, 64, 8), float("nan"), dtype=torch.float16, device=DEV); ol = torch.full_like(oh, float("nan"))
            n.attn_fwd(qh, q_len * H * D, H * D, arena[0].unsqueeze(0), arena[1].unsqueeze(0), 2 * Hkv * cap * D, cap * D, None, 0, 0,
                       1, H, Hkv, D, q_len, past, 1.0 / np.sqrt(D), ws, out_frag=(oh, ol), q_lo=ql,
                       kv_lo=(lo[0], lo[1], Hkv * q_len * D, q_len * D, -1))
            torch.cuda.synchronize()
            outs.append((n.from_act_frags(oh, q_len).float() + n.from_act_frags(ol, q_len).float()).cpu().numpy())
        finally:
            os.environ.pop("PC_ATTN_NO_WIDE", None)
    assert np.isfinite(outs[0]).all() and np.isfinite(outs[1]).all()
    assert np.abs(outs[0] - outs[1]).max() < 2e-6, np.abs(outs[0] - outs[1]).max()
