"""The attention that stages while it reads (pc_attn `gather_rows` + pc_kv_row_table): what PromptCache.update's
copy loop (promptcache/cache_engine.py:135-151) leaves in the staged buffer must be there after the first forward, bit for bit,
and the forward's result must not depend on which way the rows got there."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _n():
    from promptcache_amd import _native
    return _native


def _rand_half(shape, rng, scale=1.0):
    return torch.from_numpy((scale * rng.standard_normal(shape, dtype=np.float32)).astype(np.float16)).to(DEV)


def _plan_block(n, ptrs, lens, offs, total):
    """Device copy of a staging plan the way the model's input block holds it: (segs, nseg word, total word)."""
    arr = np.zeros(max(len(ptrs), 1), dtype=np.dtype([("src", "<u8"), ("dst_row", "<i4"), ("len", "<i4")]))
    for i, (p, ln, o) in enumerate(zip(ptrs, lens, offs)):
        arr[i] = (p, o, ln)
    segs = torch.from_numpy(arr.view(np.uint8).copy()).to(DEV)
    words = torch.tensor([len(ptrs), total], dtype=torch.int32, device=DEV)
    return segs, words


def test_row_table_expands_the_plan_and_marks_rows_nobody_stages():
    n = _n()
    L, Hkv, D, cap = 2, 3, 128, 64
    lens, offs = [5, 1, 17], [4, 9, 10]                 # rows 0..3 kept from a previous prompt, 27.. = this pass's own rows
    stores = [torch.zeros((L, 2, Hkv, ln, D), dtype=torch.float16, device=DEV) for ln in lens]
    arena = torch.zeros((L, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    segs, words = _plan_block(n, [s.data_ptr() for s in stores], lens, offs, 30)
    rows = torch.full((cap * 16,), 0xAB, dtype=torch.uint8, device=DEV)
    n.kv_row_table(segs, words[0:1], 8, words[1:2], arena, Hkv, D, cap, rows)
    torch.cuda.synchronize()
    tab = rows.cpu().numpy().view(np.dtype([("base", "<u8"), ("ps16", "<u4"), ("flags", "<u4")]))
    rb = D * 2
    for r in range(30):
        seg = [i for i in range(3) if offs[i] <= r < offs[i] + lens[i]]
        if seg:
            i = seg[0]
            assert tab[r]["base"] == stores[i].data_ptr() + (r - offs[i]) * rb and tab[r]["ps16"] == lens[i] * rb // 16
            assert tab[r]["flags"] == 0
        else:
            assert tab[r]["base"] == arena.data_ptr() + r * rb and tab[r]["ps16"] == cap * rb // 16
            assert tab[r]["flags"] == n.KV_ROW_STAGED
    assert (rows.cpu().numpy()[30 * 16:] == 0xAB).all()           # entries behind the total are not touched


GATHER_CASES = [
    # H, Hkv, D, q_len, segment lengths, rows kept from a previous staging, tail (split-precision rows of the pass itself)
    (32, 32, 128, 12, [275, 1, 1, 1, 1, 1, 84, 1, 1, 174, 1, 1, 256, 1, 1, 155, 1, 1, 267, 1, 1, 265, 1, 1, 232], 0, True),   # persona
    (32, 32, 128, 12, [275, 1, 1, 1, 1, 1, 84, 1, 1, 174, 1, 1, 256, 1, 1, 155, 1, 1, 267, 1, 1, 265, 1, 1, 232], 0, False),
    (32, 32, 128, 14, [306, 2, 2, 2, 2, 76, 800, 800, 800, 800, 800], 0, True),        # game prompt: several tiles per wave
    (8, 2, 128, 16, [40, 1, 300, 7, 129], 0, True),                                    # GQA: one query head per kv head stages
    (8, 2, 128, 5, [40, 1, 300, 7, 129], 41, True),                                    # first two segments kept in place
    (40, 40, 128, 9, [1000, 3, 500], 0, True),                                         # 13b head count (5 + 1 splits)
    (16, 16, 64, 3, [100, 1, 1, 250], 0, True),                                        # D = 64
    (4, 4, 32, 12, [64, 64, 1, 200], 0, False),                                        # D = 32
    (32, 32, 128, 1, [500, 1, 120], 0, False),                                         # one row (a decode-shaped first call)
    # 17..32 new rows: the tail-mode instantiation of the 64-row kernel (attn_fwd_kernel<D, true, false, false, GATHER>)
    (32, 32, 128, 26, [275, 1, 1, 1, 1, 1, 84, 1, 1, 174, 1, 1, 256, 1, 1, 155, 1, 1, 267, 1, 1, 265, 1, 1, 232], 0, True),
    (8, 2, 128, 32, [40, 1, 300, 7, 129], 41, True),                                   # GQA, kept prefix, a full 32-row tile pair
    (16, 16, 64, 20, [100, 1, 1, 250], 0, True),                                       # D = 64
    (4, 4, 32, 17, [64, 64, 1, 90], 0, True),                                          # D = 32, short cache (< 256 keys)
    # more than 32 new rows at head_dim 128: the ring kernel (attn_ring_kernel<KVLO, false, GATHER>): the q-block-0 workgroup of
    # each kv head stores every landed stage; the row-table entries travel by LDS-DMA one stage ahead of the tiles
    (32, 32, 128, 100, [275, 1, 1, 1, 1, 1, 84, 1, 1, 174, 1, 1, 256, 1, 1, 155, 1, 1, 267, 1, 1, 265, 1, 1, 232], 0, True),
    (40, 40, 128, 259, [8000], 0, True),                                               # config 4: three q-blocks, XCD remap, no KV splits
    (8, 2, 128, 130, [40, 1, 300, 7, 129], 41, True),                                  # GQA, kept prefix, two q-blocks, KV splits
    (32, 32, 128, 64, [500, 1, 120], 0, False),                                        # no residual planes for the pass's own rows
    (4, 4, 128, 40, [30], 0, True),                                                    # less than one stage of staged keys
    (4, 4, 128, 33, [127, 1, 128, 1, 255], 0, True),                                   # segment ends at and next to stage boundaries
    (16, 16, 128, 300, [1, 1, 1, 3000, 1, 1], 0, True),
]


@pytest.mark.parametrize("H,Hkv,D,q_len,lens,kept,tail", GATHER_CASES)
def test_attention_stages_while_it_reads_bit_exact(H, Hkv, D, q_len, lens, kept, tail):
    """pc_attn(gather_rows) on a poisoned arena vs pc_kv_gather + pc_attn on a second arena: same output bits, same staged
    bytes, nothing else written (layer 1 of a 3-layer store, so the plane arithmetic is exercised)."""
    n = _n()
    rng = np.random.default_rng(7)
    L, li = 3, 1
    S = sum(lens)
    cap = S + q_len + 5
    stores = [_rand_half((L, 2, Hkv, ln, D), rng) for ln in lens]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(int).tolist()
    q32 = rng.standard_normal((1, q_len, H, D), dtype=np.float32)
    q = torch.from_numpy(q32.astype(np.float16)).to(DEV)
    ql = torch.from_numpy((q32 - q.float().cpu().numpy()).astype(np.float16)).to(DEV)
    new_kv = _rand_half((L, 2, Hkv, q_len, D), rng)
    klo, vlo = _rand_half((1, Hkv, q_len, D), rng, 1e-4), _rand_half((1, Hkv, q_len, D), rng, 1e-4)
    kv_lo = (klo, vlo, Hkv * q_len * D, q_len * D, -1) if tail else None
    scale = 1.0 / np.sqrt(D)
    ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q_len, S + q_len), 4) // 4, dtype=torch.float32, device=DEV)
    mt = (q_len + 15) // 16

    def arena_with(staged_upto):
        a = torch.full((L, 2, Hkv, cap, D), float("nan"), dtype=torch.float16, device=DEV)
        a[:, :, :, S:S + q_len] = new_kv                                  # the pass's own rows (appended by the q|k|v launch)
        k = [i for i in range(len(lens)) if offs[i] + lens[i] <= staged_upto]
        if k:
            n.kv_gather([stores[i].data_ptr() for i in k], [lens[i] for i in k], [offs[i] for i in k], a, L, Hkv, D, cap)
        return a

    def attn(a, gather):
        oh = torch.full((mt, H * D // 32, 64, 8), float("nan"), dtype=torch.float16, device=DEV)
        ol = torch.full_like(oh, float("nan"))
        n.attn_fwd(q, q_len * H * D, H * D, a[li, 0].unsqueeze(0), a[li, 1].unsqueeze(0), L * 2 * Hkv * cap * D, cap * D, None, 0, 0,
                   1, H, Hkv, D, q_len, S, scale, ws, out_frag=(oh, ol), q_lo=ql, kv_lo=kv_lo, gather=gather)
        torch.cuda.synchronize()
        return oh, ol

    # reference: everything staged by the copy kernel, plain attention
    ref_arena = arena_with(S)
    ref = attn(ref_arena, None)
    # staging attention: only the kept rows are there; the plan covers the rest
    got_arena = arena_with(kept)
    todo = [i for i in range(len(lens)) if offs[i] >= kept]
    assert sum(lens[i] for i in range(len(lens)) if i not in todo) == kept
    segs, words = _plan_block(n, [stores[i].data_ptr() for i in todo], [lens[i] for i in todo], [offs[i] for i in todo], S + q_len)
    rows = torch.zeros(cap * 16, dtype=torch.uint8, device=DEV)
    n.kv_row_table(segs, words[0:1], 64, words[1:2], got_arena, Hkv, D, cap, rows)
    assert n.attn_gather_ok(q, q_len * H * D, H * D, got_arena[li, 0].unsqueeze(0), got_arena[li, 1].unsqueeze(0),
                            L * 2 * Hkv * cap * D, cap * D, None, 0, 0, 1, H, Hkv, D, q_len, S, scale, ws,
                            out_frag=(ws, ws), q_lo=ql, kv_lo=kv_lo)
    got = attn(got_arena, (rows, li * 2 * Hkv, (li * 2 + 1) * Hkv))
    assert torch.equal(got[0].view(torch.int16), ref[0].view(torch.int16)) and torch.equal(got[1].view(torch.int16), ref[1].view(torch.int16))
    # layer li now holds what pc_kv_gather leaves; the other layers and the rows behind S + q are untouched (still NaN / kept)
    assert torch.equal(got_arena[li].view(torch.int16), ref_arena[li].view(torch.int16))
    other = [x for x in range(L) if x != li]
    exp_other = arena_with(kept)
    assert torch.equal(got_arena[other].view(torch.int16), exp_other[other].view(torch.int16))


def test_gather_rows_is_refused_where_no_kernel_implements_it():
    n = _n()
    H, D, cap = 4, 128, 600
    a = torch.zeros((1, 2, H, cap, D), dtype=torch.float16, device=DEV)
    q = torch.zeros((1, 40, H, D), dtype=torch.float16, device=DEV)
    rows = torch.zeros(cap * 16, dtype=torch.uint8, device=DEV)
    ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, 40, 540), 4) // 4, dtype=torch.float32, device=DEV)
    out = torch.zeros((1, 40, H * D), dtype=torch.float16, device=DEV)
    args = (q, 40 * H * D, H * D, a[0, 0].unsqueeze(0), a[0, 1].unsqueeze(0), 2 * H * cap * D, cap * D, out, 40 * H * D, H * D,
            1, H, H, D, 40, 500, 0.1, ws)
    assert not n.attn_gather_ok(*args)                        # 40 rows: the 64-row kernel, which reads the arena only
    with pytest.raises(RuntimeError, match="gather_rows needs"):
        n.attn_fwd(*args, gather=(rows, 0, H))


# ---------------------------------------------------------------------------------------------------------------------------
# engine level: CacheEngine.process defers the copy, the first lm() call stages
# ---------------------------------------------------------------------------------------------------------------------------

def _engine(defer: bool, shape_name="mid", seed=5):
    from promptcache_amd import CacheEngine
    from promptcache_amd.model import Llama2
    from promptcache_amd.model.config import SHAPES
    from promptcache_amd.model.weights import make_weights_np
    from promptcache_amd import synth
    shape = SHAPES[shape_name]
    lm = Llama2(name="t", shape=shape, weights=make_weights_np(shape, seed, 0.05), device="cuda:0")
    eng = CacheEngine(2048, lm)
    eng.prompt_cache.defer_gather = defer
    schema, prompt = synth.persona_like(system_len=120, seed=3)
    eng.add_schema(lm.get_formatter()(schema))
    return lm, eng, prompt


def test_first_forward_stages_what_the_copy_kernel_would_have_staged():
    from promptcache_amd import Prompt
    out = {}
    for defer in (False, True):
        lm, eng, prompt_pml = _engine(defer)
        prompt = Prompt(prompt_pml, [lm.get_formatter()])
        ids, pos, _, cache = eng.process(prompt)
        arena = eng.prompt_cache.arena
        S = len(eng.prompt_cache)
        assert S + len(ids) >= 256 and len(ids) <= 16, "the synthetic prompt must land in the staging kernel's regime"
        assert (arena.pending is not None) == defer
        if defer:
            arena.buf.fill_(float("nan"))                         # nothing may be read from the arena before it is staged
        # host tensors: the engine's own calling convention (one pinned copy per call)
        o = lm(input_ids=torch.tensor([ids]), position_ids=torch.tensor([pos]), past_key_values=cache, use_cache=True)
        torch.cuda.synchronize()
        assert arena.pending is None
        assert lm.hf_model.stats["fused_gather"] == (1 if defer else 0)
        # ... and a decode step on top
        o2 = lm(input_ids=torch.tensor([[7]]), position_ids=torch.tensor([[max(pos) + 2]]), past_key_values=o.past_key_values, use_cache=True)
        out[defer] = (o.logits.clone(), arena.buf[0, :, :, :, :S + len(ids) + 1].clone(), o2.logits.clone())
    assert torch.isfinite(out[True][0]).all()
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][2], out[False][2])
    assert torch.equal(out[True][1].view(torch.int16), out[False][1].view(torch.int16))


def test_looking_at_the_returned_views_carries_the_staging_out():
    from promptcache_amd import Prompt
    lm, eng, prompt_pml = _engine(True)
    lm2, eng2, _ = _engine(False)
    prompt = Prompt(prompt_pml, [lm.get_formatter()])
    ids, pos, _, cache = eng.process(prompt)
    ids2, pos2, _, cache2 = eng2.process(prompt)
    arena = eng.prompt_cache.arena
    assert arena.pending is not None
    k0 = cache[0][0]                                              # indexing = looking
    assert arena.pending is None
    assert torch.equal(k0.view(torch.int16), cache2[0][0].view(torch.int16))
    # the reference's calling convention: device tensors, the cache rebuilt as a plain list (generation_engine.py:96-102)
    o = lm(input_ids=torch.tensor([ids], device=DEV), position_ids=torch.tensor([pos], device=DEV),
           past_key_values=[(k.unsqueeze(0), v.unsqueeze(0)) for k, v in cache], use_cache=True)
    o2 = lm2(input_ids=torch.tensor([ids2], device=DEV), position_ids=torch.tensor([pos2], device=DEV),
             past_key_values=[(k.unsqueeze(0), v.unsqueeze(0)) for k, v in cache2], use_cache=True)
    assert lm.hf_model.stats["fused_gather"] == 0
    assert torch.equal(o.logits, o2.logits)


def test_a_second_prompt_keeps_the_common_prefix_and_stages_the_rest():
    """Two prompts over one schema back to back: the segments both stage at the same place are kept (flagged STAGED in the row
    table: read in place, not rewritten), the rest arrives with the second prompt's first forward."""
    from promptcache_amd import Prompt, synth
    res = {}
    for defer in (False, True):
        lm, eng, prompt_pml = _engine(defer)
        fmt = lm.get_formatter()
        _, prompt_b = synth.persona_like(system_len=120, seed=3, pick=(0, 0, 3, 0, 1, 2))      # other members of four unions
        logits = []
        for pml in (prompt_pml, prompt_b, prompt_pml):
            ids, pos, _, cache = eng.process(Prompt(pml, [fmt]))
            o = lm(input_ids=torch.tensor([ids]), position_ids=torch.tensor([pos]), past_key_values=cache, use_cache=True)
            logits.append(o.logits.clone())
        S = len(eng.prompt_cache)
        res[defer] = (logits, eng.prompt_cache.arena.buf[0, :, :, :, :S].clone())
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    assert torch.equal(res[True][1].view(torch.int16), res[False][1].view(torch.int16))


def test_generate_through_the_deferred_staging_matches_the_copy_first_engine():
    from promptcache_amd import GenerationEngine, GenerationParameters, Prompt
    texts = {}
    for defer in (False, True):
        lm, eng, prompt_pml = _engine(defer)
        ids, pos, _, cache = eng.process(Prompt(prompt_pml, [lm.get_formatter()]))
        params = GenerationParameters(temperature=0.0, max_new_tokens=12, stop_token_ids=[], stop_str=[])
        outs = list(GenerationEngine(lm).generate(ids, pos, params, cache, stream_interval=1))
        texts[defer] = outs[-1].new_text
        assert lm.hf_model.stats["fused_gather"] == (1 if defer else 0)
    assert texts[True] == texts[False] and len(texts[True]) > 0


def test_device_ids_and_positions_take_the_staging_forward_without_a_read_back():
    """the reference hands the model DEVICE tensors (generation_engine.py:96-97).  With the StagedKV object as the
    cache they take the same captured, staging forward as host tensors do -- copied into the graph's input block on the stream,
    read there by pc_prefill_prologue (words[5]) -- and give the same bits; also with the default position ids and through the
    row-bucket padding (q not a multiple of 16), prefill and decode."""
    from promptcache_amd import Prompt
    res = {}
    for where in ("host", "device"):
        lm, eng, prompt_pml = _engine(True)
        ids, pos, _, cache = eng.process(Prompt(prompt_pml, [lm.get_formatter()]))
        arena = eng.prompt_cache.arena
        arena.buf.fill_(float("nan"))
        kw = dict(device=DEV) if where == "device" else {}
        o = lm(input_ids=torch.tensor([ids], **kw), position_ids=torch.tensor([pos], **kw), past_key_values=cache, use_cache=True)
        assert lm.hf_model.stats["fused_gather"] == 1 and arena.pending is None
        o2 = lm(input_ids=torch.tensor([[7]], **kw), position_ids=torch.tensor([[max(pos) + 2]], **kw), past_key_values=o.past_key_values,
                use_cache=True)
        o3 = lm(input_ids=torch.tensor([[9, 11, 13]], **kw), past_key_values=o2.past_key_values, use_cache=True)   # default positions
        S = len(eng.prompt_cache)
        res[where] = (o.logits.clone(), o2.logits.clone(), o3.logits.clone(), arena.buf[0, :, :, :, :S + len(ids) + 4].clone())
    for a, b in zip(res["host"], res["device"]):
        assert torch.isfinite(a.float()).all() and torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a,
                                                              b.view(torch.int16) if b.dtype == torch.float16 else b)


def test_a_partial_depth_forward_does_not_consume_the_staging_plan():
    """``num_layers`` < L over an arena with a pending plan would stage only those layers' planes inside the attention
    launches while the plan is dropped: such a forward materialises the whole plan first, and the full-depth forward that
    follows reads every layer's staged rows."""
    from promptcache_amd import Prompt
    lm, eng, prompt_pml = _engine(True)
    lm2, eng2, _ = _engine(False)
    prompt = Prompt(prompt_pml, [lm.get_formatter()])
    ids, pos, _, cache = eng.process(prompt)
    ids2, pos2, _, cache2 = eng2.process(prompt)
    arena = eng.prompt_cache.arena
    arena.buf.fill_(float("nan"))
    S = len(eng.prompt_cache)
    lm(input_ids=torch.tensor([ids]), position_ids=torch.tensor([pos]), past_key_values=cache, use_cache=True, num_layers=1)
    assert arena.pending is None and lm.hf_model.stats["fused_gather"] == 0
    assert torch.equal(arena.buf[0, :, :, :, :S].view(torch.int16), eng2.prompt_cache.arena.buf[0, :, :, :, :S].view(torch.int16))
    o = lm(input_ids=torch.tensor([ids]), position_ids=torch.tensor([pos]), past_key_values=eng.prompt_cache.cache, use_cache=True)
    o2 = lm2(input_ids=torch.tensor([ids2]), position_ids=torch.tensor([pos2]), past_key_values=cache2, use_cache=True)
    assert torch.equal(o.logits, o2.logits)


def test_staging_graphs_are_keyed_on_the_row_table_and_a_failed_capture_leaves_no_entry():
    """(i) a staging forward captures the address of the arena's row table, so that address is part of the graph key
    (a new arena on a recycled buffer address must not replay a graph that writes through the old arena's freed table);
    (ii) a capture that raises leaves no half-built entry behind -- the next call captures again instead of replaying None."""
    from promptcache_amd import Prompt
    lm, eng, prompt_pml = _engine(True)
    m = lm.hf_model
    prompt = Prompt(prompt_pml, [lm.get_formatter()])
    ids, pos, _, cache = eng.process(prompt)
    arena = eng.prompt_cache.arena
    m._lo_mode = m._tail_mode(arena, 16, len(eng.prompt_cache))
    k_plain = m._graph_key(arena, 1, 16, len(eng.prompt_cache), False, None, False)
    k_stage = m._graph_key(arena, 1, 16, len(eng.prompt_cache), False, None, True)
    assert k_plain[-1] == 0 and k_stage[-1] == arena.row_table().data_ptr() != 0
    old = arena.row_tab
    arena.row_tab = None                                          # what a fresh arena on the same buffer address would hold
    assert m._graph_key(arena, 1, 16, len(eng.prompt_cache), False, None, True) != k_stage
    del old
    real, calls = m._capture, []

    def failing(*a, **k):
        calls.append(1)
        raise RuntimeError("capture failed (test)")
    m._capture = failing
    n_before = len(m._graphs)
    with pytest.raises(RuntimeError, match="capture failed"):
        lm(input_ids=torch.tensor([ids]), position_ids=torch.tensor([pos]), past_key_values=cache, use_cache=True)
    assert len(m._graphs) == n_before and calls
    m._capture = real
    eng.prompt_cache.reset()
    ids, pos, _, cache = eng.process(prompt)
    o = lm(input_ids=torch.tensor([ids]), position_ids=torch.tensor([pos]), past_key_values=cache, use_cache=True)
    assert torch.isfinite(o.logits).all() and m.stats["fused_gather"] == 1


@pytest.mark.parametrize("T,nseg_plan", [(12, True), (16, False), (1, False), (77, True)])
def test_prefill_prologue_equals_the_separate_launches(T, nseg_plan):
    """pc_prefill_prologue (block fetch + embedding rows as fp32 + rotation table + row table in ONE launch, every role reading
    the pinned host block itself) against pc_fetch_block, pc_embed_gather + cast, pc_rope_table and pc_kv_row_table: bit for bit."""
    n = _n()
    rng = np.random.default_rng(3)
    hid, vocab, D, Hkv, cap, max_seg = 512, 1000, 128, 2, 700, 64
    table = _rand_half((vocab, hid), rng)
    inv = (1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))).to(DEV)
    o_pos, o_words = 8 * T, 12 * T
    o_segs = (o_words + 32 + 15) // 16 * 16
    nbytes = o_segs + 16 * max_seg
    host = torch.zeros(nbytes, dtype=torch.uint8, pin_memory=True)
    h = host.numpy()
    ids = rng.integers(-3, vocab + 5, size=T)                        # (out-of-range ids are clamped, as pc_embed_gather does)
    pos = rng.integers(0, 9000, size=T).astype(np.int32)
    h[:8 * T].view(np.int64)[:] = ids
    h[o_pos:o_pos + 4 * T].view(np.int32)[:] = pos
    words = h[o_words:o_words + 32].view(np.int32)
    lens, offs = [40, 1, 300, 7], [10, 50, 51, 351]
    stores = [torch.zeros((1, 2, Hkv, ln, D), dtype=torch.float16, device=DEV) for ln in lens]
    words[:] = [500, 0, T, len(lens), 358 + T, 0, 0, 0]
    sa = h[o_segs:].view(np.dtype([("src", "<u8"), ("dst_row", "<i4"), ("len", "<i4")]))
    for i, (st, o, ln) in enumerate(zip(stores, offs, lens)):
        sa[i] = (st.data_ptr(), o, ln)
    arena = torch.zeros((1, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
    dev = torch.full((nbytes,), 0xEE, dtype=torch.uint8, device=DEV)
    x = torch.full((T, hid), float("nan"), device=DEV)
    cs = torch.full((T, D // 2, 2), float("nan"), device=DEV)
    rows = torch.full((cap * 16,), 0xAB, dtype=torch.uint8, device=DEV) if nseg_plan else None
    n.prefill_prologue(host, dev, nbytes, T, o_pos, o_words, o_segs, max_seg, table, hid, vocab, x, inv, D, cs,
                       rows=rows, dst=arena if nseg_plan else None, max_ctx=cap if nseg_plan else 0)
    torch.cuda.synchronize()
    assert torch.equal(dev.cpu(), host)
    ids_d = torch.from_numpy(ids.astype(np.int64)).to(DEV)
    h16 = torch.empty((T, hid), dtype=torch.float16, device=DEV)
    n.embed_gather(table, ids_d, h16, T, hid, vocab)
    assert torch.equal(x, h16.float())
    cs2 = torch.empty_like(cs)
    n.rope_table(torch.from_numpy(pos).to(DEV), inv, cs2, T, D)
    assert torch.equal(cs.view(torch.int32), cs2.view(torch.int32))
    if nseg_plan:
        rows2 = torch.full((cap * 16,), 0xAB, dtype=torch.uint8, device=DEV)
        blk = dev
        n.kv_row_table(blk[o_segs:], blk[o_words + 12:o_words + 16].view(torch.int32), max_seg, blk[o_words + 16:o_words + 20].view(torch.int32),
                       arena, Hkv, D, cap, rows2)
        assert torch.equal(rows, rows2)


@pytest.mark.parametrize("q_len,stage", [(22, False), (22, True), (9, False), (30, True)])
def test_tail_mode_streams_sized_for_a_longer_cache_run_empty_without_harm(q_len, stage):
    """tail_stream_splits sizes the streams of a <= 32-row tail-mode launch from the HOST past length (one per key tile); a captured
    forward replays that grid for any shorter cache in its bucket (the kernels read the length from the device word).  Launch with a
    host length of 900 keys (15 streams) over 130 real ones -- 12 streams find no keys -- and compare with the launch sized for 130:
    same result up to the merge's fp32 order, same staged bytes when the launch stages."""
    n = _n()
    rng = np.random.default_rng(5)
    H = Hkv = 8
    D, L, li, S, S_host = 128, 2, 1, 130, 900
    cap = S_host + q_len + 8
    store = _rand_half((L, 2, Hkv, S, D), rng)
    q32 = rng.standard_normal((1, q_len, H, D), dtype=np.float32)
    q = torch.from_numpy(q32.astype(np.float16)).to(DEV)
    ql = torch.from_numpy((q32 - q.float().cpu().numpy()).astype(np.float16)).to(DEV)
    new_kv = _rand_half((L, 2, Hkv, q_len, D), rng)
    klo, vlo = _rand_half((1, Hkv, q_len, D), rng, 1e-4), _rand_half((1, Hkv, q_len, D), rng, 1e-4)
    kv_lo = (klo, vlo, Hkv * q_len * D, q_len * D, -1)
    ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q_len, S_host + q_len), 4) // 4, dtype=torch.float32, device=DEV)
    mt = (q_len + 15) // 16
    past_dev = torch.tensor([S, 0, q_len, 0], dtype=torch.int32, device=DEV)
    outs, arenas = [], []
    for host_len in (S, S_host):
        a = torch.full((L, 2, Hkv, cap, D), float("nan"), dtype=torch.float16, device=DEV)
        a[:, :, :, S:S + q_len] = new_kv
        gather = None
        if stage:
            segs, words = _plan_block(n, [store.data_ptr()], [S], [0], S + q_len)
            rows = torch.zeros(cap * 16, dtype=torch.uint8, device=DEV)
            n.kv_row_table(segs, words[0:1], 64, words[1:2], a, Hkv, D, cap, rows)
            gather = (rows, li * 2 * Hkv, (li * 2 + 1) * Hkv)
        else:
            n.kv_gather([store.data_ptr()], [S], [0], a, L, Hkv, D, cap)
        oh = torch.full((mt, H * D // 32, 64, 8), float("nan"), dtype=torch.float16, device=DEV)
        ol = torch.full_like(oh, float("nan"))
        n.attn_fwd(q, q_len * H * D, H * D, a[li, 0].unsqueeze(0), a[li, 1].unsqueeze(0), L * 2 * Hkv * cap * D, cap * D, None, 0, 0,
                   1, H, Hkv, D, q_len, host_len, 1.0 / np.sqrt(D), ws, past_len_dev=past_dev, out_frag=(oh, ol), q_lo=ql, kv_lo=kv_lo,
                   gather=gather)
        torch.cuda.synchronize()
        outs.append(n.from_act_frags(oh, q_len).float() + n.from_act_frags(ol, q_len).float())
        arenas.append(a[li, :, :, :S].clone())
    assert torch.isfinite(outs[1]).all()
    assert float((outs[0] - outs[1]).abs().max()) < 2e-6 * max(1.0, float(outs[0].abs().max()))
    assert torch.equal(arenas[0].view(torch.int16), arenas[1].view(torch.int16))
    assert torch.equal(arenas[1].view(torch.int16), store[li].view(torch.int16))
