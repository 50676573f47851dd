"""Host-side checks of the KV arena bookkeeping (no GPU)."""
import torch


def test_plain_views_resolve_to_the_live_arena():
    """The reference's GenerationEngine hands the model a rebuilt list of views (generation_engine.py:101-102); the arena
    behind them must be found again as the SAME object (its residual tail and captured graphs hang off it)."""
    from promptcache_amd.model.kv_arena import KVArena, arena_from_past
    a = KVArena(1, 2, 4, 64, 32, "cpu")
    b = KVArena(1, 2, 4, 64, 32, "cpu")
    a.tail_base, a.tail_len = 7, 3
    rebuilt = [(k[0].unsqueeze(0), v[0].unsqueeze(0)) for k, v in a.views(10)]
    got, S = arena_from_past(rebuilt, 2, 4, 32)
    assert got is a and S == 10 and (got.tail_base, got.tail_len) == (7, 3)
    got_b, _ = arena_from_past([(k[0].unsqueeze(0), v[0].unsqueeze(0)) for k, v in b.views(5)], 2, 4, 32)
    assert got_b is b
    foreign = [(torch.zeros(1, 4, 10, 32, dtype=torch.float16), torch.zeros(1, 4, 10, 32, dtype=torch.float16)) for _ in range(2)]
    assert arena_from_past(foreign, 2, 4, 32) is None


def test_staged_views_answer_every_read_accessor():
    """``StagedKV`` builds its views lazily; every way of LOOKING at it -- not only indexing and iterating -- must see
    all layers (a ``list`` subclass would have shown ``copy()``, ``+``, ``==``, ``in``, ``reversed``, pickling and
    ``PySequence_Fast`` consumers an empty list)."""
    import copy
    import pickle
    from promptcache_amd.model.kv_arena import KVArena, StagedKV
    a = KVArena(1, 3, 2, 16, 8, "cpu")
    a.buf.copy_(torch.arange(a.buf.numel(), dtype=torch.float32).reshape(a.buf.shape).to(torch.float16))
    v = a.views(5)
    assert isinstance(v, StagedKV) and len(v) == 3
    fresh = lambda: a.views(5)
    assert len(fresh().copy()) == 3 and len(fresh() + []) == 3 and len([] + fresh()) == 3
    assert len(list(reversed(fresh()))) == 3 and len(tuple(fresh())) == 3 and len([*fresh()]) == 3
    k2 = fresh()[2][0]
    assert k2.shape == (1, 2, 5, 8) and torch.equal(k2, a.buf[:, 2, 0, :, :5])
    assert fresh().unbatched()[1][1].shape == (2, 5, 8)
    assert len(copy.copy(fresh())) == 3
    restored = pickle.loads(pickle.dumps(fresh()))
    assert isinstance(restored, list) and len(restored) == 3 and torch.equal(restored[1][0], a.buf[:, 1, 0, :, :5])
    w = fresh()
    w[0] = ("k", "v")
    assert w[0] == ("k", "v") and len(w) == 3
    # a sequence consumer that is not written for this class
    assert len(torch.nn.utils.rnn.pad_sequence([t[0][0, 0] for t in fresh()], batch_first=True)) == 3


def test_assigned_entry_switches_the_arena_shortcut_off():
    """``past[i] = (K, V)`` used to change only the lazily built view list while the model kept reading ``.arena``;
    now a replaced entry makes the object foreign data: ``arena_from_past`` no longer returns the aliasing arena (the model then
    copies the items -- the replacement included -- into a fresh one)."""
    from promptcache_amd.model.kv_arena import KVArena, arena_from_past
    a = KVArena(1, 2, 4, 64, 32, "cpu")
    v = a.views(10)
    assert arena_from_past(v, 2, 4, 32)[0] is a
    k1 = torch.ones(1, 4, 10, 32, dtype=torch.float16)
    v[1] = (k1, k1.clone())
    v[1] = (k1, k1.clone())                  # (a second assignment must not lose the arena behind len() / repr())
    assert v.arena is None and arena_from_past(v, 2, 4, 32) is None
    assert len(v) == 2 and v[1][0] is k1 and "replaced=True" in repr(v)
    assert v[0][0].data_ptr() == a.buf[:, 0, 0].data_ptr()      # untouched layers still alias the arena's rows
