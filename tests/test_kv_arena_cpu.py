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
