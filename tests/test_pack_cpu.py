"""SchemaCache._pack: how the encode groups its scaffold passes into right-padded batches (host logic, CPU)."""
import types

from promptcache_amd.cache_engine import SchemaCache


def _sc(batch_invariant=True):
    sc = SchemaCache.__new__(SchemaCache)
    sc.lm = types.SimpleNamespace(hf_model=types.SimpleNamespace(batch_invariant=batch_invariant))
    return sc


def _padded(groups, lengths):
    return sum(len(g) * max(lengths[i] for i in g) for g in groups)


def test_pack_groups_similar_lengths_and_is_a_partition():
    # the persona-structured schema's suffix passes (tokens up to the last owned one)
    lengths = [0, 232, 240, 226, 229, 265, 250, 258, 271, 267, 259, 270, 255, 263, 155, 149, 161, 158, 150, 256, 240, 249, 262,
               251, 174, 169, 181, 160, 177]
    mine = list(range(1, len(lengths)))
    sc = _sc()
    groups = sc._pack(mine, lengths, 1)
    assert sorted(i for g in groups for i in g) == mine
    for g in groups:
        assert [lengths[i] for i in g] == sorted((lengths[i] for i in g), reverse=True)
        assert len(g) <= sc.encode_rows_max and (len(g) == 1 or len(g) * lengths[g[0]] <= sc.encode_token_budget)
    real = sum(lengths[i] for i in mine)
    assert _padded(groups, lengths) <= 1.07 * real                  # greedy fill-to-budget padded this plan by 17 %
    # never worse than one batch per budget-full of longest-first passes, forward cost included
    order = sorted(mine, key=lambda i: -lengths[i])
    greedy, cur = [], []
    for i in order:
        if cur and (len(cur) + 1) * lengths[cur[0]] > sc.encode_token_budget:
            greedy.append(cur); cur = []
        cur.append(i)
    greedy.append(cur)
    cost = lambda gs: _padded(gs, lengths) + sc.encode_forward_cost * len(gs)     # noqa: E731
    assert cost(groups) <= cost(greedy)


def test_pack_edge_cases():
    sc = _sc()
    assert sc._pack([], [], 1) == []
    assert sc._pack([3], [0, 0, 0, 9000], 1) == [[3]]                            # a pass longer than the budget travels alone
    assert sc._pack([0, 1, 2, 3, 4], [5, 5, 5, 5, 5], 2) == [[0, 1], [2, 3], [4]]   # the reference's batch_size knob: in order
    same = sc._pack(list(range(40)), [100] * 40, 1)
    assert sorted(i for g in same for i in g) == list(range(40)) and all(len(g) <= sc.encode_rows_max for g in same)
    assert _sc(batch_invariant=False)._pack([0, 1, 2], [7, 9, 8], 1) == [[0], [1], [2]]   # LLM.int8: one scaffold per call
